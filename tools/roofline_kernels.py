"""Launch each roofline kernel of bench.py a few times (nothing else) — the target of the rocprofv3 PMC passes:

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python tools/roofline_kernels.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python tools/roofline_kernels.py
"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("3dvlp_amd.synth")
gs = importlib.import_module("3dvlp_amd.grounding_step")
pu = importlib.import_module("3dvlp_amd.pointnet2_utils")
fa = importlib.import_module("3dvlp_amd.fused_attention")
ext = importlib.import_module("3dvlp_amd._lib")
dev = torch.device("cuda:0")
batch = gs.batch_to_device(synth.make_batch(0, 8, 40000, 8), dev)
pc = batch["point_clouds"]
xyz, feat_pm = pc[..., :3].contiguous(), pc[..., 3:].contiguous()
B, n, m = 8, 40000, 2048
for _ in range(3):
    inds = pu.furthest_point_sample(xyz, m)
new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
for _ in range(3):
    idx = pu.ball_query(0.2, 64, xyz, new_xyz)
rowptr_, crow_ = ext.sa_compact(idx, n)   # the bf16 step evaluates the distinct rows only
cm = (crow_, rowptr_, B * m)
for bf in (1, 0):
    dt = torch.bfloat16 if bf else torch.float32
    C, cout, R = 132, 64, B * m * 64
    K1 = 144 if bf else 136
    W = (torch.randn(cout, K1, device=dev) * 0.05).to(dt)
    Y = torch.empty((R, cout), dtype=dt, device=dev)
    stats = torch.empty((int(ext.load().vlp3d_sa_stat_slabs(R)), 2, cout), dtype=torch.float64, device=dev)
    for _ in range(3):
        ext.call("vlp3d_sa_fwd_gather", xyz, new_xyz, idx, feat_pm, B, n, m, 64, C, 0.2, W, K1, cout, Y, stats, bf, *(cm if bf else (None, None, 0)))
q = torch.randn(64, 256, 128, device=dev)
kc = torch.randn(64, 49, 128, device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "self"   # the self- and cross-attention launches share kernel names: two passes
for bf in (True, False):
    for _ in range(3):
        if mode == "cross":
            fa.sdpa(q, kc, kc, 4, bf16_mma=bf)
        else:
            fa.sdpa(q, q, q, 4, bf16_mma=bf)
torch.cuda.synchronize()
