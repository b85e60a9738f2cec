"""Launch the SA1 layer-1 gather GEMM (bf16) in its processing-order variants — target of rocprofv3 passes:
  rocprofv3 --kernel-trace --stats ... / --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... / --pmc TCC_HIT_sum TCC_MISS_sum
argv[1]: 'fps' (perm None), 'morton' (perm + XCD walk), 'both'."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("3dvlp_amd.synth")
gs = importlib.import_module("3dvlp_amd.grounding_step")
pu = importlib.import_module("3dvlp_amd.pointnet2_utils")
sf = importlib.import_module("3dvlp_amd.sa_fused")
ext = importlib.import_module("3dvlp_amd._lib")
dev = torch.device("cuda:0")
batch = gs.batch_to_device(synth.make_batch(0, 8, 40000, 8), dev)
pc = batch["point_clouds"]
xyz, feat_pm = pc[..., :3].contiguous(), pc[..., 3:].contiguous()
B, n, m = 8, 40000, 2048
inds = pu.furthest_point_sample(xyz, m)
new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
idx = pu.ball_query(0.2, 64, xyz, new_xyz)
C, cout, R, K1 = 132, 64, B * m * 64, 144
W = (torch.randn(cout, K1, device=dev) * 0.05).to(torch.bfloat16)
Y = torch.empty((R, cout), dtype=torch.bfloat16, device=dev)
stats = torch.empty((int(ext.load().vlp3d_sa_stat_slabs(R)), 2, cout), dtype=torch.float64, device=dev)
which = sys.argv[1] if len(sys.argv) > 1 else "both"
import time
for name, p in (("fps", None),):
    if which not in (name, "both"):
        continue
    for _ in range(3):
        ext.call("vlp3d_sa_fwd_gather", xyz, new_xyz, idx, feat_pm, B, n, m, 64, C, 0.2, W, K1, cout, Y, stats, 1, None, None, 0)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ext.call("vlp3d_sa_fwd_gather", xyz, new_xyz, idx, feat_pm, B, n, m, 64, C, 0.2, W, K1, cout, Y, stats, 1, None, None, 0)
    e.record(); e.synchronize()
    print(name, "ms per launch", s.elapsed_time(e) / 10)
