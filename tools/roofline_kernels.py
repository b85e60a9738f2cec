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
    inds, fps_ws = ext.furthest_point_sampling(xyz, m, "pruned", return_workspace=True)
new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
for _ in range(3):
    idx = ext.ball_query_sorted(new_xyz, xyz, 0.2, 64, fps_ws)   # the step's form: one launch on the FPS's spatial sort
for _ in range(3):
    idx_grid = ext.ball_query(new_xyz, xyz, 0.2, 64, "grid")      # round 3's six-launch form, for comparison
assert torch.equal(idx, idx_grid)
rowptr_, crow_ = ext.sa_compact(idx, n)   # the bf16 step evaluates the distinct rows only
cm = (crow_, rowptr_, B * m)
for bf in (1, 0):
    dt = torch.bfloat16 if bf else torch.float32
    C, cout, R = 132, 64, B * m * 64
    K1 = 144 if bf else 136
    W = (torch.randn(cout, K1, device=dev) * 0.05).to(dt)
    Y = torch.empty((R, cout), dtype=dt, device=dev)
    stats = torch.empty((int(ext.load().vlp3d_sa_stat_slabs(R)), 2, cout), dtype=torch.float64, device=dev)
    rows = feat_pm
    if bf:   # what the bf16 step reads: the loader's bf16 copy of the channels (prepare_batch(feat_bf16=True))
        rows = torch.zeros(B, n, (C + 7) // 8 * 8, dtype=torch.bfloat16, device=dev)
        rows[..., :C] = feat_pm
        rows_bf = rows
    for _ in range(3):
        ext.call("vlp3d_sa_fwd_gather", xyz, new_xyz, idx, rows, B, n, m, 64, C, 0.2, W, K1, cout, Y, stats, 3 if bf else 0, *(cm if bf else (None, None, 0)))
# SA1 layer-3 input gradient (pooled-gradient loader, 128 -> 64, mask epilogue) and the gather layer's weight gradient on the
# compact rows, bf16: the two largest main-stream kernels of the backward pass (bench.py roofline candidates)
R = B * m * 64
Y3 = torch.randn(R, 128, device=dev).to(torch.bfloat16)
Y2 = torch.randn(R, 64, device=dev).to(torch.bfloat16)
Y1 = torch.randn(R, 64, device=dev).to(torch.bfloat16)
G1 = torch.randn(R, 64, device=dev).to(torch.bfloat16)
c5_3, c5_1 = torch.rand(5, 128, device=dev) + 0.5, torch.rand(5, 64, device=dev) + 0.5
WT3 = (torch.randn(64, 128, device=dev) * 0.05).to(torch.bfloat16)
vec2 = torch.rand(4, 64, device=dev) + 0.5
nslab = int(ext.load().vlp3d_sa_stat_slabs(R))
t2 = torch.empty((nslab, 2, 64), dtype=torch.float64, device=dev)
gsel = torch.randn(B * m, 128, device=dev)
sel = torch.randint(0, 16, (B * m, 128), device=dev, dtype=torch.uint8)
G2 = torch.empty((R, 64), dtype=torch.bfloat16, device=dev)
dW = torch.empty((64, 144), device=dev)
nblk = 1024
part = torch.empty((nblk, 64, 144), device=dev)
for _ in range(3):
    ext.call("vlp3d_sa_bwd_layer", None, Y3, R, 128, c5_3, WT3, 64, Y2, vec2, G2, t2, gsel, sel, 64, 1, *cm)
    ext.call("vlp3d_sa_wgrad", G1, Y1, R, 64, c5_1, 1, None, 144, None, None, xyz, new_xyz, idx, rows_bf, n, m, 64, 132, 0.2,
             dW, part, nblk, None, None, 0, 3, 0, *cm)
# round 4: the same layer's input / weight gradient WITHOUT its pre-activation (csrc/sa_last.hip): what the step runs at SA1
W3 = (torch.randn(128, 64, device=dev) * 0.05).to(torch.bfloat16)
nb3 = 512
part3 = torch.empty((nb3, 128, 64), device=dev)
for _ in range(3):
    ext.call("vlp3d_sa_last_dgrad", Y2, vec2, c5_3, WT3, gsel, sel, B * m, 64, 64, 128, G2, t2, nslab, *cm)
    ext.call("vlp3d_sa_last_wgrad", Y2, vec2, c5_3, W3, gsel, sel, B * m, 64, 64, 128, part3, nb3, *cm)
q = torch.randn(64, 256, 128, device=dev)
kc = torch.randn(64, 49, 128, device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "self"   # the self- and cross-attention launches share kernel names: two passes
# what the step's match decoder launches since round 4: the cores on bf16 rows (vlp3d_sdpa_fwd_io): q|k|v merged bf16 (self,
# sdpa_fwd_lds_kernel<7>), q bf16 + the tokens' k|v bf16 (cross: the same instantiation, the second pass of this script); out bf16
qkv16 = torch.randn(64, 256, 384, device=dev).bfloat16()
q16 = qkv16[..., :128].contiguous()
kvc = torch.randn(64, 49, 256, device=dev).bfloat16()   # the tokens' k|v: bf16 rows too (linear_rows16)
for _ in range(3):
    if mode == "cross":
        ext.sdpa_fwd_rows(q16, kvc[..., :128], kvc[..., 128:], 4, None, True)
    else:
        ext.sdpa_fwd_rows(qkv16[..., :128], qkv16[..., 128:256], qkv16[..., 256:], 4, None, True)
for _ in range(3):  # the fp32-row form (relation module; exact-fp32 configuration) for comparison
    if mode == "cross":
        fa.sdpa(q, kc, kc, 4, bf16_mma=False)
    else:
        fa.sdpa(q, q, q, 4, bf16_mma=False)
# the decoder layer's row chain (fc_o -> add & norm -> FFN -> add & norm -> the next layer's q|k|v) and its backward, R = 16 384
if mode == "self":
    rc = importlib.import_module("3dvlp_amd.row_chain")
    ml = importlib.import_module("3dvlp_amd.mfma_linear")
    lin = lambda n_, k_: torch.nn.Linear(k_, n_).to(dev)
    fo, l1, l2, nx = lin(128, 128), lin(256, 128), lin(128, 256), lin(384, 128)
    n1, n2 = torch.nn.LayerNorm(128).to(dev), torch.nn.LayerNorm(128).to(dev)
    a0 = torch.randn(16384, 128, device=dev, requires_grad=True)
    x0 = torch.randn(16384, 128, device=dev, requires_grad=True)
    g3, gq = torch.randn(16384, 128, device=dev), torch.randn(16384, 384, device=dev)
    with ml.bf16_mma(True):
        for _ in range(3):
            # as the step launches it since round 4: the core's output as bf16 rows in, q|k|v as bf16 rows out, the FFN stage
            # keeping only its bf16 output (transformer.decoder_stack_chained)
            t = rc.run(a0, [rc.linear_add_norm(fo.weight, fo.bias, n1, x0, 0.1), rc.linear(l1.weight, l1.bias, "relu", 0.1),
                            rc.linear_add_norm(l2.weight, l2.bias, n2, ("tile", 1), 0.1), rc.linear(nx.weight, nx.bias)],
                       x_rows=a0.detach().bfloat16(), last_rows=True, compact_acts=True)
            torch.autograd.backward([t[2], t[3]], [g3, gq])
torch.cuda.synchronize()
