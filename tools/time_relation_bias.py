"""Times the relation-bias backward kernel (csrc/relation_bias.hip) at cfg2's size for several workgroup counts."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ext = importlib.import_module("3dvlp_amd._lib")


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): g.replay()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / (5 * n) * 1e3


B, K = 8, 256
n = int(ext.load().vlp3d_relation_bias_nparam())
centre = torch.randn(B, K, 3, device="cuda")
params = torch.randn(n, device="cuda") * 0.2
dout = torch.randn(B, 4, K, K, device="cuda")
ref = None
for blocks in (256, 384, 512, 768, 1024):
    slabs = torch.empty(blocks, n, device="cuda")
    dp = torch.empty(n, device="cuda")
    us = t(lambda: ext.call("vlp3d_relation_bias_bwd", centre, params, dout, B, K, dp, slabs, blocks, 0))
    if ref is None: ref = dp.clone()
    print(f"blocks {blocks}: {us:.1f} us (bwd + slab sum), rel diff vs 256 blocks {((dp - ref).norm() / ref.norm()).item():.1e}")
out = torch.empty(B, 4, K, K, device="cuda")
print(f"fwd {t(lambda: ext.call('vlp3d_relation_bias_fwd', centre, params, B, K, out)):.1f} us")
