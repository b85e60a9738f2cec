"""Time the fused attention kernels (fwd, bwd) at the grounding shapes (bf16 MFMA): 20 launches per captured graph."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ext = importlib.import_module("3dvlp_amd._lib")
dev = torch.device("cuda:0")


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


for name, b, nq, nk in (("match self", 64, 256, 256), ("match cross", 64, 256, 49), ("relation self+bias", 8, 256, 256)):
    q = torch.randn(b, nq, 128, device=dev); k = torch.randn(b, nk, 128, device=dev); v = torch.randn(b, nk, 128, device=dev)
    bias = torch.randn(b, 4, nq, nk, device=dev) if "bias" in name else None
    go = torch.randn_like(q)
    mode = 1 if bias is not None else 0
    out, lse = ext.sdpa_fwd(q, k, v, 4, bias, mode, None, True)
    f = t(lambda: ext.sdpa_fwd(q, k, v, 4, bias, mode, None, True))
    bw = t(lambda: ext.sdpa_bwd(q, k, v, 4, bias, mode, None, out, lse, go, bias is not None, True))
    print(f"{name:20s} fwd {f:7.1f} us   bwd (dq+dkv) {bw:7.1f} us")
