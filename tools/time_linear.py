"""Time the MFMA linear kernels at the transformer shapes of the grounding step: 20 launches per captured graph (launched
one by one from Python these kernels are host bound), with the bytes each one has to move."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ext = importlib.import_module("3dvlp_amd._lib")
dev = torch.device("cuda:0")
BF = int(sys.argv[1]) if len(sys.argv) > 1 else 1


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


for R, K, N in ((2048, 128, 128), (2048, 128, 384), (2048, 128, 256), (2048, 256, 128), (3200, 128, 128), (16384, 128, 128),
                (16384, 128, 256), (16384, 256, 128), (16384, 128, 384)):
    x = torch.randn(R, K, device=dev); w = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev)
    y = torch.empty(R, N, device=dev); dy = torch.randn(R, N, device=dev); dx = torch.empty(R, K, device=dev)
    nblk = max(16, min(128, R // 64)); dwb = torch.empty(N * K + N, device=dev); part = torch.empty(nblk, N * K + N, device=dev)
    f = t(lambda: ext.call("vlp3d_linear_fwd", x, w, b, R, K, N, y, BF))
    g = t(lambda: ext.call("vlp3d_linear_dgrad", dy, w, R, N, K, dx, None, BF))
    h = t(lambda: ext.call("vlp3d_linear_wgrad", dy, x, R, K, N, dwb, part, nblk, 1, 0, BF))
    h2 = t(lambda: ext.call("vlp3d_linear_wgrad", dy, x, R, K, N, dwb, part, nblk, 1, 1, BF))
    mb = (R * K + R * N + N * K) * 4 / 1e6
    print(f"R={R:6d} K={K:3d} N={N:3d} ({mb:5.1f} MB): fwd {f:6.1f} us ({mb / f * 1e3 / 1e3:5.2f} TB/s)  dgrad {g:6.1f} us  "
          f"wgrad+reduce {h:6.1f} us  wgrad (deferred reduce) {h2:6.1f} us")
