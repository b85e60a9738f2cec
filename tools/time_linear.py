"""Time the MFMA linear kernels at the transformer shapes of the grounding step."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ext = importlib.import_module("3dvlp_amd._lib")
dev = torch.device("cuda:0")
BF = int(sys.argv[1]) if len(sys.argv) > 1 else 0
def t(fn, reps=30):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for R, K, N in ((16384, 128, 128), (16384, 128, 256), (16384, 256, 128), (2048, 128, 128)):
    x = torch.randn(R, K, device=dev); w = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev)
    y = torch.empty(R, N, device=dev); dy = torch.randn(R, N, device=dev); dx = torch.empty(R, K, device=dev)
    nblk = max(16, min(256, R // 64)); dwb = torch.empty(N * K + N, device=dev); part = torch.empty(nblk, N * K + N, device=dev)
    f = t(lambda: ext.call("vlp3d_linear_fwd", x, w, b, R, K, N, y, BF))
    g = t(lambda: ext.call("vlp3d_linear_dgrad", dy, w, R, N, K, dx, None, BF))
    h = t(lambda: ext.call("vlp3d_linear_wgrad", dy, x, R, K, N, dwb, part, nblk, 1, 0, BF))
    print(f"R={R} K={K} N={N}: fwd {f:6.1f} us  dgrad {g:6.1f} us  wgrad+reduce {h:6.1f} us")
