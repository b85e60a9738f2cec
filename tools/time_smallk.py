import importlib, sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
ext = importlib.import_module("3dvlp_amd._lib")
dev = torch.device("cuda:0")
x = torch.randn(2048, 27, device=dev); W = torch.randn(128, 27, device=dev); b = torch.randn(128, device=dev)
base = torch.randn(2048, 128, device=dev); out = torch.empty(2048, 128, device=dev)
xl = torch.randn(2048, 128, device=dev); Wl = torch.randn(128, 128, device=dev); yl = torch.empty(2048, 128, device=dev)
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n
print("smallk_fwd %.2f us" % t(lambda: ext.call("vlp3d_smallk_fwd", x, 27, W, b, base, 2048, 27, 128, out)))
print("smallk_fwd no base %.2f us" % t(lambda: ext.call("vlp3d_smallk_fwd", x, 27, W, b, None, 2048, 27, 128, out)))
print("linear_fwd 128x128 %.2f us" % t(lambda: ext.call("vlp3d_linear_fwd", xl, Wl, b, 2048, 128, 128, yl, 1)))
print("empty %.2f us" % t(lambda: ext.load().vlp3d_probe_empty(1, 64, None, ext._stream())))
for (K, N, R) in ((4, 128, 2048), (27, 64, 2048), (27, 256, 2048), (27, 128, 256), (27, 128, 16384), (32, 128, 2048)):
    x2 = torch.randn(R, K, device=dev); W2 = torch.randn(N, K, device=dev); b2 = torch.randn(N, device=dev); o2 = torch.empty(R, N, device=dev)
    print("smallk_fwd K=%d N=%d R=%d  %.2f us" % (K, N, R, t(lambda: ext.call("vlp3d_smallk_fwd", x2, K, W2, b2, None, R, K, N, o2))))
