import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ext = importlib.import_module("3dvlp_amd._lib")
BF = int(sys.argv[1]) if len(sys.argv) > 1 else 0
def t(fn, n=20):
    """n back-to-back launches replayed from a captured graph (from Python one by one these kernels are host bound)"""
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
for R, K, N in [(16384, 128, 128), (16384, 128, 384), (16384, 128, 256), (16384, 256, 128), (2048, 128, 128), (3136, 128, 256), (64, 128, 128)]:
    x = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1; b = torch.randn(N, device="cuda")
    y1 = torch.empty(R, N, device="cuda"); y2 = torch.empty(R, N, device="cuda")
    dy = torch.randn(R, N, device="cuda"); dx1 = torch.empty(R, K, device="cuda"); dx2 = torch.empty(R, K, device="cuda")
    a = t(lambda: ext.call("vlp3d_linear_fwd", x, w, b, R, K, N, y1, BF))
    c = t(lambda: ext.call("vlp3d_rows_fwd", x, K, R, K, None, w, b, N, y2, N, None, BF))
    d = t(lambda: ext.call("vlp3d_linear_dgrad", dy, w, R, N, K, dx1, None, BF))
    e = t(lambda: ext.call("vlp3d_rows_dgrad", dy, None, N, None, w, R, N, K, None, 0, None, dx2, K, None, BF))
    print(f"R={R} K={K} N={N}: fwd linear {a:.1f} us rows {c:.1f} us (maxdiff {(y1-y2).abs().max().item():.1e}) | dgrad linear {d:.1f} rows {e:.1f} ({(dx1-dx2).abs().max().item():.1e})")
