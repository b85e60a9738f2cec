"""Time the relation-bias kernels at the grounding shape (B=8, K=256)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
det = importlib.import_module("3dvlp_amd.detection")
ext = importlib.import_module("3dvlp_amd._lib")
dev = torch.device("cuda:0")
B, K = 8, 256
centre = torch.randn(B, K, 3, device=dev)
n = int(ext.load().vlp3d_relation_bias_nparam())
params = torch.randn(n, device=dev) * 0.2
out = torch.empty(B, 4, K, K, device=dev)
dout = torch.randn_like(out)
dpar = torch.empty_like(params)
nb = det._RelationBias.SLAB_BLOCKS
slabs = torch.empty(nb, n, device=dev)
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps * 1e3
print("fwd %.1f us" % t(lambda: ext.call("vlp3d_relation_bias_fwd", centre, params, B, K, out)))
print("bwd %.1f us" % t(lambda: ext.call("vlp3d_relation_bias_bwd", centre, params, dout, B, K, dpar, slabs, nb, 0)))
