"""Bytes of weight-gradient slabs written (and read again) per step at the bench configuration, by entry."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
gs = importlib.import_module("3dvlp_amd.grounding_step")
synth = importlib.import_module("3dvlp_amd.synth")
ext = importlib.import_module("3dvlp_amd._lib")
dev = torch.device("cuda:0")
step = gs.GroundingStep(dev, sa_dtype=torch.bfloat16)
batch = gs.batch_to_device(synth.make_batch(0, 8, 40000, 8), dev)
log = []
orig = ext.SlabReduceQueue.add
def add(self, partials, nblk, dst, n_mat, K, ldo, dbias=None, n_bias=0, ncol_out=0, rot=0):
    log.append((nblk, n_mat // K, K, 4 * nblk * (n_mat + n_bias)))
    return orig(self, partials, nblk, dst, n_mat, K, ldo, dbias, n_bias, ncol_out, rot)
ext.SlabReduceQueue.add = add
step.run(batch)
torch.cuda.synchronize()
tot = sum(b for *_, b in log)
for nblk, N, K, b in log:
    print(f"nblk {nblk:5d}  N {N:4d} K {K:4d}  {b / 1e6:8.2f} MB")
print("entries", len(log), "total MB", tot / 1e6)
