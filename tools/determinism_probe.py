"""Run the captured step twice from the same state and report which gradients differ bit-wise between the two replays
(atomic float accumulation is allowed to; anything else would be a race).  python tools/determinism_probe.py [fp32]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gs = importlib.import_module("3dvlp_amd.grounding_step")
synth = importlib.import_module("3dvlp_amd.synth")
dev = torch.device("cuda:0")
dt = None if (len(sys.argv) > 1 and sys.argv[1] == "fp32") else torch.bfloat16
mode = dict(eager=dict(), pipe=dict(pipeline=True), graph1=dict(use_graph=True)).get(sys.argv[2] if len(sys.argv) > 2 else "", dict(use_graph=True, pipeline=True))
print("mode", mode, "dtype", dt)
step = gs.GroundingStep(dev, epoch=50, lr=0.0, sa_dtype=dt, **mode, seed=0)
batch = gs.batch_to_device(synth.make_batch(0, 8, 40000, 8), dev)
add_norm = importlib.import_module("3dvlp_amd.add_norm")
runs = []
for r in range(4):
    add_norm.state(dev).fill_(1234567)      # dropout masks of the add & norm kernels
    torch.manual_seed(0)                    # the copy-paste module's coin (torch.rand: the generator's offset also feeds a replay)
    torch.cuda.manual_seed_all(0)
    loss = step.run(batch, batch)
    torch.cuda.synchronize()
    runs.append((float(loss), step.bucket.flat.clone()))
names = [n for n, p in step.model.named_parameters() if p.requires_grad]
params = [p for n, p in step.model.named_parameters() if p.requires_grad]
print("losses", [f"{l:.9g}" for l, _ in runs])
a = runs[1][1]
for k in (2, 3):
    b = runs[k][1]
    print(f"run 1 vs run {k}: flat buffers equal = {torch.equal(a, b)}, max abs diff {float((a - b).abs().max()):.3e}")
off = 0
views = step.bucket.views
for n, v in zip([n for n, p in zip(names, params)], views):
    pass
bad = []
for (n, p), v in zip(step.model.named_parameters(), step.bucket.views) if len(step.bucket.views) == len(list(step.model.parameters())) else []:
    pass
# per-parameter report through the bucket's own views
for p, v in zip(step.bucket.params, step.bucket.views):
    o, m = v.storage_offset(), v.numel()
    x, y = runs[1][1][o:o + m], runs[3][1][o:o + m]
    if not torch.equal(x, y):
        name = next(n for n, q in step.model.named_parameters() if q is p)
        bad.append((name, float((x - y).abs().max()), float(x.abs().max())))
print(f"{len(bad)} of {len(step.bucket.params)} parameters differ between two replays")
for b in bad:
    print("  %-70s max diff %.3e  (max |g| %.3e)" % b)
