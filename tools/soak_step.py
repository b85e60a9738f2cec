"""Soak of the captured, pipelined bf16 step: three alternating batches, N steps, every loss and the final parameters finite,
the loss of each batch lower at the end than at the start.   python tools/soak_step.py [steps]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
gs = importlib.import_module("3dvlp_amd.grounding_step")
synth = importlib.import_module("3dvlp_amd.synth")
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
batches = [gs.batch_to_device(synth.make_batch(8 * i, 8, 40000, 8), dev) for i in range(3)]
step = gs.GroundingStep(dev, epoch=50, sa_dtype=torch.bfloat16, use_graph=True, pipeline=True, lr=2e-4)
losses = []
for i in range(steps):
    losses.append(step.run(batches[i % 3], batches[(i + 1) % 3]).clone())
torch.cuda.synchronize()
ls = torch.stack(losses).float().cpu()
assert torch.isfinite(ls).all(), "non-finite loss at step %d" % int((~torch.isfinite(ls)).nonzero()[0])
flat = torch.cat([p.detach().reshape(-1) for p in step.model.parameters()])
assert torch.isfinite(flat).all()
for b in range(3):
    first, last = ls[b:30:3].mean(), ls[-30 + b::3].mean()
    print(f"batch {b}: mean loss of its first 10 visits {float(first):.3f} -> last 10 visits {float(last):.3f}")
    assert last < first
print("soak ok:", steps, "steps")
