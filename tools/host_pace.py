"""Is the replayed step host-bound?  Host time of step.run() (no sync) against the wall time per step, and the host time of
each graph replay inside it."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
gs = importlib.import_module("3dvlp_amd.grounding_step")
synth = importlib.import_module("3dvlp_amd.synth")
dev = torch.device("cuda:0")
batch = gs.batch_to_device(synth.make_batch(0, 8, num_points=40000, lang_num_max=8), dev)
step = gs.GroundingStep(dev, epoch=50, sa_dtype=torch.bfloat16, use_graph=True, pipeline=True)
for _ in range(5):
    step.run(batch)
torch.cuda.synchronize()
n = 200
host = []
t0 = time.perf_counter()
for _ in range(n):
    a = time.perf_counter()
    step.run(batch)
    host.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
host.sort()
print("wall/step %.3f ms; host enqueue loop %.3f ms/step (median call %.3f, p90 %.3f, max %.3f); drain after loop %.2f ms"
      % ((t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3, host[n // 2] * 1e3, host[int(n * 0.9)] * 1e3, host[-1] * 1e3, (t2 - t1) * 1e3))
for name in ("_gC", "_gS", "_gM", "_gD", "_gM2"):
    g = getattr(step, name)
    torch.cuda.synchronize()
    a = time.perf_counter()
    g.replay()
    b = time.perf_counter()
    torch.cuda.synchronize()
    print("%-4s replay host %.3f ms" % (name, (b - a) * 1e3))
