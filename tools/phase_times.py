"""Event-timed phases of the replayed step (no profiler): when the main stream reaches the backward cut, when the side stream
finishes the next batch's geometry, how long the deferred graph and the SA2 / SA1 backward take beside each other.
Usage (GPU box): python tools/phase_times.py [steps]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gs = importlib.import_module("3dvlp_amd.grounding_step")
synth = importlib.import_module("3dvlp_amd.synth")
ip = importlib.import_module("3dvlp_amd.input_pipeline")

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
step = gs.GroundingStep(dev, epoch=50, lr=1e-3, sa_dtype=torch.bfloat16, use_graph=True, pipeline=True, seed=0)
batch = gs.batch_to_device(synth.make_batch(0, 8, 40000, 8), dev, feat_bf16=True)
for _ in range(5):
    step.run(batch)
torch.cuda.synchronize()
ev = lambda: torch.cuda.Event(enable_timing=True)
rec = []


def replay():
    cur = torch.cuda.current_stream()
    side = step._side
    e = {k: ev() for k in ("t0", "gS_end", "gM_end", "gD_end", "gM2_end")}
    e["t0"].record(cur)
    step._gC.replay()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        step._gS.replay()
        e["gS_end"].record(side)
    step._geom_for = step._next_src_tag
    step._gM.replay()
    e["gM_end"].record(cur)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        step._gD.replay()
        e["gD_end"].record(side)
    step._gM2.replay()
    e["gM2_end"].record(cur)
    cur.wait_stream(side)
    rec.append(e)


step._replay = replay
for _ in range(steps):
    step.run(batch)
torch.cuda.synchronize()
rows = [[e["t0"].elapsed_time(e[k]) for k in ("gS_end", "gM_end", "gD_end", "gM2_end")] for e in rec[5:]]
med = [sorted(c)[len(c) // 2] for c in zip(*rows)]
print("ms from the step's start (median of %d steps): geometry of the next batch done %.3f | main at the backward cut %.3f | "
      "deferred graph done %.3f | SA2 / SA1 backward done %.3f" % ((len(rows),) + tuple(med)))
wall = [rec[i]["t0"].elapsed_time(rec[i + 1]["t0"]) for i in range(5, len(rec) - 1)]
print("step to step %.3f ms (median)" % sorted(wall)[len(wall) // 2])
