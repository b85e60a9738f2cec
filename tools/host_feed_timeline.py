"""Where does the host-fed step spend its extra time?  Events on the launch stream around the refill and the graphs, host
clock around Prefetcher.next().   python tools/host_feed_timeline.py [nocompress]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gs = importlib.import_module("3dvlp_amd.grounding_step")
synth = importlib.import_module("3dvlp_amd.synth")
ip = importlib.import_module("3dvlp_amd.input_pipeline")
dev = torch.device("cuda:0")
copy_stream = torch.cuda.Stream()
step = gs.GroundingStep(dev, epoch=50, sa_dtype=torch.bfloat16, use_graph=True, pipeline=True)
with torch.cuda.stream(step._side):
    torch.zeros(1, device=dev)
with torch.cuda.stream(copy_stream):
    torch.zeros(1, device=dev)
host = []
for j in range(3):
    hb = {k: torch.from_numpy(v) for k, v in synth.make_batch(8 * j, 8, 40000, 8).items()}
    if "nocompress" not in sys.argv:
        hb = ip.compress_cloud(hb)
    host.append({k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in hb.items()})


def endless():
    i = 0
    while True:
        yield host[i % 3]
        i += 1


class TimedPrefetcher(ip.Prefetcher):
    """Prefetcher.preload with the host clock around it."""
    acc = {"preload": 0.0, "n": 0}

    def preload(self):
        t0 = time.perf_counter()
        super().preload()
        self.acc["preload"] += time.perf_counter() - t0
        self.acc["n"] += 1


feed = TimedPrefetcher(endless(), device=dev, prepare=gs.prepare_batch, stream=copy_stream)
cur, nxt = feed.next(), feed.next()
orig_refill, orig_replay = step._refill, step._replay
marks = []


def refill(static, batch):
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    orig_refill(static, batch)
    e1 = torch.cuda.Event(enable_timing=True); e1.record()
    marks.append(("refill", e0, e1))


def replay():
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    orig_replay()
    e1 = torch.cuda.Event(enable_timing=True); e1.record()
    marks.append(("graphs", e0, e1))


step._refill = refill
step._replay = replay
host_ms = {"run": 0.0, "next": 0.0}
steps = 60
for i in range(20 + steps):
    if i == 20:
        torch.cuda.synchronize()
        marks.clear()
        host_ms = {"run": 0.0, "next": 0.0}
        for k_ in TimedPrefetcher.acc:
            TimedPrefetcher.acc[k_] = 0
        t_all = time.perf_counter()
        first = torch.cuda.Event(enable_timing=True); first.record()
    t0 = time.perf_counter()
    step.run(cur, nxt)
    t1 = time.perf_counter()
    en0 = torch.cuda.Event(enable_timing=True); en0.record()
    cur, nxt = nxt, feed.next()
    en1 = torch.cuda.Event(enable_timing=True); en1.record()
    marks.append(("next()", en0, en1))
    t2 = time.perf_counter()
    host_ms["run"] += 1e3 * (t1 - t0)
    host_ms["next"] += 1e3 * (t2 - t1)
last = torch.cuda.Event(enable_timing=True); last.record()
torch.cuda.synchronize()
wall = 1e3 * (time.perf_counter() - t_all) / steps
print(f"ms/step {wall:.3f} (events {first.elapsed_time(last) / steps:.3f}); host per step: run {host_ms['run'] / steps:.3f}, next {host_ms['next'] / steps:.3f}")
agg = {}
for name, a, b in marks:
    agg.setdefault(name, []).append(a.elapsed_time(b))
for k, v in agg.items():
    print(f"  {k:8s} {len(v) / steps:4.1f} per step, {np.sum(v) / steps:.3f} ms per step on the launch stream (median {np.median(v):.3f})")
print("  graphs ms, last 24 steps:", " ".join(f"{x:.2f}" for x in agg["graphs"][-24:]))
print("  next() ms, last 24 steps:", " ".join(f"{x:.2f}" for x in agg["next()"][-24:]))
a = TimedPrefetcher.acc
print(f"  preload host ms per call: {1e3 * a['preload'] / max(a['n'], 1):.3f}")
print(f"  device memory reserved {torch.cuda.memory_reserved() / 2 ** 30:.2f} GiB, allocated {torch.cuda.memory_allocated() / 2 ** 30:.2f} GiB")
