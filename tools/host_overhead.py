"""Where does the host spend its time in one step? (graph replay enqueue vs optimizer vs GPU wait)"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("3dvlp_amd.synth")
gs = importlib.import_module("3dvlp_amd.grounding_step")
dev = torch.device("cuda:0")
batch = gs.batch_to_device(synth.make_batch(0, 8, 40000, 8), dev)
import sys as _s; PIPE = "--no-pipeline" not in _s.argv
step = gs.GroundingStep(dev, sa_dtype=torch.bfloat16, use_graph=True, pipeline=PIPE)
for _ in range(4):
    step.run(batch)
torch.cuda.synchronize()
t_replay = t_opt = t_total = 0.0
n = 20
t00 = time.perf_counter()
for _ in range(n):
    t0 = time.perf_counter()
    step._replay()
    t1 = time.perf_counter()
    step.bucket.all_reduce()
    step.opt.step()
    t2 = time.perf_counter()
    t_replay += t1 - t0
    t_opt += t2 - t1
torch.cuda.synchronize()
t_total = time.perf_counter() - t00
print(f"per step: host in graph.replay() {1e3*t_replay/n:.2f} ms, host in opt.step() {1e3*t_opt/n:.2f} ms, wall {1e3*t_total/n:.2f} ms")
# GPU-only time of a replay (events)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
s.record()
for _ in range(n):
    step._replay()
e.record()
torch.cuda.synchronize()
print(f"replay only (no optimizer), GPU events: {s.elapsed_time(e)/n:.2f} ms per replay")
s.record()
for _ in range(n):
    step.opt.step()
e.record()
torch.cuda.synchronize()
print(f"optimizer only: {s.elapsed_time(e)/n:.2f} ms per step")
