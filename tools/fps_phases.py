"""Phase breakdown of the pruned FPS kernel's iteration (SA1 of cfg2: 8 x 40 000 -> 2048): shader-clock cycles that thread 0
of every workgroup spends in each phase, summed over the 2047 iterations (csrc/fps_pruned.hip, PROF instantiation), next to
the production kernel's duration.
    python tools/fps_phases.py"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ext = importlib.import_module("3dvlp_amd._lib")
synth = importlib.import_module("3dvlp_amd.synth")
B, N, m = 8, 40000, 2048
xyz = torch.from_numpy(np.stack([synth.make_scene(1000 + i, N)["xyz"] for i in range(B)])).cuda()
nbytes = int(ext.load().vlp3d_fps_workspace_bytes(B, N))
ws = torch.empty((nbytes,), dtype=torch.uint8, device="cuda")
idx = torch.empty((B, m), dtype=torch.int32, device="cuda")
idx2 = torch.empty_like(idx)
ph = torch.zeros((B, 8), dtype=torch.int64, device="cuda")


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / reps


t_prod = timed(lambda: ext.call("vlp3d_furthest_point_sampling_pruned", xyz, B, N, m, ws, nbytes, idx))
t_prof = timed(lambda: ext.call("vlp3d_fps_pruned_profile", xyz, B, N, m, ws, nbytes, idx2, ph))
assert torch.equal(idx, idx2)
c = ph.cpu().numpy()[:, :5].astype(np.float64)
tot = c.sum(1)
names = ["slot bounding-box test + ballot", "distance updates of the active slots", "wave candidate (reduce + readlanes)",
         "LDS write + workgroup barrier", "block reduction + next coordinates"]
print(f"production entry point (pre-pass + kernel) {t_prod * 1e3:.0f} us; profiled {t_prof * 1e3:.0f} us; "
      f"{m - 1} iterations -> {t_prod * 1e3 / (m - 1):.3f} us per iteration")
mean = c.mean(0)
for n_, v in zip(names, mean):
    print(f"  {n_:42s} {v / (m - 1):8.0f} cycles/iteration  {100 * v / mean.sum():5.1f} %")
print(f"  (thread 0 of each workgroup, mean over the {B} scenes; total {mean.sum() / (m - 1):.0f} cycles/iteration; "
      f"scene spread of the total {tot.min() / (m - 1):.0f} .. {tot.max() / (m - 1):.0f})")
