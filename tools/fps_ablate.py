"""Timing experiments on the register-resident pruned FPS kernel (SA1 of cfg2, alone on the chip): the kernel with parts of
its iteration left out (the outputs of those runs are NOT the sampling; only the production run is checked).
    python tools/fps_ablate.py"""
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
dev = torch.device("cuda:0")
xyz = torch.from_numpy(np.stack([synth.make_scene(1000 + i, N)["xyz"] for i in range(B)])).to(dev)
nbytes = int(ext.load().vlp3d_fps_workspace_bytes(B, N))
ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
idx = torch.empty((B, m), dtype=torch.int32, device=dev)
ref = torch.empty_like(idx)
ext.call("vlp3d_fps_pruned_trace", xyz, B, N, m, ws, nbytes, ref, None, None, -1, 1)


def timed(variant, reps=5):
    f = lambda: ext.call("vlp3d_fps_pruned_trace", xyz, B, N, m, ws, nbytes, idx, None, None, -1, variant)
    f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        f()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / reps


rows = [(1, "round-3 kernel"), (0, "register-resident kernel"), (16 + 4, "  ... wave candidate reduced every iteration"),
        (16 + 2, "  ... no slot ever reduced again"), (16 + 1, "  ... no slot updates at all (fixed chain: test, candidate, barrier, block reduction)")]
for v, name in rows:
    t = timed(v)
    if v == 0:
        assert torch.equal(idx, ref), "register-resident kernel differs from the round-3 kernel"
    print(f"{t * 1e3:8.0f} us  {(t * 1e3 - 85) / (m - 1) * 2.4e3:6.0f} cycles/iteration at 2.4 GHz (85 us pre-pass taken off)  {name}", flush=True)
