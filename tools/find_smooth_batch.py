"""Which synthetic batches have NO ReLU / max-pool kink within fp32 round-off of the initial model's activations?
For each candidate: CpuStep (fp64) at the batch and at two copies whose input features are perturbed by 1e-6 (relative);
prints the largest per-block gradient difference.  A smooth batch (diff < 1e-4) makes tests/test_step_parity.py a tight
comparison; at a kink the gradient is discontinuous and no two fp32 evaluations agree better than ~1e-3."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import baseline  # noqa: E402
from tests import test_step_parity as T  # noqa: E402

synth = importlib.import_module("3dvlp_amd.synth")
torch.set_num_threads(16)
scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for first in range(0, int(sys.argv[2]) if len(sys.argv) > 2 else 24, scenes):
    batch_np = synth.make_batch(first, scenes, num_points=40000, lang_num_max=8)
    grads = []
    for trial in range(3):
        b = dict(batch_np)
        if trial:
            rng = np.random.default_rng(trial)
            pc = b["point_clouds"].astype(np.float64)
            pc[..., 3:] *= 1 + 1e-6 * rng.standard_normal(pc[..., 3:].shape)
            b["point_clouds"] = pc
        cpu = baseline.CpuStep(lr=1e-3, dtype=torch.float64)
        T._dropout_off(cpu.net)
        cpu.step(baseline.to_torch(b, scenes, torch.float64))
        grads.append({n: p.grad.detach().clone() for n, p in cpu.net.named_parameters() if p.grad is not None})
    e1, e2 = T._per_block(grads[1], grads[0]), T._per_block(grads[2], grads[0])
    worst = max(max(e1.values()), max(e2.values()))
    print(f"first_scene {first:3d}: worst block diff {worst:.2e}   " +
          " ".join(f"{k.split('.')[-1]}={max(e1[k], e2[k]):.0e}" for k in e1 if max(e1[k], e2[k]) > 0), flush=True)
