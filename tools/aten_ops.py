"""Which aten ops (= framework kernels) does one step issue, and from which source line of the package?  A TorchDispatchMode
logs every non-view aten call of one eager bf16 step with the innermost 3dvlp_amd frame (autograd-engine ops have no Python
frame: they are attributed to the backward Function that is running, when there is one).
    python tools/aten_ops.py"""
import collections
import importlib
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
gs = importlib.import_module("3dvlp_amd.grounding_step")
synth = importlib.import_module("3dvlp_amd.synth")

VIEWS = {"view", "reshape", "_unsafe_view", "t", "transpose", "permute", "expand", "slice", "select", "unsqueeze", "squeeze",
         "as_strided", "detach", "alias", "split", "split_with_sizes", "unbind", "_reshape_alias", "empty", "empty_like",
         "empty_strided", "new_empty", "new_empty_strided", "set_", "is_pinned", "_local_scalar_dense", "lift_fresh", "unfold",
         "narrow", "chunk", "stride", "sym_size", "sym_stride", "sym_numel", "is_contiguous", "dim", "size", "storage_offset"}


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.cnt = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        if name not in VIEWS:
            where = "(autograd engine)"
            for fr in reversed(traceback.extract_stack()):
                if "3dvlp_amd/" in fr.filename and "tools/" not in fr.filename:
                    where = f"{fr.filename.split('3dvlp_amd/')[-1]}:{fr.lineno} {fr.name}"
                    break
            shapes = [tuple(a.shape) for a in args if torch.is_tensor(a)][:2]
            self.cnt[(where, name, str(shapes))] += 1
        return func(*args, **(kwargs or {}))


dev = torch.device("cuda:0")
step = gs.GroundingStep(dev, sa_dtype=torch.bfloat16, use_graph=False, pipeline=True)
batch = gs.batch_to_device(synth.make_batch(0, 8, 40000, 8), dev)
for _ in range(3):
    step.run(batch)
torch.cuda.synchronize()
log = Log()
with log:
    step.run(batch)
torch.cuda.synchronize()
print("aten calls (non-view):", sum(log.cnt.values()))
for (where, name, shapes), n in sorted(log.cnt.items()):
    print(f"{n:3d}x  {where:60s} {name:28s} {shapes}")
