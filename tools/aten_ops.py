"""List the framework (ATen / runtime-copy) launches of ONE eager training step with the operator and the autograd node that
issued each: the launches the step driver's graphs replay besides this library's own kernels.  Usage (GPU box):
python tools/aten_ops.py > gpurun_out/aten_ops.txt"""
import importlib
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gs = importlib.import_module("3dvlp_amd.grounding_step")
synth = importlib.import_module("3dvlp_amd.synth")

dev = torch.device("cuda:0")
step = gs.GroundingStep(dev, epoch=50, lr=1e-3, sa_dtype=torch.bfloat16, use_graph=False, pipeline=True, seed=0)
batch = gs.batch_to_device(synth.make_batch(0, 8, 40000, 8), dev)
for _ in range(2):
    step.run(batch, batch)
torch.cuda.synchronize()
split = len(sys.argv) > 1 and sys.argv[1] == "split"   # the second pass of the step driver's split backward only
if split:
    _ext = importlib.import_module("3dvlp_amd._lib")
    step.bucket.zero()
    loss, out = step.forward_loss(batch, step.model.backbone_net.compute_geometry(step._coords(batch)))
    boundary = [out["sa2_features"]]
    head = [p for n, p in step.model.named_parameters()
            if p.requires_grad and not n.startswith(("backbone_net.sa1.", "backbone_net.sa2."))]
    with _ext.deferred_slab_reduce():
        torch.autograd.backward([loss], inputs=head + boundary, retain_graph=True)
        g = boundary[0].grad
        print("sa2_features", tuple(boundary[0].shape), boundary[0].stride(), "grad", tuple(g.shape), g.stride())
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            torch.autograd.backward(boundary, [g])
            torch.cuda.synchronize()
else:
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step.run(batch, batch)
        torch.cuda.synchronize()


def chain(e):
    names = []
    while e is not None:
        names.append(e.name)
        e = e.cpu_parent
    return names


n = 0
for e in sorted(prof.events(), key=lambda e: e.time_range.start):
    if not e.kernels:
        continue
    ks = [k.name for k in e.kernels]
    if not any(("at::native" in k or "rocclr" in k or "Memcpy" in k or "Memset" in k) for k in ks):
        continue
    if any(c.kernels for c in e.cpu_children):   # report the innermost operator only
        continue
    ch = chain(e)
    node = next((c for c in ch if "evaluate_function" in c or c.endswith("Backward") or "Function" in c), "")
    n += 1
    print(f"{n:3d} {e.name:28s} shapes={str(e.input_shapes)[:70]:70s} <- {' < '.join(ch[1:4])[:110]}  [{node[:60]}]")
