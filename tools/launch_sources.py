"""Where do the framework (non-library) launches of one step come from?  One eager step of the bench configuration under
torch.profiler with Python stacks: every device kernel that is not one of csrc/'s, grouped by (kernel, aten op, innermost
source line inside the package)."""
import importlib, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
gs = importlib.import_module("3dvlp_amd.grounding_step")
synth = importlib.import_module("3dvlp_amd.synth")
dev = torch.device("cuda:0")
step = gs.GroundingStep(dev, sa_dtype=torch.bfloat16, use_graph=False, pipeline=True)
batch = gs.batch_to_device(synth.make_batch(0, 8, 40000, 8), dev)
for _ in range(3):
    step.run(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step.run(batch)
    torch.cuda.synchronize()
cnt = collections.Counter(); dur = collections.Counter()
total = 0
for e in prof.events():
    if not getattr(e, "kernels", None):
        continue
    # only the innermost op that owns the kernels
    for k in e.kernels:
        total += 1
        if "anonymous namespace)::" in k.name and "at::native" not in k.name:
            continue  # ours
        where = "?"
        p = e
        while p is not None and where == "?":
            for fr in (p.stack or []):
                if "3dvlp_amd/" in fr:
                    where = fr.split("3dvlp_amd/")[-1]
                    break
            p = p.cpu_parent
        top = e
        chain = [e.name]
        while top.cpu_parent is not None and len(chain) < 3:
            top = top.cpu_parent; chain.append(top.name)
        key = (k.name[:60], " < ".join(chain)[:90], where[:70])
        cnt[key] += 1; dur[key] += k.duration
print("device kernels in the step:", total, " framework:", sum(cnt.values()))
for key, n in sorted(cnt.items(), key=lambda kv: (kv[0][2], kv[0][1])):
    print(f"{n:3d}x {dur[key]:7.1f}us | {key[2]:70s} | {key[1]:90s} | {key[0]}")
