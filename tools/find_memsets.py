"""Which ops of one eager step call hipMemset*/hipMemcpy* (these become memset/memcpy NODES under graph capture)."""
import importlib, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
gs = importlib.import_module("3dvlp_amd.grounding_step")
synth = importlib.import_module("3dvlp_amd.synth")
dev = torch.device("cuda:0")
step = gs.GroundingStep(dev, sa_dtype=torch.bfloat16, use_graph=False, pipeline=False)
batch = gs.batch_to_device(synth.make_batch(0, 8, 40000, 8), dev)
for _ in range(2):
    step.run(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step.run(batch)
    torch.cuda.synchronize()
evs = prof.events()
cnt = collections.Counter()
for e in evs:
    if e.name.startswith("hipMemset") or e.name.startswith("hipMemcpy"):
        chain, p = [], e.cpu_parent
        while p is not None and len(chain) < 4:
            chain.append(p.name); p = p.cpu_parent
        cnt[(e.name, " < ".join(chain))] += 1
for k, v in cnt.most_common():
    print(v, k[0], "|", k[1][:200], flush=True)
