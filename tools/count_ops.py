"""Count ATen ops (~ kernel launches) of one eager step per phase: forward of each top-level module, loss, backward.
With `--where` every forward op is listed with the innermost source line inside the package (and its tensor shapes)."""
import importlib, os, sys, traceback
from collections import Counter, defaultdict
import torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gs = importlib.import_module("3dvlp_amd.grounding_step")
synth = importlib.import_module("3dvlp_amd.synth")
dev = torch.device("cuda:0")
step = gs.GroundingStep(dev, use_graph=False, pipeline=False, sa_dtype=torch.bfloat16)
batch = gs.batch_to_device(synth.make_batch(0, 8, 40000, 8), dev)
step.run(batch)
SKIP = ("aten.view", "aten._unsafe_view", "aten.t.", "aten.transpose", "aten.permute", "aten.expand", "aten.slice", "aten.select",
        "aten.unsqueeze", "aten.squeeze", "aten.detach", "aten.alias", "aten.as_strided", "aten.empty", "aten.reshape",
        "aten.split", "aten.unbind", "aten.chunk", "aten._reshape_alias", "aten.lift_fresh", "aten.narrow", "aten.stride", "aten.size",
        "aten.sym_", "aten.is_", "aten.unfold", "aten.diagonal", "aten.new_empty.", "aten.empty_like", "aten.result_type", "aten.item", "aten._local_scalar")
phase = ["init"]
WHERE = "--where" in sys.argv
where = Counter()
counts = defaultdict(Counter)
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        n = str(func)
        if not n.startswith(SKIP):
            counts[phase[0]][n] += 1
            if WHERE:
                fr = [f for f in traceback.extract_stack() if "3dvlp_amd/" in f.filename]
                loc = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "(autograd)"
                shp = ",".join(str(tuple(a.shape)) for a in args if torch.is_tensor(a))[:60]
                where[(phase[0], n, loc, shp)] += 1
        return func(*args, **(kwargs or {}))
def pre(name):
    def f(m, a): phase[0] = "fwd:" + name
    return f
def post(m, a, o): phase[0] = "fwd:glue"
for name, m in step.model.named_children():
    m.register_forward_pre_hook(pre(name)); m.register_forward_hook(post)
# as the pipelined step runs it: geometry prepared beforehand (side stream), bf16 linears, deferred counters / slab sums
geometry = step.model.backbone_net.compute_geometry(step._coords(batch))
_orig_loss = gs.grounding_loss
def _loss(*a, **k):
    phase[0] = "loss"
    return _orig_loss(*a, **k)
gs.grounding_loss = _loss
with Mode():
    step.bucket.zero()
    phase[0] = "fwd:glue"
    loss, d = step.forward_loss(batch, geometry)
    phase[0] = "backward"
    step._backward(loss)
    phase[0] = "collect+opt"
    step.bucket.collect(); step.opt.step()
torch.cuda.synchronize()
tot = 0
for ph, c in counts.items():
    n = sum(c.values()); tot += n
    print(f"== {ph}: {n} ops")
    for k, v in c.most_common(int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 12):
        print(f"     {v:4d}  {k}")
print("total", tot)
if WHERE:
    for (ph, n, loc, shp), v in sorted(where.items(), key=lambda kv: (kv[0][0], kv[0][2])):
        print(f"{v:3d} {ph:18s} {loc:28s} {n:38s} {shp}")
