"""Where does a gradient difference enter?  GPU fp32 step vs CpuStep fp64 / fp32 on 1 scene: gradients with respect to
the backbone's intermediate feature tensors, then per parameter.
    python tools/step_parity_detail.py [parameter prefix]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import baseline  # noqa: E402
from tests import test_step_parity as T  # noqa: E402

prefix = sys.argv[1] if len(sys.argv) > 1 else "backbone_net"
gs = importlib.import_module("3dvlp_amd.grounding_step")
ext = importlib.import_module("3dvlp_amd._lib")
synth = importlib.import_module("3dvlp_amd.synth")
batch_np = synth.make_batch(0, 1, num_points=40000, lang_num_max=8)
devc = torch.device("cuda:0")
step = gs.GroundingStep(devc, epoch=50, lr=1e-3)
T._dropout_off(step.model)
state = {n: p.detach().clone().cpu() for n, p in step.model.named_parameters()}
batch = gs.batch_to_device(batch_np, devc)
batch["random"] = torch.tensor(0.75, device=devc)
step.bucket.zero()
loss, d = step.forward_loss(batch)
KEYS = ("sa1_features", "sa2_features", "sa3_features", "sa4_features", "fp2_features", "vote_xyz", "vote_features",
        "aggregated_vote_xyz", "aggregated_vote_features")
for k in KEYS:
    d[k].retain_grad()
with ext.deferred_slab_reduce():
    loss.backward()
torch.cuda.synchronize()
gint = {k: d[k].grad.detach().cpu() for k in KEYS}
gact = {k: d[k].detach().cpu() for k in KEYS}
grads = {n: p.grad.detach().clone().cpu() for n, p in step.model.named_parameters() if p.grad is not None}
runs, ints, acts = {}, {}, {}
for name, dt in (("cpu64", torch.float64), ("cpu32", torch.float32)):
    cpu = baseline.CpuStep(lr=1e-3, dtype=dt)
    cpu.net.load_state_dict({k: v.to(dt) for k, v in state.items()}, strict=False)
    T._dropout_off(cpu.net)
    cpu.keep = {}
    cpu.step(baseline.to_torch(batch_np, 1, dt))
    runs[name] = {n: p.grad.detach().clone() for n, p in cpu.net.named_parameters() if p.grad is not None}
    ints[name] = {k: v.grad.detach().clone() for k, v in cpu.keep.items()}
    acts[name] = {k: v.detach().clone() for k, v in cpu.keep.items()}
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))
print("loss", float(loss))
for k in KEYS:
    print(f"act  {k:14s} gpu {rel(gact[k], acts['cpu64'][k]):9.2e}  cpu32 {rel(acts['cpu32'][k], acts['cpu64'][k]):9.2e}")
for k in KEYS:
    print(f"grad {k:14s} |g| {float(ints['cpu64'][k].norm()):9.3e} gpu {rel(gint[k], ints['cpu64'][k]):9.2e}  cpu32 {rel(ints['cpu32'][k], ints['cpu64'][k]):9.2e}")
for n, g in runs["cpu64"].items():
    if n.startswith(prefix):
        print(f"{n:60s} |g| {float(g.norm()):9.3e}  gpu {rel(grads[n], g):9.2e}  cpu32 {rel(runs['cpu32'][n], g):9.2e}")

# ---- the voting module in isolation: fp64 torch ops ON THE GPU, fed the step's own upstream gradients -------------------
vg = step.model.vgen
P = {n: p.detach().double().requires_grad_(True) for n, p in vg.named_parameters()}
sf = d["fp2_features"].detach().double().requires_grad_(True)      # (B,C,S)
sx = d["fp2_xyz"].detach().double()
B, C, S = sf.shape
x = sf.transpose(1, 2).reshape(B * S, C)
for i in (1, 2):
    y = x @ P[f"conv{i}.weight"][:, :, 0].t() + P[f"conv{i}.bias"]
    mean, var = y.mean(0), y.var(0, unbiased=False)
    x = torch.relu((y - mean) / torch.sqrt(var + 1e-5) * P[f"bn{i}.weight"] + P[f"bn{i}.bias"])
net = (x @ P["conv3.weight"][:, :, 0].t() + P["conv3.bias"]).view(B, S, 3 + C)
vx = sx + net[..., :3]
vf = sf.transpose(1, 2) + net[..., 3:]
vf = (vf / vf.norm(dim=-1, keepdim=True)).transpose(1, 2)            # (B,C,S)
gvx, gvf = d["vote_xyz"].grad.double(), d["vote_features"].grad.double()
print("vgen isolated: act vote_features", rel(d["vote_features"].detach().cpu(), vf.detach().cpu()))
((vx * gvx).sum() + (vf * gvf).sum()).backward()
for n, p in vg.named_parameters():
    if P[n].grad is not None and p.grad is not None:
        print(f"vgen isolated {n:14s} gpu-kernels vs fp64-same-upstream {rel(p.grad.cpu(), P[n].grad.cpu()):9.2e}")

print("column sums of d(vote_xyz):  gpu", gint["vote_xyz"].double().sum((0, 1)).tolist(), " cpu64", ints["cpu64"]["vote_xyz"].sum((0, 1)).tolist(),
      " cpu32", ints["cpu32"]["vote_xyz"].double().sum((0, 1)).tolist())
a, b, c32 = gint["vote_features"].double().sum((0, 2)), ints["cpu64"]["vote_features"].sum((0, 2)), ints["cpu32"]["vote_features"].double().sum((0, 2))
print("channel sums of d(vote_features): |cpu64|", float(b.norm()), " gpu err", float((a - b).norm()), " cpu32 err", float((c32 - b).norm()))
e = gint["vote_features"].double() - ints["cpu64"]["vote_features"]
print("error of d(vote_features): norm", float(e.norm()), " norm of its per-channel mean x sqrt(S)", float(e.mean((0, 2)).norm() * e.shape[2] ** 0.5))
top = e.abs().flatten().topk(5)
print("largest element errors", top.values.tolist(), " typical |g|", float(ints["cpu64"]["vote_features"].abs().mean()))
rows = e.pow(2).sum(1).sqrt()[0]
print("rows (votes) with the largest error:", rows.topk(5).indices.tolist(), rows.topk(5).values.tolist(), " median row error", float(rows.median()))


def vgen64(gvx_, gvf_, feats):
    P_ = {n: p.detach().double().requires_grad_(True) for n, p in vg.named_parameters()}
    sf_ = feats.double().cuda().requires_grad_(True)
    x_ = sf_.transpose(1, 2).reshape(B * S, C)
    for i in (1, 2):
        y_ = x_ @ P_[f"conv{i}.weight"][:, :, 0].t() + P_[f"conv{i}.bias"]
        x_ = torch.relu((y_ - y_.mean(0)) / torch.sqrt(y_.var(0, unbiased=False) + 1e-5) * P_[f"bn{i}.weight"] + P_[f"bn{i}.bias"])
    net_ = (x_ @ P_["conv3.weight"][:, :, 0].t() + P_["conv3.bias"]).view(B, S, 3 + C)
    vf_ = sf_.transpose(1, 2) + net_[..., 3:]
    vf_ = (vf_ / vf_.norm(dim=-1, keepdim=True)).transpose(1, 2)
    (((sx + net_[..., :3]) * gvx_.double().cuda()).sum() + (vf_ * gvf_.double().cuda()).sum()).backward()
    return {n: p.grad for n, p in P_.items()}, sf_.grad


fa = d["fp2_features"].detach()
ga, da = vgen64(gint["vote_xyz"], gint["vote_features"], fa)
gb, db = vgen64(ints["cpu64"]["vote_xyz"], ints["cpu64"]["vote_features"], fa)
gc, dc = vgen64(ints["cpu32"]["vote_xyz"], ints["cpu32"]["vote_features"], fa)
gd, dd = vgen64(ints["cpu64"]["vote_xyz"], ints["cpu64"]["vote_features"], acts["cpu64"]["fp2_features"])
for n in ("bn2.bias", "bn2.weight", "conv2.weight", "conv3.weight"):
    print(f"fp64 vgen, {n:13s}: gpu-upstream vs cpu64-upstream {rel(ga[n], gb[n]):9.2e}   cpu32-upstream vs cpu64-upstream "
          f"{rel(gc[n], gb[n]):9.2e}   gpu activations vs cpu64 activations (cpu64 upstream) {rel(gb[n], gd[n]):9.2e}")
print("d seed_features: gpu-up vs cpu64-up", rel(da, db), " cpu32-up vs cpu64-up", rel(dc, db), " acts", rel(db, dd))
e32 = ints["cpu32"]["vote_features"].double() - ints["cpu64"]["vote_features"]
r32 = e32.pow(2).sum(1).sqrt()[0]
print("cpu32 rows with the largest error:", r32.topk(5).indices.tolist(), r32.topk(5).values.tolist(), " median", float(r32.median()))
ex = gint["vote_xyz"].double() - ints["cpu64"]["vote_xyz"]
rx = ex.pow(2).sum(2).sqrt()[0]
print("d(vote_xyz) gpu rows with the largest error:", rx.topk(5).indices.tolist(), rx.topk(5).values.tolist(), " median", float(rx.median()))

for k in ("sa1_features", "sa2_features", "sa3_features", "sa4_features", "fp2_features"):
    for who, t in (("gpu", gact[k]), ("cpu32", acts["cpu32"][k])):
        e_ = (t.double() - acts["cpu64"][k])[0]                    # (C, n)
        r_ = e_.pow(2).sum(0).sqrt() / (acts["cpu64"][k][0].pow(2).sum(0).sqrt() + 1e-30)
        cm = e_.mean(1).norm() / acts["cpu64"][k][0].mean(1).norm()
        print(f"act {k:13s} {who:5s}: per-point rel error median {float(r_.median()):.2e} max {float(r_.max()):.2e}  "
              f"error of the per-channel MEAN over points {float(cm):.2e}")

base = acts["cpu64"]["fp2_features"]
for seed in range(6):
    g_ = torch.Generator().manual_seed(seed)
    pert = base * (1 + 7e-6 * torch.randn(base.shape, generator=g_, dtype=torch.float64))
    gp, dp = vgen64(ints["cpu64"]["vote_xyz"], ints["cpu64"]["vote_features"], pert)
    print(f"random 7e-6 perturbation of fp2_features, seed {seed}: bn2.bias {rel(gp['bn2.bias'], gd['bn2.bias']):.2e} "
          f"conv2.weight {rel(gp['conv2.weight'], gd['conv2.weight']):.2e} d seed_features {rel(dp, dd):.2e}")
