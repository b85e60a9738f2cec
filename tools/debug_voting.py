import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
det = importlib.import_module("3dvlp_amd.detection")
torch.manual_seed(6)
B, S, C = 4, 1024, 256
for training in (False, True):
    mod = det.VotingModule(1, 256).cuda().train(training)
    with torch.no_grad():
        for m_ in mod.modules():
            if isinstance(m_, torch.nn.BatchNorm1d):
                m_.weight.uniform_(0.5, 1.5); m_.bias.uniform_(-0.3, 0.3)
                m_.running_mean.uniform_(-0.2, 0.2); m_.running_var.uniform_(0.5, 1.5)
        for n_, p in mod.named_parameters():
            if n_.endswith("bias"):
                p.uniform_(-0.3, 0.3)
    f0 = torch.randn(B, S, C, device="cuda")
    xyz = torch.rand(B, S, 3, device="cuda")
    g1, g2 = torch.randn(B, S, 3, device="cuda"), torch.randn(B, S, C, device="cuda")
    res = {}
    for fused in (True, False, "f64"):
        mod.zero_grad()
        if fused == "f64":
            P = {n_: p.detach().double().requires_grad_(True) for n_, p in mod.named_parameters()}
            fd = f0.double().requires_grad_(True)
            x = fd.reshape(B * S, C)
            for conv, bn in (("conv1", mod.bn1), ("conv2", mod.bn2), ("conv3", None)):
                y = x @ P[conv + ".weight"][:, :, 0].t() + P[conv + ".bias"]
                if bn is None:
                    x = y; break
                bnn = "bn1" if bn is mod.bn1 else "bn2"
                mean, var = (y.mean(0), y.var(0, unbiased=False)) if training else (bn.running_mean.double(), bn.running_var.double())
                x = torch.relu((y - mean) / torch.sqrt(var + bn.eps) * P[bnn + ".weight"] + P[bnn + ".bias"])
            net = x.view(B, S, 1, 3 + C)
            vx = (xyz.double().unsqueeze(2) + net[..., :3]).reshape(B, S, 3)
            vf = (fd.unsqueeze(2) + net[..., 3:]).reshape(B, S, C)
            ((vx * g1.double()).sum() + (vf * g2.double()).sum()).backward()
            res[fused] = dict({n_: p.grad for n_, p in P.items()}, f=fd.grad)
        else:
            import copy
            m2 = copy.deepcopy(mod)
            m2.fused = fused
            f = f0.clone().requires_grad_(True)
            vx, vf = m2(xyz, f.transpose(1, 2))
            ((vx * g1).sum() + (vf.transpose(1, 2) * g2).sum()).backward()
            res[fused] = dict({n_: p.grad.clone() for n_, p in m2.named_parameters()}, f=f.grad)
    print("training", training)
    for k in res["f64"]:
        r = res["f64"][k]
        e1 = ((res[True][k].double() - r).norm() / (r.norm() + 1e-30)).item()
        e2 = ((res[False][k].double() - r).norm() / (r.norm() + 1e-30)).item()
        print(f"   {k:14s} fused {e1:.2e}   literal {e2:.2e}   |ref| {r.norm().item():.3e}")
