import importlib, os, sys, copy
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pm = importlib.import_module("3dvlp_amd.pointnet2_modules")
pu = importlib.import_module("3dvlp_amd.pointnet2_utils")
torch.manual_seed(4)
B, n, m = 4, 1024, 512
unknown = torch.rand(B, n, 3, device="cuda") * 4
known = unknown[:, torch.randperm(n)[:m]].contiguous() + 0.01
fp = pm.PointnetFPModule(mlp=[512, 256, 256]).cuda().train()
with torch.no_grad():
    for mod in fp.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.uniform_(0.5, 1.5); mod.bias.uniform_(-0.3, 0.3)
            mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.5, 1.5)
ref = copy.deepcopy(fp); ref.fused = False
uf0, kf0 = torch.randn(B, n, 256, device="cuda"), torch.randn(B, m, 256, device="cuda")
g = torch.randn(B, 256, n, device="cuda")
res = {}
for name in ("fused", "literal", "rows_torch"):
    uf = uf0.clone().requires_grad_(True); kf = kf0.clone().requires_grad_(True)
    mod = fp if name == "fused" else ref
    mod.zero_grad()
    if name == "rows_torch":
        idx, w = pm.PointnetFPModule.compute_geometry(unknown, known)
        kfl = kf.reshape(B * m, 256)
        base = (torch.arange(B, device="cuda") * m)[:, None, None]
        gi = (idx.long() + base).reshape(-1, 3)
        interp = (kfl[gi] * w.reshape(-1, 3, 1)).sum(1)
        x = torch.cat([interp, uf.reshape(B * n, 256)], 1)
        for layer in mod.mlp:
            x = torch.nn.functional.linear(x, layer.conv.weight[:, :, 0, 0])
            bn = layer.bn.bn
            x = torch.relu(torch.nn.functional.batch_norm(x, None, None, bn.weight, bn.bias, True, 0.0, bn.eps))
        out = x.view(B, n, 256).transpose(1, 2)
    else:
        out = mod(unknown, known, uf.transpose(1, 2), kf.transpose(1, 2))
    (out * g).sum().backward()
    res[name] = (out.detach(), uf.grad, kf.grad, [p.grad.clone() for p in mod.parameters()])
for a in ("fused", "literal"):
    r, t = res[a], res["rows_torch"]
    print(a, "vs rows_torch: out", (r[0] - t[0]).abs().max().item(), "duf", (r[1] - t[1]).abs().max().item(), "/", t[1].abs().max().item(),
          "dkf", (r[2] - t[2]).abs().max().item(), "/", t[2].abs().max().item(),
          "params", [((x - y).abs().max() / y.abs().max()).item() for x, y in zip(r[3], t[3])])
