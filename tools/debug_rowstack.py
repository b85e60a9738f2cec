"""Stage-by-stage check of row_mlp.row_stack against torch autograd on the same row-major formulas."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rm = importlib.import_module("3dvlp_amd.row_mlp")
torch.manual_seed(0)
TRAIN = "--eval" not in sys.argv
dev = "cuda"
for (R, dims, last_plain) in [(4096, [512, 256, 256], False), (8192, [256, 256, 256, 259], True), (2048, [128, 128, 128, 28], True),
                              (4096, [256, 256], False), (4096, [256, 128, 256], False)]:
    L = len(dims) - 1
    Ws = [torch.randn(dims[i + 1], dims[i], device=dev) * 0.1 for i in range(L)]
    bs = [torch.randn(dims[i + 1], device=dev) * 0.1 for i in range(L)]
    bns = [None if (last_plain and i == L - 1) else torch.nn.BatchNorm1d(dims[i + 1]).to(dev).train(TRAIN) for i in range(L)]
    for bn in bns:
        if bn is not None:
            with torch.no_grad():
                bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3); bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.5, 1.5)
    x0 = torch.randn(R, dims[0], device=dev)
    go = torch.randn(R, dims[-1], device=dev)
    out = []
    for impl in ("rows", "torch"):
        x = x0.clone().requires_grad_(True)
        W = [w.clone().requires_grad_(True) for w in Ws]
        b = [t.clone().requires_grad_(True) for t in bs]
        for bn in bns:
            if bn is not None:
                bn.weight.grad = bn.bias.grad = None
        if impl == "rows":
            y = rm.row_stack(x, [(W[i], b[i], bns[i]) for i in range(L)])
        else:
            y = x
            for i in range(L):
                y = torch.nn.functional.linear(y, W[i], b[i])
                if bns[i] is not None:
                    y = torch.relu(bns[i](y))
        (y * go).sum().backward()
        out.append([y.detach(), x.grad] + [w.grad for w in W] + [t.grad for t in b] +
                   [bn.weight.grad.clone() for bn in bns if bn is not None] + [bn.bias.grad.clone() for bn in bns if bn is not None])
    names = ["y", "dx"] + [f"dW{i}" for i in range(L)] + [f"db{i}" for i in range(L)] + ["dgamma"] * sum(b is not None for b in bns) + ["dbeta"] * sum(b is not None for b in bns)
    print(R, dims, "plain-last" if last_plain else "bn-last")
    for n, a, c in zip(names, out[0], out[1]):
        err = (a - c).abs().max().item() / (c.abs().max().item() + 1e-12)
        print(f"   {n:8s} rel err {err:.2e}  scale {c.abs().max().item():.3e}")
