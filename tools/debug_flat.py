import importlib, os, sys, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ddp = importlib.import_module("3dvlp_amd.ddp"); tr = importlib.import_module("3dvlp_amd.transformer")
torch.manual_seed(2)
net = torch.nn.ModuleDict({"att": tr.MultiHeadAttention(128, 32, 32, 4, dropout=0.0), "unused": torch.nn.Linear(8, 8)}).cuda()
ref = copy.deepcopy(net)
layout = ddp.FlatParams(net); bucket = ddp.FlatGradBucket(net, layout=layout)
opt = ddp.FlatAdamW(layout, bucket, lr=1e-2, weight_decay=0.1)
ropt = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=0.1)
x = torch.randn(4, 256, 128, device="cuda")
for step in range(3):
    bucket.zero(); net["att"](x, x, x).pow(2).mean().backward(); bucket.collect()
    ropt.zero_grad(set_to_none=True); ref["att"](x, x, x).pow(2).mean().backward()
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        if p.grad is not None:
            print(step, "grad", n, ((p.grad - q.grad).abs().max() / (q.grad.abs().max() + 1e-30)).item())
    opt.step(); ropt.step()
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        print(step, "param", n, (p - q).abs().max().item())
