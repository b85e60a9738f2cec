import importlib, torch, numpy as np, sys
sys.path.insert(0, "/root/repo")
cap = importlib.import_module("3dvlp_amd.caption")
ml = importlib.import_module("3dvlp_amd.mfma_linear")
torch.manual_seed(0)
for (R, V, bf) in [(4, 1000, False), (66, 1000, False), (66, 1000, True), (1984, 30522, True)]:
    x = torch.randn(R, 128, device="cuda") * 0.7
    W = torch.randn(V, 128, device="cuda") * 0.2
    b = torch.randn(V, device="cuda") * 0.3
    tgt = torch.randint(0, V, (R,), device="cuda")
    with ml.bf16_mma(bf):
        nll, arg = cap.vocab_nll(x, W, b, tgt)
    torch.cuda.synchronize()
    logits = x.double() @ W.double().t() + b.double()
    want = torch.nn.functional.cross_entropy(logits, tgt, reduction="none")
    print(R, V, bf, "nll err", float((nll.double() - want).abs().max()), "argmax ok", float((arg.long() == logits.argmax(-1)).float().mean()), arg[:6].tolist(), nll[:4].tolist(), want[:4].tolist(), flush=True)
