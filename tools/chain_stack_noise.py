"""MatchModule gradients: exact-fp32 MFMA vs bf16 layer modules vs bf16 row chains — is the chain inside the bf16 noise?"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
gr = importlib.import_module("3dvlp_amd.grounding")
ml = importlib.import_module("3dvlp_amd.mfma_linear")
rc = importlib.import_module("3dvlp_amd.row_chain")
torch.manual_seed(5)
B, L, K, C, T = 2, 4, 256, 128, 20
mm = gr.MatchModule(num_proposals=K, hidden_size=C).cuda().train()
for mod in mm.modules():
    if hasattr(mod, "fused_norm"):
        mod.fused_norm = True
    if isinstance(mod, torch.nn.Dropout):
        mod.p = 0.0
feats = torch.randn(B, K, C, device="cuda")
lang = torch.randn(B * L, T + 1, C, device="cuda")
gconf, gfeat = torch.randn(B * L, K, device="cuda"), torch.randn(B * L, K, C, device="cuda")


def run(bf, chain):
    rc.ENABLED = chain
    for mod in mm.modules():
        if hasattr(mod, "bf16_mma"):
            mod.bf16_mma = bf
    mm.zero_grad()
    x = feats.clone().requires_grad_()
    dd = {"bbox_feature": x, "input_ids": torch.zeros(B, L, T + 1), "istrain": [0], "lang_fea": lang}
    with ml.bf16_mma(bf):
        dd = mm(dd)
        ((dd["cluster_ref"] * gconf).sum() + (dd["cross_box_feature"] * gfeat).sum()).backward()
    return {n: p.grad.double().clone() for n, p in mm.named_parameters() if p.grad is not None}


ref, a, b = run(False, False), run(True, False), run(True, True)
rel = lambda u, v: float((u - v).norm() / (v.norm() + 1e-30))
for n in ref:
    if n.endswith("fc_k.bias"):
        continue
    print("%-60s modules-vs-fp32 %.2e  chain-vs-fp32 %.2e  chain-vs-modules %.2e" % (n, rel(a[n], ref[n]), rel(b[n], ref[n]), rel(b[n], a[n])))
