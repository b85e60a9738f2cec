"""Relation-bias backward: parameter-gradient error of the exact-fp32 and the bf16-MFMA form against fp64, per parameter."""
import copy, importlib, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
det = importlib.import_module("3dvlp_amd.detection")
ml = importlib.import_module("3dvlp_amd.mfma_linear")
torch.manual_seed(4)
for (B, K) in ((3, 70), (8, 256)):
    m = det.RelationModule(num_proposals=K, det_channel=128).cuda()
    fc = m.self_attn_fc[1]
    with torch.no_grad():
        for p in fc.parameters():
            p.add_(0.05 * torch.randn_like(p))
    centre = torch.rand(B, K, 3, device="cuda") * 4
    fc64 = copy.deepcopy(fc).double()
    c64 = centre.double()
    delta = c64[:, None, :, :] - c64[:, :, None, :]
    pair = torch.cat([delta, delta.pow(2).sum(-1, keepdim=True).sqrt()], dim=-1)
    ref = fc64(pair).permute(0, 3, 1, 2)
    g = torch.randn(B, 4, K, K, device="cuda")
    exp = torch.autograd.grad(ref, list(fc64.parameters()), g.double())
    for mode in (False, True):
        with ml.bf16_mma(mode):
            out = det.relation_bias(centre, fc)
        got = torch.autograd.grad(out, list(fc.parameters()), g)
        print(f"B={B} K={K} bf16={mode}:", "  ".join(f"{n}:{float((a.double() - b).abs().max() / b.abs().max()):.1e}/{float((a.double() - b).norm() / b.norm()):.1e}"
                                                       for (n, _), a, b in zip(fc.named_parameters(), got, exp)))
