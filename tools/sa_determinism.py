"""Is the fused SA stack (bf16, compact rows) bitwise reproducible on fixed inputs?  backbone only, two passes."""
import importlib, sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
gs = importlib.import_module("3dvlp_amd.grounding_step")
synth = importlib.import_module("3dvlp_amd.synth")
devc = torch.device("cuda:0")
batch = gs.batch_to_device(synth.make_batch(0, 2, num_points=8192, lang_num_max=2), devc)
step = gs.GroundingStep(devc, sa_dtype=torch.bfloat16)
bb = step.model.backbone_net
outs = []
for it in range(4):
    for p in bb.parameters():
        p.grad = None
    d = bb(dict(batch))
    for k in ("sa1_features", "sa2_features", "sa3_features"):
        d[k].retain_grad()
    keys = ["sa1_features", "sa2_features", "sa3_features", "sa4_features", "fp2_features"]
    torch.manual_seed(1)
    cot = {k: torch.randn_like(d[k]) for k in keys[:4]}
    loss = sum((d[k] * cot[k]).sum() for k in keys[:4])
    loss.backward()
    torch.cuda.synchronize()
    outs.append(({k: d[k].detach().clone() for k in keys}, {**{n: p.grad.clone() for n, p in bb.named_parameters() if p.grad is not None}, **{"d/" + k: d[k].grad.clone() for k in ("sa1_features", "sa2_features", "sa3_features")}}))
for it in range(1, 4):
    bad_f = [k for k in outs[0][0] if not torch.equal(outs[0][0][k], outs[it][0][k])]
    bad_g = [(n, float((outs[0][1][n] - outs[it][1][n]).abs().max() / (outs[0][1][n].abs().max() + 1e-30))) for n in outs[0][1] if not torch.equal(outs[0][1][n], outs[it][1][n])]
    print("pass", it, "forward differs:", bad_f, "| grads differ:", [(n, f'{v:.1e}') for n, v in bad_g if n.startswith('d/') or 'layer2.conv' in n or 'layer0.conv' in n], len(bad_g))
