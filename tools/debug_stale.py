import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gs = importlib.import_module("3dvlp_amd.grounding_step"); synth = importlib.import_module("3dvlp_amd.synth")
devc = torch.device("cuda:0")
A = gs.batch_to_device(synth.make_batch(0, 2, num_points=8192, lang_num_max=2), devc)
Bb = gs.batch_to_device(synth.make_batch(2, 2, num_points=8192, lang_num_max=2), devc)
for b in (A, Bb): b["random"] = torch.tensor(0.25, device=devc)
for kw in ({}, {"pipeline": True, "use_graph": True}):
    step = gs.GroundingStep(devc, lr=0.0, **kw)
    step.model.eval()
    for m in step.model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm): m.train()
    for i, (cur, nxt) in enumerate([(A, None), (Bb, None), (A, Bb), (Bb, A)]):
        loss = float(step.run(cur, nxt)); torch.cuda.synchronize()
        g = step.bucket.flat
        pf = step.layout.flat
        bad = (~torch.isfinite(g)).nonzero()
        print(kw, i, "loss", loss, "grad max", g.abs().max().item(), "nonfinite grads", bad.numel(), "param nonfinite", (~torch.isfinite(pf)).sum().item(),
              "first bad", bad[:3].flatten().tolist(), "argmax", int(g.abs().argmax()))
        if g.abs().max().item() > 1e6:
            off = int(g.abs().argmax())
            for p, o in zip(step.layout.params, step.layout.offsets):
                if o <= off < o + p.numel():
                    name = [n for n, q in step.model.named_parameters() if q is p][0]
                    print("   in", name, tuple(p.shape), "offset in param", off - o)
