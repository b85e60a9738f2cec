import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gs = importlib.import_module("3dvlp_amd.grounding_step"); synth = importlib.import_module("3dvlp_amd.synth")
ddp = importlib.import_module("3dvlp_amd.ddp")
mode = sys.argv[1]
if mode == "noop":
    ddp.FlatAdamW.step = lambda self: None
elif mode == "setactive_only":
    ddp.FlatAdamW.step = lambda self: self._set_active()
devc = torch.device("cuda:0")
A = gs.batch_to_device(synth.make_batch(0, 2, num_points=8192, lang_num_max=2), devc)
Bb = gs.batch_to_device(synth.make_batch(2, 2, num_points=8192, lang_num_max=2), devc)
for b in (A, Bb): b["random"] = torch.tensor(0.25, device=devc)
step = gs.GroundingStep(devc, lr=0.0, pipeline=True, use_graph=True)
step.model.eval()
for m in step.model.modules():
    if isinstance(m, torch.nn.modules.batchnorm._BatchNorm): m.train()
for i, (cur, nxt) in enumerate([(A, None), (Bb, None), (A, Bb), (Bb, A)]):
    loss = float(step.run(cur, nxt)); torch.cuda.synchronize()
    print(mode, i, "loss", loss, "grad max", step.bucket.flat.abs().max().item())
