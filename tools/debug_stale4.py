import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gs = importlib.import_module("3dvlp_amd.grounding_step"); synth = importlib.import_module("3dvlp_amd.synth")
devc = torch.device("cuda:0")
A = gs.batch_to_device(synth.make_batch(0, 2, num_points=8192, lang_num_max=2), devc)
Bb = gs.batch_to_device(synth.make_batch(2, 2, num_points=8192, lang_num_max=2), devc)
for b in (A, Bb): b["random"] = torch.tensor(0.25, device=devc)
mode = sys.argv[1]
kw = {"pipeline": mode != "single", "use_graph": True}
step = gs.GroundingStep(devc, lr=0.0, **kw)
step.model.eval()
for m in step.model.modules():
    if isinstance(m, torch.nn.modules.batchnorm._BatchNorm): m.train()
seq = [(A, None)] * 4 if mode == "same" else [(A, None), (Bb, None), (A, None), (Bb, None)]
for i, (cur, nxt) in enumerate(seq):
    loss = float(step.run(cur, nxt)); torch.cuda.synchronize()
    print(mode, i, "loss", loss, "ptr", step._static_loss.data_ptr(), "gradmax", step.bucket.flat.abs().max().item())
