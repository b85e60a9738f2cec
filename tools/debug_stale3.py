import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gs = importlib.import_module("3dvlp_amd.grounding_step"); synth = importlib.import_module("3dvlp_amd.synth")
devc = torch.device("cuda:0")
A = gs.batch_to_device(synth.make_batch(0, 2, num_points=8192, lang_num_max=2), devc)
Bb = gs.batch_to_device(synth.make_batch(2, 2, num_points=8192, lang_num_max=2), devc)
for b in (A, Bb): b["random"] = torch.tensor(0.25, device=devc)
step = gs.GroundingStep(devc, lr=0.0, pipeline=True, use_graph=True)
orig = step.forward_loss
stash = {}
def fl(batch, geometry=None):
    loss, d = orig(batch, geometry)
    stash["d"] = d
    return loss, d
step.forward_loss = fl
step.model.eval()
for m in step.model.modules():
    if isinstance(m, torch.nn.modules.batchnorm._BatchNorm): m.train()
keys = ("vote_loss", "objectness_loss", "box_loss", "ref_loss", "diou_loss", "lang_con_loss", "iou_con_loss", "con_loss", "loss", "lang_loss")
for i, (cur, nxt) in enumerate([(A, None), (Bb, None), (A, Bb)]):
    loss = float(step.run(cur, nxt)); torch.cuda.synchronize()
    d = stash["d"]
    print(i, "loss", loss, {k: float(d[k]) for k in keys if k in d})
