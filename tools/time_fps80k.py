import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ext = importlib.import_module("3dvlp_amd._lib"); synth = importlib.import_module("3dvlp_amd.synth")
xyz = torch.from_numpy(np.stack([synth.make_scene(10 + i, 80000)["xyz"] for i in range(8)])).cuda()
for alg in ("dense", "pruned"):
    ext.furthest_point_sampling(xyz, 2048, alg); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3): ext.furthest_point_sampling(xyz, 2048, alg)
    e.record(); e.synchronize()
    print(alg, "80000 -> 2048, B=8:", round(s.elapsed_time(e) / 3, 3), "ms")
