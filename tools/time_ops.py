"""Quick per-op timing on the GPU box (development aid, not the bench contract)."""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pu = importlib.import_module("3dvlp_amd.pointnet2_utils")
ext = importlib.import_module("3dvlp_amd._lib")
synth = importlib.import_module("3dvlp_amd.synth")


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


B = 8
xyz = torch.from_numpy(np.stack([synth.make_scene(1000 + i, 40000)["xyz"] for i in range(B)])).cuda()
feat = torch.randn(B, 132, 40000, device="cuda")
print("device", torch.cuda.get_device_name(0))
cfgs = [("sa1", 40000, 2048, 0.2, 64), ("sa2", 2048, 1024, 0.4, 32), ("sa3", 1024, 512, 0.8, 16),
        ("sa4", 512, 256, 1.2, 16)]
cur = xyz
f = feat
for name, N, m, r, ns in cfgs:
    t_fps = timeit(lambda: pu.furthest_point_sample(cur, m))
    inds = pu.furthest_point_sample(cur, m)
    new_xyz = pu.gather_operation(cur.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    t_bq = timeit(lambda: pu.ball_query(r, ns, cur, new_xyz))
    idx = pu.ball_query(r, ns, cur, new_xyz)
    t_gp = timeit(lambda: pu.grouping_operation(f, idx))
    g = pu.grouping_operation(f, idx)
    go = torch.randn_like(g)
    t_gg = timeit(lambda: ext.group_points_grad(go, idx, N))
    print(f"{name}: N={N} m={m} fps {t_fps:.3f} ms  ball_query {t_bq:.3f} ms  group({f.shape[1]}ch) {t_gp:.3f} ms  "
          f"group_grad {t_gg:.3f} ms", flush=True)
    cur = new_xyz
    f = torch.randn(B, 128 if name == "sa1" else 256, m, device="cuda")
    del g, go
