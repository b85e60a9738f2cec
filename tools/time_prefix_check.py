"""The FPS prefix proof (vlp3d_fps_prefix_check) on a real prefix: SA1's 2048 samples as the point set of SA2 (m = 1024), then
1024 -> 512, 512 -> 256: us per call inside a graph replay."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ext = importlib.import_module("3dvlp_amd._lib")
pu = importlib.import_module("3dvlp_amd.pointnet2_utils")
dev = torch.device("cuda:0")
torch.manual_seed(0)
xyz = torch.rand(8, 40000, 3, device=dev) * 8
idx = pu.furthest_point_sample(xyz, 2048)
pts = torch.gather(xyz, 1, idx.long()[..., None].expand(-1, -1, 3)).contiguous()
for (N, m) in ((2048, 1024), (1024, 512), (512, 256)):
    sub = pts[:, :N].contiguous()
    f = lambda: ext.furthest_point_sampling(sub, m, prefix_hint=True)
    for _ in range(3):
        out = f()
    assert (out.cpu() == torch.arange(m)[None].expand(8, m)).all()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            f()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"N={N} m={m}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (proof + conditional FPS)")
