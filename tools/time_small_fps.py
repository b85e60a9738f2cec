"""Dense FPS at the proposal-sampling shape (8 x 1024 -> 256) and the SA fall-back shapes: us per launch."""
import importlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

ext = importlib.import_module("3dvlp_amd._lib")
pu = importlib.import_module("3dvlp_amd.pointnet2_utils")
dev = torch.device("cuda:0")
for (B, N, m) in ((8, 1024, 256), (8, 2048, 1024), (8, 512, 256), (8, 256, 256), (8, 4096, 1024)):
    xyz = torch.rand(B, N, 3, device=dev) * 8
    for _ in range(3):
        idx = pu.furthest_point_sample(xyz, m)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        idx = pu.furthest_point_sample(xyz, m)
    e1.record()
    torch.cuda.synchronize()
    print(f"B={B} N={N} m={m}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call, {e0.elapsed_time(e1) / 20 * 1e3 / (m - 1):.3f} us per iteration")
