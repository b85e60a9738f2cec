"""Time the grouped-MLP gather layer at the 256-channel levels (K = 272 -> 128) in isolation: rows, locality and tiles per wave.
python tools/time_wide_gather.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ext = importlib.import_module("3dvlp_amd._lib")
dev = torch.device("cuda:0")


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): g.replay()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / (5 * n) * 1e3


def case(name, B, N, M, S, C, cout, idx_mode, bf=False):
    g = torch.Generator(device="cpu").manual_seed(0)
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    new_xyz = xyz[:, :M].contiguous()
    feat = torch.randn(B, N, C, generator=g).to(dev)
    if idx_mode == "random":
        idx = torch.randint(0, N, (B, M, S), generator=g, dtype=torch.int32).to(dev)
    elif idx_mode == "same":
        idx = torch.zeros((B, M, S), dtype=torch.int32, device=dev)
    else:  # local: neighbours of the centre in index order
        idx = ((torch.arange(M)[None, :, None] + torch.arange(S)[None, None, :]) % N).expand(B, M, S).contiguous().to(torch.int32).to(dev)
    K1 = (C + 4 + 15) // 16 * 16
    W = (torch.randn(cout, K1, generator=g) * 0.05).to(dev).to(torch.bfloat16)
    R = B * M * S
    Y = torch.empty((R, cout), dtype=torch.bfloat16, device=dev)
    ns = int(ext.load().vlp3d_sa_stat_slabs(R))
    st = torch.empty((ns, 2, cout), dtype=torch.float64, device=dev)
    if bf:
        Cp = (C + 7) // 8 * 8
        fb = torch.zeros(B, N, Cp, device=dev, dtype=torch.bfloat16)
        fb[..., :C] = feat.to(torch.bfloat16)
        feat = fb
    us = t(lambda: ext.call("vlp3d_sa_fwd_gather", xyz, new_xyz, idx, feat, B, N, M, S, C, 0.3, W, K1, cout, Y, st, 3 if bf else 1, None, None, 0))
    gb = R * C * 4 / 1e9
    print(f"{name:44s} R={R:7d} K={K1:3d}  {us:7.1f} us   gathered {gb / (us * 1e-6) / 1e3:5.2f} TB/s")


case("SA1 shape fp32 features", 8, 40000, 2048, 64, 132, 64, "random")
case("SA1 shape bf16 features", 8, 40000, 2048, 64, 132, 64, "random", True)
case("SA1 shape fp32 features, local", 8, 40000, 2048, 64, 132, 64, "local")
case("SA1 shape bf16 features, local", 8, 40000, 2048, 64, 132, 64, "local", True)
case("SA2 shape fp32 features", 8, 2048, 1024, 32, 128, 128, "random")
case("SA2 shape bf16 features", 8, 2048, 1024, 32, 128, 128, "random", True)
