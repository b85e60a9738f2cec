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


def case(name, B, N, M, S, C, cout, idx_mode):
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
    us = t(lambda: ext.call("vlp3d_sa_fwd_gather", xyz, new_xyz, idx, feat, B, N, M, S, C, 0.3, W, K1, cout, Y, st, 1, None, None, 0))
    gb = R * C * 4 / 1e9
    print(f"{name:44s} R={R:7d} K={K1:3d}  {us:7.1f} us   gathered {gb / (us * 1e-6) / 1e3:5.2f} TB/s")


case("SA4 shape, random idx", 8, 512, 256, 16, 256, 128, "random")
case("SA4 shape, local idx", 8, 512, 256, 16, 256, 128, "local")
case("SA4 shape, same row", 8, 512, 256, 16, 256, 128, "same")
case("SA3 shape, random idx", 8, 1024, 512, 16, 256, 128, "random")
case("2x SA3 rows", 8, 1024, 1024, 16, 256, 128, "random")
case("4x SA3 rows", 8, 1024, 2048, 16, 256, 128, "random")
case("SA2 shape (K = 144)", 8, 2048, 1024, 32, 128, 128, "random")
case("K = 144 at SA4's rows", 8, 512, 256, 16, 128, 128, "random")
case("K = 144 at SA3's rows", 8, 1024, 512, 16, 128, 128, "random")
case("K = 16 at SA4's rows, cout 128", 8, 512, 256, 16, 12, 128, "random")
case("K = 16 at SA4's rows, cout 64", 8, 512, 256, 16, 12, 64, "random")
case("K = 80 at SA4's rows, cout 128", 8, 512, 256, 16, 76, 128, "random")
case("K = 16, 8 tiles per wave, cout 128", 8, 2048, 1024, 32, 12, 128, "random")
case("K = 16, 16 tiles per wave, cout 128", 8, 2048, 2048, 32, 12, 128, "random")
