"""SA1's gather-layer weight gradient (compact rows of the bench scenes) alone on the chip, against the number of workgroups
(= partial-dW slabs): where does the kernel's time go?   python tools/time_wgrad.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("3dvlp_amd.synth")
gs = importlib.import_module("3dvlp_amd.grounding_step")
pu = importlib.import_module("3dvlp_amd.pointnet2_utils")
ext = importlib.import_module("3dvlp_amd._lib")
dev = torch.device("cuda:0")
batch = gs.batch_to_device(synth.make_batch(0, 8, 40000, 8), dev)
pc = batch["point_clouds"]
xyz, feat_pm = pc[..., :3].contiguous(), pc[..., 3:].contiguous()
B, n, m = 8, 40000, 2048
inds, fps_ws = ext.furthest_point_sampling(xyz, m, "pruned", return_workspace=True)
new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
idx = ext.ball_query_sorted(new_xyz, xyz, 0.2, 64, fps_ws)
rowptr_, crow_ = ext.sa_compact(idx, n)
cm = (crow_, rowptr_, B * m)
P = int(rowptr_[-1])
R = B * m * 64
print(f"compact rows {P} of {R} ({100.0 * P / R:.1f} %), {P // 32} tiles")
Y1 = torch.randn(R, 64, device=dev).to(torch.bfloat16)
G1 = torch.randn(R, 64, device=dev).to(torch.bfloat16)
c5_1 = torch.rand(5, 64, device=dev) + 0.5
dW = torch.empty((64, 144), device=dev)


def t(fn, n=10):
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


for nblk in (64, 128, 256, 455, 512, 1024, 2048):
    part = torch.empty((nblk, 64, 144), device=dev)
    us = t(lambda: ext.call("vlp3d_sa_wgrad", G1, Y1, R, 64, c5_1, 1, None, 144, None, None, xyz, new_xyz, idx, feat_pm, n, m, 64,
                            132, 0.2, dW, part, nblk, None, None, 0, 1, 1, *cm))
    byts = P * (64 * 2 * 2 + 132 * 4) + nblk * 64 * 144 * 4
    print(f"workgroups {nblk:5d}: {us:7.1f} us  ({P / 32 / nblk:5.1f} tiles each, slabs {nblk * 64 * 144 * 4 / 2 ** 20:5.1f} MB, {byts / us / 1e6:5.2f} TB/s of operands + slabs)")
