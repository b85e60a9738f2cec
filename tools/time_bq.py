"""Ball query of SA1 (cfg2: 8 x 2048 centres, 40 000 points, r = 0.2, 64 samples): the one-launch form on the FPS's spatial sort
against the six-launch grid form and the all-pairs scan, alone on the chip; number of centres that take the fall-back scan."""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ext = importlib.import_module("3dvlp_amd._lib")
synth = importlib.import_module("3dvlp_amd.synth")
B, N, m = 8, 40000, 2048
xyz = torch.from_numpy(np.stack([synth.make_scene(1000 + i, N)["xyz"] for i in range(B)])).cuda()
inds, ws = ext.furthest_point_sampling(xyz, m, "pruned", return_workspace=True)
new_xyz = ext.gather_xyz(xyz, inds)


def timed(f, reps=20):
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(4):
            f()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps / 4 * 1e3


a = ext.ball_query_sorted(new_xyz, xyz, 0.2, 64, ws)
b = ext.ball_query(new_xyz, xyz, 0.2, 64, "scan")
assert torch.equal(a, b)
print(f"sorted {timed(lambda: ext.ball_query_sorted(new_xyz, xyz, 0.2, 64, ws)):.1f} us   grid {timed(lambda: ext.ball_query(new_xyz, xyz, 0.2, 64, 'grid')):.1f} us   "
      f"scan {timed(lambda: ext.ball_query(new_xyz, xyz, 0.2, 64, 'scan')):.1f} us")
# candidates per centre (host emulation of the cell box)
pts = xyz.cpu().numpy()
lo, hi = pts.min(1), pts.max(1)
c = np.clip(((pts - lo[:, None]) / (hi - lo)[:, None] * 32).astype(np.int64), 0, 31)
q = new_xyz.cpu().numpy()
tot = []
for b_ in range(2):
    key = (c[b_, :, 2] * 32 + c[b_, :, 1]) * 32 + c[b_, :, 0]
    cnt = np.bincount(key, minlength=32768).reshape(32, 32, 32)
    for i in range(0, m, 16):
        r = 0.2 * 1.001
        l = np.clip(((q[b_, i] - r - lo[b_]) / (hi[b_] - lo[b_]) * 32).astype(np.int64), 0, 31)
        h = np.clip(((q[b_, i] + r - lo[b_]) / (hi[b_] - lo[b_]) * 32).astype(np.int64), 0, 31)
        tot.append((cnt[l[2]:h[2] + 1, l[1]:h[1] + 1, l[0]:h[0] + 1].sum(), (h - l + 1).prod()))
tot = np.array(tot)
print(f"candidates per centre: mean {tot[:, 0].mean():.0f}, p90 {np.percentile(tot[:, 0], 90):.0f}, max {tot[:, 0].max()}; cells per box mean "
      f"{tot[:, 1].mean():.1f} max {tot[:, 1].max()}; above 1024 candidates: {(tot[:, 0] > 1024).mean() * 100:.1f} %")
