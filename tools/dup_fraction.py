"""Fraction of UNIQUE rows in the grouped MLP inputs (ball query pads a ball with copies of its first neighbour) for the
bench's synthetic scenes and for a denser, ScanNet-like room (same generator, smaller room)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
synth = importlib.import_module("3dvlp_amd.synth")
pu = importlib.import_module("3dvlp_amd.pointnet2_utils")
def uniq(idx):
    S = idx.shape[-1]
    same = (idx[..., 1:] == idx[..., :1])
    first = torch.where(same.any(-1), same.float().argmax(-1) + 1, torch.full(same.shape[:-1], S, device=idx.device))
    return first.float()
for name, scale in (("bench scenes (8 x 8 x 3 m room)", 1.0), ("same generator scaled to 4.5 x 4.5 x 2.7 m", 4.5 / 8)):
    xyz = np.stack([synth.make_scene(i, 40000)["xyz"] for i in range(4)]).astype(np.float32)
    xyz = torch.from_numpy(xyz * np.array([scale, scale, 0.9 if scale < 1 else 1.0], np.float32)).cuda().contiguous()
    cur = xyz
    print(name)
    for lvl, (m, r, S) in enumerate(((2048, 0.2, 64), (1024, 0.4, 32), (512, 0.8, 16), (256, 1.2, 16))):
        inds = pu.furthest_point_sample(cur, m)
        new = pu.gather_operation(cur.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
        idx = pu.ball_query(r, S, cur, new)
        u = uniq(idx)
        print(f"  SA{lvl + 1}: nsample {S:2d}  mean unique {u.mean().item():5.1f}  ({100 * u.mean().item() / S:4.1f} % of rows)  full balls {100 * (u == S).float().mean().item():4.1f} %")
        cur = new
