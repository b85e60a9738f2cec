"""Randomised check of ball_query / three_nn against the oracle on lattice data (exact distance ties, points exactly
on the radius) and random data."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ext = importlib.import_module("3dvlp_amd._lib")
pu = importlib.import_module("3dvlp_amd.pointnet2_utils")
from oracle import oracle as orc
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    B = int(rng.integers(1, 4)); N = int(rng.choice([70, 500, 3000, 9000])); M = int(rng.choice([1, 33, 256, 700]))
    lattice = case % 2 == 0
    mk = (lambda *s: rng.integers(0, 9, s).astype(np.float32) * 0.25) if lattice else (lambda *s: rng.uniform(0, 2, s).astype(np.float32))
    xyz, new_xyz = mk(B, N, 3), mk(B, M, 3)
    r = float(rng.choice([0.25, 0.5, 0.3, 1.0])); ns = int(rng.choice([1, 16, 64]))
    got = pu.ball_query(r, ns, torch.from_numpy(xyz).cuda(), torch.from_numpy(new_xyz).cuda()).cpu().numpy()
    if not (got == orc.ball_query(new_xyz, xyz, r, ns)).all():
        bad += 1; print("ball_query mismatch", case, B, N, M, r, ns, lattice, flush=True)
    if N >= 3:
        d, i = pu.three_nn(torch.from_numpy(new_xyz).cuda(), torch.from_numpy(xyz).cuda())
        od, oi = orc.three_nn(new_xyz, xyz)
        if not (i.cpu().numpy() == oi).all():
            bad += 1; print("three_nn index mismatch", case, B, N, M, lattice, flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
