"""Randomised bit-exactness check of the pruned FPS against the dense kernel (itself tested against the oracle):
uniform / clustered / lattice (many exact ties) / duplicated / planar / skipped-ball point sets."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ext = importlib.import_module("3dvlp_amd._lib")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for case in range(ncase):
    B = int(rng.integers(1, 4))
    N = int(rng.choice([8192, 9000, 12345, 16384, 20000, 33333, 40000, 50000, 65536]))
    m = int(rng.choice([1, 2, 17, 256, 700, 1024, 2048]))
    kind = rng.choice(["uniform", "clusters", "lattice", "dups", "plane", "skip", "line"])
    if kind == "uniform":
        p = rng.uniform(-3, 3, (B, N, 3))
    elif kind == "clusters":
        c = rng.uniform(-4, 4, (B, 20, 3)); p = c[:, rng.integers(0, 20, N)] + rng.normal(0, 0.05, (B, N, 3))
    elif kind == "lattice":
        g = rng.integers(0, 12, (B, N, 3)).astype(np.float64) * 0.25 + 0.5   # heavy duplication and exact distance ties
        p = g
    elif kind == "dups":
        base = rng.uniform(-2, 2, (B, N // 8 + 1, 3)); p = base[:, rng.integers(0, N // 8 + 1, N)]
    elif kind == "plane":
        p = rng.uniform(-3, 3, (B, N, 3)); p[..., 2] = 1.0
    elif kind == "line":
        t = rng.uniform(-5, 5, (B, N, 1)); p = np.concatenate([t, 0.5 * t + 1, np.full_like(t, 0.3)], -1)
    else:
        p = rng.uniform(-1, 1, (B, N, 3)); p[:, rng.integers(0, N, N // 10)] *= 0.01   # many points inside the skip ball
    x = torch.from_numpy(p.astype(np.float32)).cuda().contiguous()
    a = ext.furthest_point_sampling(x, m, algorithm="dense")
    b = ext.furthest_point_sampling(x, m, algorithm="pruned")
    ok = torch.equal(a, b)
    bad += (not ok)
    if not ok:
        d = (a != b).nonzero()[0].tolist()
        print(f"MISMATCH case {case}: kind={kind} B={B} N={N} m={m} first diff at {d}: dense {int(a[d[0], d[1]])} pruned {int(b[d[0], d[1]])}", flush=True)
print(f"{ncase} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
