"""Writes tests/golden/contract_vectors.npz: small fixed-seed inputs for the three index-producing ops of
lib/pointnet2/_ext_src (furthest_point_sampling sampling_gpu.cu:74-178, ball_query ball_query_gpu.cu:14-49, three_nn
interpolate_gpu.cu:14-64) and the indices the C restatement (oracle/pointnet2_oracle.c) returns for them under each of the
three fp32 evaluation orders of a*a + b*b + c*c that a build of the reference can have:

    mode 0   no contraction            (a*a + b*b) + c*c, every product and sum rounded      (nvcc -fmad=false)
    mode 1   fma(c,c, fma(a,a, b*b))    LLVM-NVPTX's aggressive FMA fusion of the expression   (nvcc default, assumed)
    mode 2   fma(c,c, fma(b,b, a*a))    the left chain

Inputs and indices only — no reference source, no reference output.  WHY: the reference's native code cannot be built in this
image (CUDA only), so nobody has been able to check which mode its `nvcc -O2` build (lib/pointnet2/setup.py:26-27) really
emits.  Anyone who holds a built `pointnet2._ext` can run tools/check_contract_vectors.py against this file: the mode whose
indices it reproduces is the one to select here (3dvlp_amd._lib.set_fp_contract / VLP3D_FP_CONTRACT).  The point sets are
chosen so that the modes DISAGREE: lattices with spacing 0.1 (not representable in binary: symmetric neighbours' distances
differ in the last bit by evaluation order), exact duplicates, radii equal to lattice distances, a scene with points inside
the FPS skip ball |p|^2 <= 1e-3 (sampling_gpu.cu:106).

    python tests/golden/make_contract_vectors.py          (CPU only; needs the built oracle)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import oracle as orc  # noqa: E402


def lattice(rng, n_side, spacing, jitter=0.0, origin=0.35):
    g = np.stack(np.meshgrid(*[np.arange(n_side)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    p = origin + g * spacing + (rng.normal(0, jitter, g.shape) if jitter else 0.0)
    return p[rng.permutation(len(p))].astype(np.float32)


def build():
    rng = np.random.default_rng(20241004)
    out = {}
    # ---- furthest_point_sampling: (name, xyz (B,N,3), npoint)
    fps_cases = {
        "fps_lattice": (np.stack([lattice(rng, 12, 0.1), lattice(rng, 12, 0.1, origin=-0.55)]), 256),
        "fps_lattice_jitter": (np.stack([lattice(rng, 11, 0.1, jitter=1e-7)]), 300),
        "fps_uniform": (rng.uniform(-2, 2, (2, 3000, 3)).astype(np.float32), 512),
        "fps_dups": (rng.uniform(-1, 1, (1, 200, 3)).astype(np.float32)[:, rng.integers(0, 200, 1500)], 200),
    }
    skip = rng.uniform(0.5, 3.0, (1, 2000, 3)).astype(np.float32)
    skip[0, rng.choice(2000, 16, replace=False)] = rng.uniform(-0.015, 0.015, (16, 3)).astype(np.float32)
    skip[0, 7] = np.float32(np.sqrt(1e-3 / 3))      # |p|^2 at the threshold of the double-literal comparison
    fps_cases["fps_skip_ball"] = (skip, 300)
    for name, (xyz, m) in fps_cases.items():
        out[name + "/xyz"] = xyz
        out[name + "/npoint"] = np.int32(m)
        for mode in (0, 1, 2):
            out[f"{name}/idx_mode{mode}"] = orc.furthest_point_sampling(xyz, m, contract=mode)
    # ---- ball_query: radius equal to lattice distances (d2 == r2 decided by the last bit), dense balls (more hits than nsample)
    lat = np.stack([lattice(rng, 12, 0.1), lattice(rng, 12, 0.1, jitter=1e-7)])
    bq_cases = {
        "bq_lattice_r3": (lat[:, :160].copy(), lat, np.float32(0.3), 16),
        "bq_lattice_r2": (lat[:, 100:260].copy(), lat, np.float32(0.2), 32),
        "bq_uniform": (rng.uniform(-1, 1, (2, 100, 3)).astype(np.float32), rng.uniform(-1, 1, (2, 2500, 3)).astype(np.float32),
                       np.float32(0.25), 16),
    }
    for name, (new_xyz, xyz, r, ns) in bq_cases.items():
        out[name + "/new_xyz"], out[name + "/xyz"], out[name + "/radius"], out[name + "/nsample"] = new_xyz, xyz, r, np.int32(ns)
        for mode in (0, 1, 2):
            out[f"{name}/idx_mode{mode}"] = orc.ball_query(new_xyz, xyz, float(r), ns, contract=mode)
    # ---- three_nn: unknown points at lattice cell centres / on lattice points: equidistant known points
    known = np.stack([lattice(rng, 8, 0.1)])
    centres = (known[:, :300] + np.float32(0.05)).astype(np.float32)
    nn_cases = {
        "nn_cell_centres": (centres, known),
        "nn_on_lattice": (known[:, 100:400].copy(), known),
        "nn_uniform": (rng.uniform(0, 1, (2, 300, 3)).astype(np.float32), rng.uniform(0, 1, (2, 77, 3)).astype(np.float32)),
    }
    for name, (unknown, kn) in nn_cases.items():
        out[name + "/unknown"], out[name + "/known"] = unknown, kn
        for mode in (0, 1, 2):
            _, idx = orc.three_nn(unknown, kn, contract=mode)
            out[f"{name}/idx_mode{mode}"] = idx
    return out, list(fps_cases) + list(bq_cases) + list(nn_cases)


if __name__ == "__main__":
    data, names = build()
    path = os.path.join(HERE, "contract_vectors.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, os.path.getsize(path), "bytes")
    for n in names:
        a, b, c = (data[f"{n}/idx_mode{k}"] for k in (0, 1, 2))
        print(f"  {n:20s} entries {a.size:6d}; differing entries mode0|1 {int((a != b).sum()):5d}  mode1|2 {int((b != c).sum()):5d}  "
              f"mode0|2 {int((a != c).sum()):5d}")
