"""CPU emulation of the pruned FPS kernel's per-iteration WORK (csrc/fps_pruned.hip) on one bench scene: how many 64-point
slots pass the bounding-box test per iteration, how they fall on the 16 waves (static ownership: chunk c -> wave c % 16),
how often the slot's maximum holder moves (re-reduction), LDS-resident share.  Statistics only (numpy fp32; the kernel's
exact fma form is irrelevant here).
    python tools/fps_emulate.py [scene] [slot_points]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
synth = importlib.import_module("3dvlp_amd.synth")


def hilbert_cells(q):
    """Skilling's transpose form, 5 bits per axis (fps_cell_kernel)."""
    q = [q[:, 0].copy(), q[:, 1].copy(), q[:, 2].copy()]
    M = 1 << 4
    Q = M
    while Q > 1:
        P = Q - 1
        for a in range(3):
            hit = (q[a] & Q) != 0
            q[0] = np.where(hit, q[0] ^ P, q[0])
            t = (q[0] ^ q[a]) & P
            q0n = np.where(hit, q[0], q[0] ^ t)
            qan = np.where(hit, q[a], q[a] ^ t)
            q[0] = q0n
            if a != 0:
                q[a] = qan
        Q >>= 1
    q[1] ^= q[0]
    q[2] ^= q[1]
    t = np.zeros_like(q[0])
    Q = M
    while Q > 1:
        t = np.where((q[2] & Q) != 0, t ^ (Q - 1), t)
        Q >>= 1
    q = [v ^ t for v in q]

    def spread(v):
        return (v & 1) | ((v & 2) << 2) | ((v & 4) << 4) | ((v & 8) << 6) | ((v & 16) << 8)
    return (spread(q[0]) << 2) | (spread(q[1]) << 1) | spread(q[2])


def main():
    scene = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    SP = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    N, m, W = 40000, 2048, 16
    xyz = synth.make_scene(scene, N)["xyz"].astype(np.float32)
    lo, hi = xyz.min(0), xyz.max(0)
    c = np.clip(((xyz - lo) / (hi - lo) * 32).astype(np.int64), 0, 31)
    order = np.argsort(hilbert_cells(c), kind="stable")
    p = xyz[order]
    nchunk = (N + SP - 1) // SP
    pad = nchunk * SP - N
    pp = np.concatenate([p, np.repeat(p[-1:], pad, 0)]) if pad else p
    valid = np.arange(nchunk * SP) < N
    pts = pp.reshape(nchunk, SP, 3)
    vmask = valid.reshape(nchunk, SP)
    blo = np.where(vmask[..., None], pts, 3e38).min(1)
    bhi = np.where(vmask[..., None], pts, -3e38).max(1)
    temp = np.where(vmask, np.float32(1e10), np.float32(-1)).astype(np.float32)
    smax = temp.max(1)
    shold = temp.argmax(1)
    q = xyz[0]
    wave_of = np.arange(nchunk) % W
    lds_chunk = (np.arange(nchunk) // W) < 9 * (64 // SP)
    act_hist, max_hist, moved_tot, act_tot, lds_tot = [], [], 0, 0, 0
    for j in range(1, m):
        e = np.maximum(0, np.maximum(blo - q, q - bhi))
        lb2 = (e * e).sum(1) * np.float32(1 - 1e-5)
        act = np.nonzero(lb2 < smax)[0]
        for s in act:
            d = ((pts[s] - q) ** 2).sum(1).astype(np.float32)
            t = np.where(vmask[s], np.minimum(d, temp[s]), np.float32(-1))
            moved = t[shold[s]] < temp[s, shold[s]]
            temp[s] = t
            if moved:
                moved_tot += 1
                smax[s] = t.max()
                shold[s] = t.argmax()
        per_wave = np.bincount(wave_of[act], minlength=W)
        act_hist.append(len(act))
        max_hist.append(per_wave.max())
        act_tot += len(act)
        lds_tot += lds_chunk[act].sum()
        best = smax.argmax()
        q = pts[best, shold[best]]
    act_hist, max_hist = np.array(act_hist), np.array(max_hist)
    print(f"scene {scene}, {SP}-point slots: {nchunk} slots; active per iteration mean {act_hist.mean():.2f} "
          f"(p50 {np.median(act_hist):.0f}, p90 {np.percentile(act_hist, 90):.0f}, max {act_hist.max()}), points touched "
          f"{act_hist.mean() * SP:.0f}")
    print(f"  slowest wave's slots per iteration (static chunk %% 16): mean {max_hist.mean():.2f}; balanced over 16 waves: "
          f"mean {np.ceil(act_hist / W).mean():.2f}")
    print(f"  holder moved (re-reduction) in {100 * moved_tot / act_tot:.1f} % of the updates; LDS-resident {100 * lds_tot / act_tot:.1f} %")
    for lo_, hi_ in ((1, 256), (256, 1024), (1024, 2047)):
        sl = slice(lo_ - 1, hi_ - 1)
        print(f"  iterations {lo_}..{hi_}: active {act_hist[sl].mean():.2f}, slowest wave {max_hist[sl].mean():.2f}")


if __name__ == "__main__":
    main()
