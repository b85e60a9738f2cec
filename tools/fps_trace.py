"""Per-wave, per-iteration trace of the pruned FPS kernel (SA1 of cfg2) and its sensitivity to where the slots live and to
what the rest of the chip does meanwhile.
    python tools/fps_trace.py
Prints (1) production kernel time with 0 / 4 / 9 LDS slots per wave, alone and beside a streaming-read / an fp32-FMA / a bf16-MFMA
load on the main stream; (2) from the trace: a least-squares cost model of a wave's update phase (cycles = a + b * LDS slots +
c * L2 slots + d * re-reductions), and the critical wave's share of every iteration."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ext = importlib.import_module("3dvlp_amd._lib")
synth = importlib.import_module("3dvlp_amd.synth")
B, N, m = 8, 40000, 2048
dev = torch.device("cuda:0")
xyz = torch.from_numpy(np.stack([synth.make_scene(1000 + i, N)["xyz"] for i in range(B)])).to(dev)
nbytes = int(ext.load().vlp3d_fps_workspace_bytes(B, N))
ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
idx = torch.empty((B, m), dtype=torch.int32, device=dev)
ref = torch.empty_like(idx)
ext.call("vlp3d_fps_pruned_trace", xyz, B, N, m, ws, nbytes, ref, None, None, -1, 1)  # the round-3 kernel (== dense == oracle in the tests)
side = torch.cuda.Stream()
sink = torch.zeros(1, device=dev)
big = torch.empty(1 << 30, dtype=torch.uint8, device=dev).zero_()


VARIANT = 0


def fps(lds):
    ext.call("vlp3d_fps_pruned_trace", xyz, B, N, m, ws, nbytes, idx, None, None, lds, VARIANT)


LOADS = {
    "alone": None,
    "beside streaming read": lambda: ext.call("vlp3d_probe_read", big, big.numel(), 2048, sink),
    "beside fp32 FMA": lambda: ext.call("vlp3d_probe_fma_f32", 8192, 2048, sink),
    "beside bf16 MFMA": lambda: ext.call("vlp3d_probe_mfma_bf16", 1024, 2048, sink),
}


def timed(lds, load, reps=3):
    out = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            s.record()
            fps(lds)
            e.record()
        if load is not None:
            for _ in range(40):   # keep the main stream busy for longer than the FPS
                load()
        torch.cuda.synchronize()
        out.append(s.elapsed_time(e))
    return sorted(out[1:])[len(out[1:]) // 2]


for VARIANT, lds_opts in ((1, (0, 4, 9)), (0, (0, 6, 12))):
    print("round-3 kernel (running minima in LDS / L2)" if VARIANT else "register-resident kernel", flush=True)
    for name, load in LOADS.items():
        if load is not None:  # how long one load launch takes (to be sure 40 of them outlast the FPS)
            load()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            load()
            e.record()
            e.synchronize()
            tl = s.elapsed_time(e)
        else:
            tl = 0.0
        row = [timed(lds, load) for lds in lds_opts]
        assert torch.equal(idx, ref)
        print(f"{name:24s} (load launch {tl:6.3f} ms)  entry point with %d / %d / %d LDS slots per wave: " % lds_opts
              + " / ".join(f"{t:.3f}" for t in row) + " ms", flush=True)

ph = torch.zeros((B, 8), dtype=torch.int64, device=dev)
tr = torch.zeros((B, m, 16, 8), dtype=torch.int32, device=dev)
ext.call("vlp3d_fps_pruned_trace", xyz, B, N, m, ws, nbytes, idx, ph, tr, -1, 0)
torch.cuda.synchronize()
assert torch.equal(idx, ref)
t = tr.cpu().numpy().astype(np.float64)[:, 1:]          # (B, m-1, 16, 8)
names = ["test", "updates", "candidate", "LDS+barrier", "block reduction"]
tot = t[..., :5].sum(-1)                                  # per wave per iteration
print("per-iteration cycles (mean over waves and iterations): " + ", ".join(f"{n} {t[..., k].mean():.0f}" for k, n in enumerate(names))
      + f"; total {tot.mean():.0f}")
work = t[..., 0] + t[..., 1] + t[..., 2]                  # what a wave does between two barriers before it arrives
crit = work.max(-1)                                       # the slowest wave of each iteration
print(f"arrival at the barrier: mean wave {work.mean():.0f} cycles after the previous block reduction, slowest wave {crit.mean():.0f}; "
      f"barrier wait of the slowest wave (min over waves of phase 3) {t[..., 3].min(-1).mean():.0f}")
nl, ng, nr = t[..., 5].reshape(-1), t[..., 6].reshape(-1), t[..., 7].reshape(-1)
y = t[..., 1].reshape(-1)
A = np.stack([np.ones_like(nl), nl, ng, nr], 1)
coef, *_ = np.linalg.lstsq(A, y, rcond=None)
print(f"update phase of a wave ~ {coef[0]:.0f} + {coef[1]:.0f} x LDS slots + {coef[2]:.0f} x L2 slots + {coef[3]:.0f} x re-reductions cycles "
      f"(slots per wave and iteration: LDS {nl.mean():.3f}, L2 {ng.mean():.3f}, re-reduced {nr.mean():.3f})")
for k in (0, 1, 2, 3):
    sel = (nl + ng) == k
    if sel.any():
        print(f"  waves with {k} active slots: {100 * sel.mean():5.1f} % of (wave, iteration); update phase {y[sel].mean():.0f}, "
              f"candidate {t[..., 2].reshape(-1)[sel].mean():.0f}, test {t[..., 0].reshape(-1)[sel].mean():.0f}")
# the slowest wave: what it did
wi = work.argmax(-1)
take = lambda a: np.take_along_axis(a, wi[..., None], -1)[..., 0]
print(f"slowest wave of an iteration: LDS slots {take(t[..., 5]).mean():.2f}, L2 slots {take(t[..., 6]).mean():.2f}, "
      f"re-reductions {take(t[..., 7]).mean():.2f}; test {take(t[..., 0]).mean():.0f}, updates {take(t[..., 1]).mean():.0f}, "
      f"candidate {take(t[..., 2]).mean():.0f}, block reduction {take(t[..., 4]).mean():.0f}")
