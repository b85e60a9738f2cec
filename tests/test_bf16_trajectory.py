"""Does the TIMED arithmetic train like the PARITY arithmetic?  (VERDICT r3 #7b.)

bench.py times the bf16 configuration (bf16 storage + bf16 MFMA operands, distinct-row evaluation, row chains); the 1e-4 gates
are met by the exact-fp32 configuration.  At random initialisation the two differ by 10 % in the trunk's outputs and ~70 % in
its SA gradients (profiles/r03_step_parity_trunk.txt) — "no worse than the reference's own sequence under bf16 autocast", which
says nothing about training.  This test trains BOTH configurations from the same initial weights for 200 steps on the same
three cfg2 batches (8 scenes x 40 000 points, dropout on, the captured + pipelined step of the bench) and

  * bounds the gap of the smoothed loss trajectories (windows of 30 steps = 10 passes over the three batches, from step 50 on)
    by the spread of two fp32 runs that differ only in their dropout masks / copy-paste coins: the mean loss over steps
    50-199 within max(10 %, 2 x the fp32 pair's gap), every window within max(25 %, 2.5 x the pair's largest window gap) (five GPU runs of this test measured
    fp32-vs-fp32 window gaps of 0.3-23 % and bf16-vs-fp32 gaps of 0.1-19 %: the trajectory of this network is that noisy; tighter
    per-window bounds against the two-sample noise estimate failed in two of those runs, once with bf16 BELOW fp32);
  * repeats tests/test_step_parity.py's trunk comparison (fixed cotangent; fp32 kernels / bf16 padded / bf16 distinct rows /
    the reference's literal sequence under bf16 autocast) on the weights AFTER those 200 steps.

Tables: gpurun_out/bf16_trajectory.txt, gpurun_out/step_parity_trunk_trained.txt (copied to profiles/r04_*)."""
import importlib
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from tests.test_step_parity import TRUNK_BLOCKS, _gpu_trunk, _per_block, _write

pytestmark = pytest.mark.gpu
STEPS, WINDOW, FIRST = 200, 30, 50


def _train(gs, add_norm, batches, sa_dtype, side, seed_word):
    devc = torch.device("cuda:0")
    add_norm.state(devc).fill_(seed_word)          # the dropout masks of a run are a function of this word alone
    torch.manual_seed(20240 + (seed_word % 7))     # ... and the copy-paste coins of the generator's state (same for equal words)
    torch.cuda.manual_seed_all(20240 + (seed_word % 7))
    step = gs.GroundingStep(devc, epoch=50, lr=1e-3, sa_dtype=sa_dtype, use_graph=True, pipeline=True, seed=0, side_stream=side)
    init = {k: v.detach().clone() for k, v in step.model.state_dict().items()}
    losses = []
    for i in range(STEPS):
        losses.append(float(step.run(batches[i % 3], batches[(i + 1) % 3])))
    torch.cuda.synchronize()
    final = {k: v.detach().clone().cpu() for k, v in step.model.state_dict().items()}
    return step, np.array(losses), init, final


def test_bf16_configuration_trains_like_fp32_for_200_steps():
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    synth = importlib.import_module("3dvlp_amd.synth")
    add_norm = importlib.import_module("3dvlp_amd.add_norm")
    devc = torch.device("cuda:0")
    batches_np = [synth.make_batch(8 * j, 8, 40000, 8) for j in range(3)]
    batches = [gs.batch_to_device(b, devc) for b in batches_np]
    s0, fa, init_a, final_a = _train(gs, add_norm, batches, None, None, 1234567)
    side = s0._side
    del s0
    s1, fb, init_b, _ = _train(gs, add_norm, batches, None, side, 7654321)
    del s1
    s2, bf, init_c, final_c = _train(gs, add_norm, batches, torch.bfloat16, side, 1234567)
    del s2
    for k in init_a:   # the three runs start from the same weights
        assert torch.equal(init_a[k], init_b[k]) and torch.equal(init_a[k], init_c[k]), k
    assert np.isfinite(fa).all() and np.isfinite(fb).all() and np.isfinite(bf).all()
    wins = [(a, a + WINDOW) for a in range(FIRST, STEPS - WINDOW + 1, WINDOW)]
    m = lambda x: np.array([x[a:b].mean() for a, b in wins])
    wa, wb, wc = m(fa), m(fb), m(bf)
    noise = np.abs(wb - wa) / wa
    gap = np.abs(wc - wa) / wa
    lines = [f"fp32 (exact-fp32 MFMA, padded rows) vs bf16 (timing configuration) for {STEPS} steps on three cfg2 batches, same "
             f"initial weights, lr 1e-3, dropout on; loss averaged over windows of {WINDOW} steps",
             f"first step: fp32 {fa[0]:.4f}  fp32' {fb[0]:.4f}  bf16 {bf[0]:.4f};  steps 0-9 mean: {fa[:10].mean():.3f} {fb[:10].mean():.3f} "
             f"{bf[:10].mean():.3f}",
             f"{'steps':>10s} {'fp32':>9s} {'fp32 (other dropout masks)':>27s} {'bf16':>9s} {'|fp32′-fp32|/fp32':>18s} {'|bf16-fp32|/fp32':>17s}"]
    for (a, b), x, y, z, n_, g_ in zip(wins, wa, wb, wc, noise, gap):
        lines.append(f"{a:4d}-{b - 1:<5d} {x:9.4f} {y:27.4f} {z:9.4f} {n_:18.4f} {g_:17.4f}")
    _write("bf16_trajectory.txt", lines)
    assert fa[-WINDOW:].mean() < 0.8 * fa[:10].mean() and bf[-WINDOW:].mean() < 0.8 * bf[:10].mean()   # both really train
    # The trajectories are chaotic: float-atomic order alone (DESIGN.md 4.20 "Reproducibility") moves a 30-step window by up to
    # ~20 % late in the run — five GPU runs of this test measured fp32-vs-fp32 window gaps of 0.3-23 % and bf16-vs-fp32 gaps of
    # 0.1-19 %, in no fixed order.  What the two-sample noise estimate can support: the mean loss over steps 50-199 within
    # max(10 %, 2 x the fp32 pair's gap), and no window further than max(25 %, 2.5 x the pair's largest window gap).
    overall = lambda x: float(x[FIRST:].mean())
    oa, ob, oc = overall(fa), overall(fb), overall(bf)
    lines = [f"mean loss over steps {FIRST}-{STEPS - 1}: fp32 {oa:.4f}  fp32' {ob:.4f}  bf16 {oc:.4f}  (gaps {abs(ob - oa) / oa:.4f} / {abs(oc - oa) / oa:.4f})"]
    _write("bf16_trajectory_mean.txt", lines)
    assert abs(oc - oa) / oa <= max(0.10, 2.0 * abs(ob - oa) / oa), (oa, ob, oc)
    for g_ in gap:
        assert g_ <= max(0.25, 2.5 * noise.max()), (gap, noise)

    # ---- the trunk comparison of tests/test_step_parity.py on the TRAINED weights (2 scenes of the first batch) ------------
    case = dict(gs=gs, state=final_a, batch_np={k: (v[:2] if k not in ("lang_fea", "lang_emb") else v[:16]) for k, v in batches_np[0].items()})
    g = torch.Generator().manual_seed(5)
    shapes = [(2, 256, 1024), (2, 1024, 3), (2, 256, 1024)]     # fp2_features, vote_xyz, vote_features
    cot = [torch.randn(s, generator=g, dtype=torch.float64) for s in shapes]
    runs = OrderedDict()
    for name, dt, compact in (("fp32", None, False), ("bf16 padded", torch.bfloat16, False), ("bf16 distinct", torch.bfloat16, True),
                              ("autocast literal", torch.bfloat16, False)):
        runs[name] = _gpu_trunk(gs, case, dt, compact, cot, literal_autocast=(name == "autocast literal"))
    ebp = _per_block(runs["bf16 padded"][0], runs["fp32"][0])
    ebd = _per_block(runs["bf16 distinct"][0], runs["fp32"][0])
    eal = _per_block(runs["autocast literal"][0], runs["fp32"][0])
    fro = lambda a, b_: float((a.double() - b_.double()).norm() / b_.double().norm())
    out_bd = [fro(a, b_) for a, b_ in zip(runs["bf16 distinct"][1], runs["fp32"][1])]
    out_al = [fro(a, b_) for a, b_ in zip(runs["autocast literal"][1], runs["fp32"][1])]
    lines = [f"trunk (backbone + voting) backward with a fixed cotangent on the weights after {STEPS} fp32 training steps, 2 scenes x "
             "40000 points (random initialisation: profiles/r03_step_parity_trunk.txt — outputs 1.0e-01 1.4e-02 1.2e-01, SA gradients 0.73)",
             "outputs (fp2_features, vote_xyz, vote_features), Frobenius error vs the fp32 kernels: bf16 distinct " +
             " ".join(f"{e:.1e}" for e in out_bd) + " | autocast literal sequence " + " ".join(f"{e:.1e}" for e in out_al),
             f"{'block':22s} {'bf16 padded vs fp32':>19s} {'bf16 distinct vs fp32':>21s} {'autocast literal vs fp32':>24s}"]
    for k in TRUNK_BLOCKS:
        lines.append(f"{k:22s} {ebp[k]:19.2e} {ebd[k]:21.2e} {eal[k]:24.2e}")
    _write("step_parity_trunk_trained.txt", lines)
    for k in TRUNK_BLOCKS:
        assert ebd[k] <= 1.5 * ebp[k] + 1e-2, (k, ebd[k], ebp[k])
        assert ebd[k] <= 1.25 * eal[k] + 1e-2, (k, ebd[k], eal[k])
    for a, b_ in zip(out_bd, out_al):
        assert a <= 1.5 * b_ + 1e-3, (out_bd, out_al)
