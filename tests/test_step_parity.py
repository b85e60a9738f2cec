"""Whole-step parity anchored OUTSIDE the product, at the cfg2 scene size (40 000 points, 256 proposals, 8 sentences).

  * fp32 step (exact-fp32 MFMA everywhere, padded rows, dropout off, fixed coin) on the GPU  vs  oracle/baseline.CpuStep —
    the C restatement of the reference's geometry kernels + PyTorch-CPU autograd through the literal op sequence
    (group -> 1x1 conv -> BatchNorm2d -> ReLU -> max-pool, unfused attention, batched torch loss) + torch.optim.AdamW —
    on identical weights: the loss, the gradient of EVERY parameter block, the BatchNorm running statistics and the
    parameters after one AdamW step.
  * the bf16 / distinct-row timing configuration  vs  that fp32 step on the same weights, with stated per-block bounds
    (replaces the flat 15 % / 2e-2 tolerances of rounds 1-2).

Both tests write their per-block tables to gpurun_out/step_parity_*.txt (copied into DESIGN.md §2).
"""
import importlib
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOCKS = ("backbone_net.sa1", "backbone_net.sa2", "backbone_net.sa3", "backbone_net.sa4", "backbone_net.fp1",
          "backbone_net.fp2", "vgen", "proposal.vote_aggregation", "proposal.proposal", "relation", "match.grounding_cross_attn.0",
          "match.grounding_cross_attn.1", "match.match", "constrast")
SCENES, POINTS = 2, 40000


def _dropout_off(model):
    model.eval()
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.train()


def _block_of(name):
    best = ""
    for b in BLOCKS:
        if name.startswith(b + ".") and len(b) > len(best):
            best = b
    return best or name.split(".")[0]


def _per_block(named_a, named_b):
    """{block: relative Frobenius error of the concatenated tensors of the block}."""
    num, den = OrderedDict(), OrderedDict()
    for n, a in named_a.items():
        b = named_b[n]
        blk = _block_of(n)
        num[blk] = num.get(blk, 0.0) + float((a.double() - b.double()).pow(2).sum())
        den[blk] = den.get(blk, 0.0) + float(b.double().pow(2).sum())
    return OrderedDict((k, (num[k] / max(den[k], 1e-300)) ** 0.5) for k in num)


def _write(name, lines):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, name), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


def _gpu_step(gs, batch_np, state, sa_dtype, lr=1e-3, **kw):
    devc = torch.device("cuda:0")
    step = gs.GroundingStep(devc, epoch=50, lr=lr, sa_dtype=sa_dtype, **kw)
    if state is not None:
        missing, unexpected = step.model.load_state_dict(state, strict=False)  # parameters; buffers at constructor values
        assert not unexpected and all(("running_" in k or "num_batches" in k) for k in missing)
    _dropout_off(step.model)
    before = {n: p.detach().clone() for n, p in step.model.named_parameters()}
    batch = gs.batch_to_device(batch_np, devc)
    batch["random"] = torch.tensor(0.75, device=devc)  # the coin of match_module.py:97 / loss_grounding.py:249: no paste, no gating
    loss = float(step.run(batch))
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().clone().cpu() for n, p in step.model.named_parameters() if p.grad is not None}
    after = {n: p.detach().clone().cpu() for n, p in step.model.named_parameters()}
    bufs = {n: b.detach().clone().cpu() for n, b in step.model.named_buffers()}
    return step, loss, grads, {n: v.cpu() for n, v in before.items()}, after, bufs, step._last_out


@pytest.fixture(scope="module")
def case():
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    synth = importlib.import_module("3dvlp_amd.synth")
    batch_np = synth.make_batch(0, SCENES, num_points=POINTS, lang_num_max=8)
    # A random-init model predicts ~2 m boxes around surface points: no proposal reaches IoU 0.25 with a referred box, the
    # reference loss has no positive label and relation / match / contrast receive EXACTLY zero gradient (measured).  So
    # the referred boxes of this case are taken from a throw-away forward of the same initial model: sentence j of a scene
    # refers to the box proposal (37 j + 11) predicts, enlarged by 1.3 (IoU 0.455 with that proposal: well inside the
    # positive side of the 0.25 threshold, not on a knife edge).  Inputs stay identical on both sides of the comparison.
    devc = torch.device("cuda:0")
    probe = gs.GroundingStep(devc, epoch=50, lr=0.0)
    _dropout_off(probe.model)
    with torch.no_grad():
        pb = gs.batch_to_device(batch_np, devc)
        pb["random"] = torch.tensor(0.75, device=devc)
        _, d0 = probe.forward_loss(pb)
    pick = [(37 * j + 11) % d0["pred_center"].shape[1] for j in range(8)]
    mean0 = synth.mean_size_arr()[0]
    batch_np["ref_center_label_list"] = d0["pred_center"][:, pick].float().cpu().numpy().copy()
    batch_np["ref_size_class_label_list"] = np.zeros((SCENES, 8), np.int64)
    batch_np["ref_size_residual_label_list"] = (1.3 * d0["pred_size"][:, pick].float().cpu().numpy() - mean0).astype(np.float32)
    del probe, pb, d0
    step, loss, grads, before, after, bufs, out = _gpu_step(gs, batch_np, None, None)
    state = {k: v.detach().clone().cpu() for k, v in before.items()}
    # the full initial state (parameters + buffers before the step): parameters from `before`, buffers are at their
    # constructor values (running_mean 0, running_var 1, counters 0), which a fresh model also has
    return dict(gs=gs, batch_np=batch_np, loss=loss, grads=grads, before=before, after=after, bufs=bufs, state=state,
                inds={k: out[k].cpu().numpy() for k in ("sa1_inds", "sa2_inds", "sa3_inds", "sa4_inds", "fp2_inds", "seed_inds")
                      if k in out})


def _cpu_step(case, dt, perturb=0):
    from oracle import baseline
    cpu = baseline.CpuStep(lr=1e-3, dtype=dt)
    missing, unexpected = cpu.net.load_state_dict({k: v.to(dt) for k, v in case["state"].items()}, strict=False)
    assert not unexpected and all(("running_" in k or "num_batches" in k) for k in missing), (missing, unexpected)
    _dropout_off(cpu.net)
    b = dict(case["batch_np"])
    if perturb:
        pc = b["point_clouds"].astype(np.float64)
        pc[..., 3:] *= 1 + 1e-6 * np.random.default_rng(perturb).standard_normal(pc[..., 3:].shape)
        b["point_clouds"] = pc
    loss = cpu.step(baseline.to_torch(b, SCENES, dt))  # random = 0.75, istrain = [1]
    return dict(loss=loss, grads={n: p.grad.detach().clone() for n, p in cpu.net.named_parameters() if p.grad is not None},
                after={n: p.detach().clone() for n, p in cpu.net.named_parameters()},
                bufs={n: b_.detach().clone() for n, b_ in cpu.net.named_buffers()})


def test_fp32_step_equals_cpu_step_at_cfg2_scene_size(case):
    """loss, every gradient block, BN running statistics and the post-AdamW parameters: GPU fp32 step vs CpuStep.

    Yardstick: CpuStep in DOUBLE precision (geometry stays the fp32 C restatement).  The loss and the running statistics
    are smooth in the inputs and are held to 1e-4.  The GRADIENT of this network at 40 000 points is not: the step takes
    ~10^8 ReLU / max-pool decisions, some pre-activation always sits within fp32 round-off of zero, and whichever side it
    falls changes a whole row's contribution to every upstream gradient (tools/step_parity_detail.py traced the round-3
    case to ONE element of the voting module's second layer: a 7e-6 relative perturbation of its input, applied in fp64,
    moves d(bn2.bias) by 2.5e-3 and everything upstream with it; tools/find_smooth_batch.py: a 1e-6 input perturbation
    moves the fp64 gradient by 1.6e-3 ... 1.4e-1 depending on the batch).  So the bound on each block is measured, not
    assumed: `noise` = change of the fp64 gradient itself under two 1e-6 relative perturbations of the input features, and
    the GPU must be within 1e-3 + 3 x noise of the fp64 gradient.  CpuStep in fp32 (the literal op sequence at the
    reference's own precision) is listed beside it."""
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref, c32 = _cpu_step(case, torch.float64), _cpu_step(case, torch.float32)
    pert = [_cpu_step(case, torch.float64, perturb=t)["grads"] for t in (1, 2)]
    loss_cpu, cgrads, cafter, cbufs = ref["loss"], ref["grads"], ref["after"], ref["bufs"]

    lines = [f"fp32 GPU step vs oracle/baseline.CpuStep (fp64), {SCENES} scenes x {POINTS} points, dropout off, coin 0.75",
             f"loss  gpu {case['loss']:.8f}  cpu64 {loss_cpu:.8f}  rel {abs(case['loss'] - loss_cpu) / abs(loss_cpu):.2e}"
             f"   (cpu32 {c32['loss']:.8f})"]
    assert abs(case["loss"] - loss_cpu) <= 1e-4 * abs(loss_cpu), (case["loss"], loss_cpu)
    # the same parameters receive a gradient on both sides (the optimiser skips the rest, like torch.optim does)
    assert set(case["grads"]) == set(cgrads), set(case["grads"]) ^ set(cgrads)
    gerr = _per_block(case["grads"], cgrads)
    g32 = _per_block(c32["grads"], cgrads)
    n1, n2 = _per_block(pert[0], cgrads), _per_block(pert[1], cgrads)
    noise = OrderedDict((k, max(n1[k], n2[k])) for k in n1)
    gnorm = OrderedDict()
    for n, g in cgrads.items():
        gnorm[_block_of(n)] = gnorm.get(_block_of(n), 0.0) + float(g.double().pow(2).sum())
    # parameters after AdamW: compared as the UPDATE p1 - p0 too (the first AdamW step is lr * g / (|g| + eps): an element
    # whose gradient is at round-off level may move by up to 2 lr the other way, so this is looser than the gradient itself)
    upd_g = {n: case["after"][n] - case["before"][n] for n in cgrads}
    upd_c = {n: cafter[n] - case["before"][n].double() for n in cgrads}
    uerr = _per_block(upd_g, upd_c)
    perr = _per_block({n: case["after"][n] for n in cafter}, cafter)
    fbufs = {n: b for n, b in cbufs.items() if b.dtype.is_floating_point}
    berr = _per_block({n: case["bufs"][n] for n in fbufs}, fbufs)
    lines.append(f"{'block':34s} {'|grad|':>9s} {'grad gpu':>9s} {'grad cpu32':>10s} {'fp64 noise':>10s} {'adamw upd':>10s} "
                 f"{'params':>9s} {'bn running':>10s}")
    for k in gerr:
        lines.append(f"{k:34s} {gnorm[k] ** 0.5:9.2e} {gerr[k]:9.2e} {g32[k]:10.2e} {noise[k]:10.2e} {uerr[k]:10.2e} "
                     f"{perr[k]:9.2e} {berr.get(k, float('nan')):10.2e}")
    over = [k for k, e in gerr.items() if e > 1e-4]
    lines.append("gradient blocks above 1e-4: " + (", ".join(f"{k} ({gerr[k]:.1e})" for k in over) or "none"))
    _write("step_parity_fp32_vs_cpu.txt", lines)
    for k in BLOCKS:  # every block of the step really receives gradient in this case
        assert gnorm[k] > 0, k
    for k, e in gerr.items():
        assert e <= 1e-3 + 3 * noise[k], (k, e, noise[k])
    for k, e in perr.items():  # lr = 1e-3 on weights of size ~5e-2: a fully wrong update direction would read ~4e-2
        assert e <= 1e-3 + 0.1 * min(1.0, gerr.get(k, 0.0)), (k, e)
    for k, e in berr.items():
        assert e <= 1e-4, (k, e)
    for n, b in cbufs.items():
        if not b.dtype.is_floating_point:
            assert int(case["bufs"][n]) == int(b), n  # every BatchNorm counted this step exactly once on both sides


TRUNK_BLOCKS = BLOCKS[:7]


def _gpu_trunk(gs, case, sa_dtype, compact, cot, literal_autocast=False):
    """Backbone + voting on the GPU with a FIXED cotangent: (per-parameter gradients, outputs).  literal_autocast: the
    reference's literal op sequence (group -> 1x1 conv -> BatchNorm -> ReLU -> max-pool, Conv1d / BatchNorm1d stacks:
    every `fused` switch off) under torch.autocast(bfloat16) — the precision the reference itself would have in bf16."""
    ext = importlib.import_module("3dvlp_amd._lib")
    ml = importlib.import_module("3dvlp_amd.mfma_linear")
    devc = torch.device("cuda:0")
    os.environ["VLP3D_SA_COMPACT"] = "1" if compact else "0"
    try:
        torch.manual_seed(0)
        net = gs.GroundingNet().to(devc)
    finally:
        os.environ.pop("VLP3D_SA_COMPACT", None)
    net.load_state_dict(case["state"], strict=False)
    _dropout_off(net)
    for m in net.modules():
        if hasattr(m, "mlp_dtype"):
            m.mlp_dtype = None if literal_autocast else sa_dtype
        if literal_autocast and hasattr(m, "fused"):
            m.fused = False
    batch = gs.batch_to_device(case["batch_np"], devc)
    import contextlib
    amp = torch.autocast(device_type="cuda", dtype=torch.bfloat16) if literal_autocast else contextlib.nullcontext()
    with ml.bf16_mma(sa_dtype == torch.bfloat16 and not literal_autocast):
        with amp:
            d = net.backbone_net(dict(batch))
            f = d["fp2_features"]
            if literal_autocast:
                vx, vf = net.vgen(d["fp2_xyz"], f)
                vf = vf.div(torch.norm(vf, p=2, dim=1).unsqueeze(1))
            else:
                vx, vf = net.vgen.forward_normalized(d["fp2_xyz"], f)
        outs = (f.float(), vx.float(), vf.float())
        with ext.deferred_slab_reduce():
            sum((o * c.to(devc)).sum() for o, c in zip(outs, cot)).backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().float().cpu() for n, p in net.named_parameters() if p.grad is not None}
    return grads, [o.detach().float().cpu() for o in outs], {k: d[k].cpu().numpy() for k in ("sa1_inds", "sa2_inds")}


def test_trunk_backward_fp32_vs_cpu_and_bf16_bounds(case):
    """Backbone + voting (SA1-4, FP1-2, vgen, L2 norm) at the cfg2 scene size with a FIXED cotangent — the part of the step
    that is a continuous function of the dense arithmetic (the whole step is not: the vote-aggregation FPS runs on learned
    votes, so any rounding difference re-draws the proposals; bf16 vs fp32 WHOLE-step gradients differ by ~100 %, also
    between the padded and the distinct-row bf16 evaluations, measured in round 3, and say nothing about the kernels).
      * fp32 kernels vs CpuStep.trunk in fp64: per-block gradient error <= 1e-3 + 3 x noise (noise as above);
      * bf16 storage + bf16 MFMA, distinct rows (the timing configuration) and padded rows vs the fp32 kernels: per block,
        distinct-row <= 1.5 x padded-row + 1e-2 and <= 1.25 x the error of the reference's literal op sequence under bf16
        autocast + 1e-2 (table: profiles/r03_step_parity_trunk.txt) — this replaces the flat 15 % / 2e-2 tolerances of
        rounds 1-2 with a measured yardstick."""
    from oracle import baseline
    gs = case["gs"]
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    g = torch.Generator().manual_seed(5)
    cot = None
    cpu_runs = []
    for dt, perturb in ((torch.float64, 0), (torch.float64, 1), (torch.float64, 2), (torch.float32, 0)):
        cpu = baseline.CpuStep(dtype=dt)
        cpu.net.load_state_dict({k: v.to(dt) for k, v in case["state"].items()}, strict=False)
        _dropout_off(cpu.net)
        b = dict(case["batch_np"])
        if perturb:
            pc = b["point_clouds"].astype(np.float64)
            pc[..., 3:] *= 1 + 1e-6 * np.random.default_rng(perturb).standard_normal(pc[..., 3:].shape)
            b["point_clouds"] = pc
        outs = cpu.trunk(baseline.to_torch(b, SCENES, dt))
        if cot is None:
            cot = [torch.randn(o.shape, generator=g, dtype=torch.float64) for o in outs]
        sum((o * c.to(dt)).sum() for o, c in zip(outs, cot)).backward()
        cpu_runs.append(({n: p.grad.detach().clone() for n, p in cpu.net.named_parameters() if p.grad is not None},
                         [o.detach() for o in outs]))
    ref, ref_out = cpu_runs[0]
    noise = {k: max(_per_block(cpu_runs[1][0], ref)[k], _per_block(cpu_runs[2][0], ref)[k]) for k in TRUNK_BLOCKS}
    g32 = _per_block(cpu_runs[3][0], ref)
    runs = OrderedDict()
    for name, dt, compact in (("fp32", None, False), ("bf16 padded", torch.bfloat16, False), ("bf16 distinct", torch.bfloat16, True),
                              ("autocast literal", torch.bfloat16, False)):
        runs[name] = _gpu_trunk(gs, case, dt, compact, cot, literal_autocast=(name == "autocast literal"))
        for k, v in runs[name][2].items():
            assert (v == runs["fp32"][2][k]).all(), (name, k)  # geometry does not depend on the dense precision
    e32 = _per_block(runs["fp32"][0], ref)
    ebp = _per_block(runs["bf16 padded"][0], runs["fp32"][0])
    ebd = _per_block(runs["bf16 distinct"][0], runs["fp32"][0])
    eal = _per_block(runs["autocast literal"][0], runs["fp32"][0])
    fro = lambda a, b_: float((a.double() - b_.double()).norm() / b_.double().norm())
    lines = [f"trunk (backbone + voting) backward with a fixed cotangent, {SCENES} scenes x {POINTS} points",
             "outputs (fp2_features, vote_xyz, vote_features), Frobenius error: fp32 kernels vs cpu64 " +
             " ".join(f"{fro(a, b_):.1e}" for a, b_ in zip(runs["fp32"][1], ref_out)) + " | bf16 distinct vs fp32 kernels " +
             " ".join(f"{fro(a, b_):.1e}" for a, b_ in zip(runs["bf16 distinct"][1], runs["fp32"][1])) +
             " | autocast literal sequence vs fp32 kernels " +
             " ".join(f"{fro(a, b_):.1e}" for a, b_ in zip(runs["autocast literal"][1], runs["fp32"][1])),
             f"{'block':22s} {'fp32 vs cpu64':>13s} {'cpu32 vs cpu64':>14s} {'fp64 noise':>10s} | {'bf16 padded vs fp32':>19s} "
             f"{'bf16 distinct vs fp32':>21s} {'autocast literal vs fp32':>24s}"]
    for k in TRUNK_BLOCKS:
        lines.append(f"{k:22s} {e32[k]:13.2e} {g32[k]:14.2e} {noise[k]:10.2e} | {ebp[k]:19.2e} {ebd[k]:21.2e} {eal[k]:24.2e}")
    _write("step_parity_trunk.txt", lines)
    assert set(runs["fp32"][0]) == set(ref)
    for k in TRUNK_BLOCKS:
        assert e32[k] <= 1e-3 + 3 * noise[k], (k, e32[k], noise[k])
        assert ebd[k] <= 1.5 * ebp[k] + 1e-2, (k, ebd[k], ebp[k])
        # the yardstick of the bf16 configuration, as for the forward pass (tests/test_composed_parity.py): the precision the
        # reference's own op sequence has under bf16 autocast, on the same weights, scenes and cotangent
        assert ebd[k] <= 1.25 * eal[k] + 1e-2, (k, ebd[k], eal[k])
