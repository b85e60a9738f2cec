"""BASELINE cfg4 — the Scan2Cap caption head on the shared proposal features AS PART OF THE STEP
(models/jointnet/jointnet.py:103-104 `self.caption = TransformerDecoderModel(30522)`, :214-215; lib/loss_helper/loss_joint.py:
122-127, 222-223 `loss += cap_loss`; loss_captioning.py:25-80): `GroundingStep(use_caption=True)` — the head's parameters in
the flat parameter / gradient / AdamW buffers, cap_loss inside the captured graph, one backward, one all-reduce.

CPU:  the joint loss adds the caption term; the CPU step with the head attached (oracle/baseline.CpuStep + oracle/captioner.py).
GPU:  one fp32 joint grounding + caption step vs CpuStep in double precision on identical weights — total loss and caption
      loss to 1e-4, every gradient block (caption head and backbone included) under the measured noise bound of
      tests/test_step_parity.py; the captured, pipelined bf16 step at cfg4's size; two gloo ranks with the head attached."""
import importlib
import json
import os
import subprocess
import sys
from collections import OrderedDict

import numpy as np
import pytest
import torch

from tests.test_step_parity import BLOCKS, _block_of, _dropout_off, _per_block

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAP = dict(caption_mlm=False, transformer_dropout=0.0)   # deterministic: no MLM corruption, no dropout (p_attn set below)


def test_cpu_step_with_caption_head_adds_the_caption_loss():
    """CpuStep(use_caption=True): loss = grounding loss + cap_loss (loss_joint.py:222-223); the caption gradient reaches the
    head's parameters AND, through the indicator feature, the backbone."""
    from oracle import baseline
    synth = importlib.import_module("3dvlp_amd.synth")
    batch_np = synth.make_batch(0, 1, num_points=2048, lang_num_max=2, caption_tokens=12)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    kw = dict(CAP, N=2)
    cpu = baseline.CpuStep(dtype=torch.float64, use_caption=True, caption_kwargs=kw)
    _dropout_off(cpu.net)
    plain = baseline.CpuStep(dtype=torch.float64)
    plain.net.load_state_dict({k: v for k, v in cpu.net.state_dict().items() if not k.startswith("caption.")}, strict=True)
    _dropout_off(plain.net)
    b = baseline.to_torch(batch_np, 1, torch.float64)
    loss = cpu.forward_loss(b)
    base = plain.forward_loss(baseline.to_torch(batch_np, 1, torch.float64))
    cap_loss = float(cpu.last["cap_loss"].detach())
    assert 2.0 < cap_loss < 12.0                       # ~ln(30522) per real token, pads count 0
    assert abs(float(loss) - float(base) - cap_loss) < 1e-9 * float(loss)
    loss.backward()
    g = dict(cpu.net.named_parameters())
    assert float(g["caption.gen_w"].grad.abs().max()) > 0 and float(g["caption.qkv_w.1"].grad.abs().max()) > 0
    sa1 = g["backbone_net.sa1.mlp_module.layer0.conv.weight"].grad
    base.backward()
    sa1_plain = dict(plain.net.named_parameters())["backbone_net.sa1.mlp_module.layer0.conv.weight"].grad
    assert float((sa1 - sa1_plain).abs().max()) > 0   # the caption loss changes the backbone's gradient


@pytest.mark.gpu
def test_joint_caption_step_vs_cpu_step():
    """One fp32 joint grounding + caption step on the GPU (exact-fp32 MFMA, dropout off, fixed coin, no MLM corruption) vs
    CpuStep in double precision with the head by oracle/captioner.py, same weights: total loss and cap_loss to 1e-4, every
    gradient block within 1e-3 + 3 x noise (noise = change of the fp64 gradient under two 1e-6 input perturbations, the
    bound of tests/test_step_parity.py), the caption block — no ReLU / max-pool decisions between it and its loss —
    to 2e-3 outright."""
    from oracle import baseline
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    synth = importlib.import_module("3dvlp_amd.synth")
    devc = torch.device("cuda:0")
    S, L, T = 2, 4, 16
    batch_np = synth.make_batch(0, S, num_points=8192, lang_num_max=L, caption_tokens=T)
    step = gs.GroundingStep(devc, epoch=50, lr=0.0, use_caption=True, caption_kwargs=CAP)
    step.model.caption.p_attn = 0.0
    _dropout_off(step.model)
    assert any(p is q for p in step.layout.params for q in [step.model.caption.gen_w])     # the head lives in the flat buffers
    state = {k: v.detach().clone().cpu() for k, v in step.model.named_parameters()}
    batch = gs.batch_to_device(batch_np, devc)
    batch["random"] = torch.tensor(0.75, device=devc)
    loss = float(step.run(batch))
    out = step._last_out
    assert out["lang_cap_nll"].shape == (S * L, T - 1)
    grads = {n: p.grad.detach().cpu() for n, p in step.model.named_parameters() if p.grad is not None}
    flat = step.bucket.flat
    gw = step.model.caption.gen_w
    assert gw.grad.data_ptr() >= flat.data_ptr() and gw.grad.data_ptr() < flat.data_ptr() + flat.numel() * 4  # a view of the bucket

    torch.set_num_threads(min(16, os.cpu_count() or 1))

    def cpu_run(perturb=0):
        cpu = baseline.CpuStep(lr=0.0, dtype=torch.float64, use_caption=True, caption_kwargs=CAP)
        missing, unexpected = cpu.net.load_state_dict({k: v.double() for k, v in state.items()}, strict=False)
        assert not unexpected and all(("running_" in k or "num_batches" in k or k.endswith(".pe")) for k in missing), (missing, unexpected)
        _dropout_off(cpu.net)
        b = dict(batch_np)
        if perturb:
            pc = b["point_clouds"].astype(np.float64)
            pc[..., 3:] *= 1 + 1e-6 * np.random.default_rng(perturb).standard_normal(pc[..., 3:].shape)
            b["point_clouds"] = pc
        l_ = cpu.step(baseline.to_torch(b, S, torch.float64))
        return l_, {n: p.grad.detach().clone() for n, p in cpu.net.named_parameters() if p.grad is not None}, cpu.last
    loss_cpu, cg, last = cpu_run()
    assert abs(loss - loss_cpu) <= 1e-4 * abs(loss_cpu), (loss, loss_cpu)
    cl, cl_cpu = float(out["cap_loss"].detach()), float(last["cap_loss"].detach())
    assert abs(cl - cl_cpu) <= 1e-4 * cl_cpu and cl > 2.0, (cl, cl_cpu)
    assert torch.equal(out["match_idx"].cpu(), last["match_idx"])
    assert set(grads) == set(cg), set(grads) ^ set(cg)
    gerr = _per_block(grads, cg)
    noise = OrderedDict()
    pert = [cpu_run(t)[1] for t in (1, 2)]
    for pg in pert:
        for k, e in _per_block(pg, cg).items():
            noise[k] = max(noise.get(k, 0.0), e)
    lines = [f"joint grounding + caption step, fp32 GPU vs CpuStep fp64 ({S} scenes x 8192 points, {L} sentences x {T} tokens): "
             f"loss {loss:.8f} / {loss_cpu:.8f}, cap_loss {cl:.8f} / {cl_cpu:.8f}"]
    for k, e in gerr.items():
        lines.append(f"  {k:34s} grad err {e:9.2e}   fp64 noise {noise[k]:9.2e}")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "step_parity_caption.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    assert "caption" in gerr and all(b in gerr for b in BLOCKS[:9])
    for k, e in gerr.items():
        assert e <= 1e-3 + 3 * noise[k], (k, e, noise[k])
    # the head's own parameters, tensor by tensor (every tensor that takes part; src_* / sublayer-1 norms never run): the
    # head's input — the indicator feature — carries the backbone's noise, so the same kind of bound per tensor
    worst = ("", 0.0)
    for n, g in grads.items():
        if n.startswith("caption."):
            w = cg[n]
            rel = float((g.double() - w).norm()) / max(float(w.norm()), 1e-30)
            nz = max(float((pg[n] - w).norm()) for pg in pert) / max(float(w.norm()), 1e-30)   # this tensor's own fp64 noise
            worst = max(worst, (n, rel, nz), key=lambda t: t[1])
            assert rel <= 2e-3 + 3 * nz, (n, rel, nz)
    print("worst caption tensor", worst)


@pytest.mark.gpu
def test_cfg4_caption_step_captured_pipelined_bf16():
    """cfg4's size — 8 scenes x 40 000 points, 8 sentences x 32 tokens, 30 522 words, 6 layers — as the captured, pipelined bf16
    step (dropout and MLM corruption ON): finite, the head's parameters move with FlatAdamW, total and caption loss go down on
    a repeated batch."""
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    synth = importlib.import_module("3dvlp_amd.synth")
    devc = torch.device("cuda:0")
    batch = gs.batch_to_device(synth.make_batch(0, 8, num_points=40000, lang_num_max=8, caption_tokens=32), devc)
    step = gs.GroundingStep(devc, epoch=50, sa_dtype=torch.bfloat16, use_graph=True, pipeline=True, use_caption=True)
    before = step.model.caption.gen_w.detach().clone()
    tot, cap = [], []
    for _ in range(10):
        tot.append(float(step.run(batch)))
        cap.append(float(step._static_out["cap_loss"]))
    torch.cuda.synchronize()
    assert all(np.isfinite(tot)) and all(np.isfinite(cap)), (tot, cap)
    assert not torch.equal(before, step.model.caption.gen_w)
    assert min(cap[5:]) < cap[0] and min(tot[5:]) < tot[0], (tot, cap)
    assert step._head_range is not None     # the head's gradients are part of the early all-reduce piece


@pytest.mark.gpu
def test_two_gloo_ranks_with_the_caption_head_hold_identical_replicas():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, VLP3D_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-kernels", "--check-replicas", "--caption"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert "caption head" in d["config"]["workload"] and d["n_gpus"] == 2
    a, b = d["replica_param_checksums"]
    assert a == b, (a, b)
    assert d["config"]["loss"] > 0
