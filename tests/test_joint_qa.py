"""BASELINE cfg5 — the ScanQA + grounding joint step (models/jointnet/jointnet.py:109-110, 217-218 `self.answer`;
lib/loss_helper/loss_joint.py:118-119, 219-220 + loss_answering.py:2-16): the answer head attached to the grounding step.

CPU:  the product's answer loss (torch branch) == the oracle's restatement, both target forms.
GPU:  the BCE kernel vs torch fp64; one joint step vs oracle/baseline.CpuStep (fp64) on identical weights — total loss,
      answer loss and the answer head's gradients; the captured, pipelined bf16 step at 80 000-point scenes."""
import importlib

import numpy as np
import pytest
import torch

from oracle import losses as olosses

NUM_ANSWERS = 8864  # ScanQA's answer vocabulary (models/vqa/qa_module.py:142-144)


def test_answer_loss_torch_form_equals_oracle():
    L = importlib.import_module("3dvlp_amd.losses")
    rng = np.random.default_rng(0)
    x = rng.normal(0, 2, (16, 50)).astype(np.float32)
    t = (rng.random((16, 50)) > 0.9) * rng.choice([0.3, 0.6, 0.9, 1.0], size=(16, 50))
    got = L.compute_answer_classification_loss({"answer_scores": torch.from_numpy(x), "answer_cat_scores": torch.from_numpy(t)})
    assert abs(float(got) - olosses.answer_classification_loss(x, answer_cat_scores=t)) < 1e-5 * float(got)
    cat = rng.integers(0, 50, 16)
    got = L.compute_answer_classification_loss({"answer_scores": torch.from_numpy(x), "answer_cat": torch.from_numpy(cat)})
    assert abs(float(got) - olosses.answer_classification_loss(x, answer_cat=cat)) < 1e-5 * float(got)


def test_synthetic_answer_targets():
    synth = importlib.import_module("3dvlp_amd.synth")
    b = synth.make_batch(0, 2, num_points=1024, lang_num_max=3, num_answers=40)
    sc = b["answer_cat_scores"]
    assert sc.shape == (6, 40) and ((sc > 0).sum(1) >= 1).all() and ((sc > 0).sum(1) <= 3).all() and sc.max() <= 1.0
    assert (sc[np.arange(6), b["answer_cat"]] > 0).all()
    assert "answer_cat" not in synth.make_batch(0, 1, num_points=1024, lang_num_max=1)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols", [(64, NUM_ANSWERS), (3, 7), (16, 2048)])
def test_bce_logits_kernel_vs_fp64(rows, cols):
    L = importlib.import_module("3dvlp_amd.losses")
    torch.manual_seed(rows)
    x = (torch.randn(rows, cols, device="cuda") * 3).requires_grad_(True)
    t = (torch.rand(rows, cols, device="cuda") > 0.95).float() * 0.6
    loss = L.compute_answer_classification_loss({"answer_scores": x, "answer_cat_scores": t})
    (g,) = torch.autograd.grad(2.5 * loss, x)
    x6 = x.detach().double().requires_grad_(True)
    want = torch.nn.functional.binary_cross_entropy_with_logits(x6, t.double(), reduction="sum") / rows
    (w,) = torch.autograd.grad(2.5 * want, x6)
    assert abs(float(loss) - float(want)) < 1e-5 * float(want)
    assert float((g.double() - w).abs().max()) < 1e-5 * float(w.abs().max())


def _dropout_off(model):
    model.eval()
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.train()


@pytest.mark.gpu
def test_joint_qa_step_vs_cpu_step():
    """One fp32 joint step (dropout off, coin fixed) on the GPU vs CpuStep in double precision, same weights: total loss,
    answer loss, answer scores and the gradient of every answer-head parameter that takes part."""
    from oracle import baseline
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    synth = importlib.import_module("3dvlp_amd.synth")
    ml = importlib.import_module("3dvlp_amd.mfma_linear")
    devc = torch.device("cuda:0")
    batch_np = synth.make_batch(0, 2, num_points=8192, lang_num_max=4, num_answers=NUM_ANSWERS)
    step = gs.GroundingStep(devc, epoch=50, lr=0.0, use_answer=True, num_answers=NUM_ANSWERS)
    _dropout_off(step.model)
    state = {k: v.detach().clone().cpu() for k, v in step.model.named_parameters()}
    batch = gs.batch_to_device(batch_np, devc)
    batch["random"] = torch.tensor(0.75, device=devc)
    ml.FALLBACKS.clear()
    loss = float(step.run(batch))
    out = step._last_out
    assert "answer_scores" in out and out["answer_scores"].shape == (8, NUM_ANSWERS)
    # shapes outside the MFMA linear kernels are library GEMMs — and every one of them is COUNTED (nothing falls back
    # silently): at this toy batch (2 scenes x 4 sentences) the 8-row head layers and the 392-row K|V projections (rows not
    # a multiple of 32; 3136 rows at cfg2), plus the answer head's K = 512 / N = 1 / N = 8864 layers at any batch size
    assert dict(ml.FALLBACKS) == {"linear: 392 x 128 -> 256": 2, "linear: 2048 x 512 -> 1": 1, "linear: 8 x 128 -> 512": 1,
                                  "linear: 8 x 512 -> 128": 1, "linear: 8 x 128 -> 128": 1, "linear: 8 x 128 -> 8864": 1}, dict(ml.FALLBACKS)
    grads = {n: p.grad.detach().cpu() for n, p in step.model.named_parameters() if p.grad is not None and n.startswith("answer.")}
    assert {n.split(".")[1] for n in grads} == {"answer_cls", "attflat_visual"}  # the four other sub-modules never run (reference too)
    cpu = baseline.CpuStep(dtype=torch.float64, use_answer=True, num_answers=NUM_ANSWERS)
    cpu.net.load_state_dict({k: v.double() for k, v in state.items()}, strict=False)
    _dropout_off(cpu.net)
    loss_cpu = cpu.step(baseline.to_torch(batch_np, 2, torch.float64))
    assert abs(loss - loss_cpu) <= 1e-4 * abs(loss_cpu), (loss, loss_cpu)
    al, al_cpu = float(out["answer_loss"]), float(cpu.last["answer_loss"])
    assert abs(al - al_cpu) <= 1e-4 * al_cpu and al > 1.0, (al, al_cpu)
    assert abs(al - olosses.answer_classification_loss(out["answer_scores"].cpu().numpy(), batch_np["answer_cat_scores"])) < 1e-4 * al
    np.testing.assert_allclose(out["answer_scores"].cpu().numpy(), cpu.last["answer_scores"].detach().numpy(), rtol=1e-3, atol=2e-4)
    for n, g in grads.items():
        w = dict(cpu.net.named_parameters())[n].grad
        # (the glimpse-score bias has an exactly zero gradient — softmax over the proposals is shift invariant: absolute floor)
        assert float((g.double() - w).norm()) < 2e-3 * float(w.norm()) + 1e-6, (n, float((g.double() - w).norm()), float(w.norm()))


@pytest.mark.gpu
def test_cfg5_joint_qa_grounding_step_at_80k_points():
    """cfg5's shape: 80 000-point scenes, QA + grounding, captured + pipelined bf16 — finite, and both the total and the
    answer loss go down on a repeated batch."""
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    synth = importlib.import_module("3dvlp_amd.synth")
    devc = torch.device("cuda:0")
    batch = gs.batch_to_device(synth.make_batch(0, 4, num_points=80000, lang_num_max=8, num_answers=NUM_ANSWERS), devc)
    step = gs.GroundingStep(devc, epoch=50, sa_dtype=torch.bfloat16, use_graph=True, pipeline=True, use_answer=True,
                            num_answers=NUM_ANSWERS)
    tot, ans = [], []
    for _ in range(8):
        tot.append(float(step.run(batch)))
        ans.append(float(step._static_out["answer_loss"]))
    torch.cuda.synchronize()
    assert all(np.isfinite(tot)) and all(np.isfinite(ans)), (tot, ans)
    assert min(tot[4:]) < tot[0] and ans[-1] < ans[0], (tot, ans)
