"""Caption head (3dvlp_amd/caption.py: the decoder models/caption_module/transformer_captioner.py:286-626 describes, as
jointnet builds it) against the CPU restatement oracle/captioner.py (parity unpinned: the reference module cannot be
constructed here, see its header); its kernels (csrc/caption.hip: attention core, fused generator + cross entropy;
csrc/add_norm.hip vlp3d_sum_norm_*: pre-norm residual stream) against torch fp64."""
import importlib

import numpy as np
import pytest
import torch

from oracle import captioner as ocap

cap = importlib.import_module("3dvlp_amd.caption")

V = 1000  # test vocabulary (the generator is a plain Linear: its width does not change the code path)


def make_endpoints(B, L, K, T, C=128, seed=0, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, V, (B, L, T), generator=g)
    ids[..., 0] = 101
    for b in range(B):
        for l in range(L):
            n = int(torch.randint(3, T, (1,), generator=g))
            ids[b, l, n:] = 0  # padding
    e = {"aggregated_vote_features": torch.randn(B, K, C, generator=g),
         "aggregated_vote_xyz": torch.randn(B, K, 3, generator=g) * 2,
         "input_ids": ids,
         "ref_center_label_list": torch.randn(B, L, 3, generator=g) * 2,
         "objectness_scores": torch.randn(B, K, 2, generator=g)}
    return {k: v.to(device) for k, v in e.items()}


def make_model(N, early_guide, seed=1):
    torch.manual_seed(seed)
    m = cap.TransformerDecoderModel(V, N=N, early_guide=early_guide, caption_mlm=False, transformer_dropout=0.0)
    with torch.no_grad():  # non-trivial norms and biases
        for n, p in m.named_parameters():
            if n.startswith(("norm_", "final_")) or n.split(".")[0].endswith("_b"):
                p.add_(torch.randn_like(p) * 0.1)
    return m


def sd64(m):
    return {k: v.detach().double().cpu() for k, v in m.state_dict().items()}


def test_state_dict_contract():
    """Keys / shapes of TransformerDecoderModel(30522) as transformer_captioner.py:286-365 registers them."""
    m = cap.TransformerDecoderModel(30522)
    sd = m.state_dict()
    want = {"model.tgt_embed.0.lut.weight": (30522, 128), "model.tgt_embed.1.pe": (1, 5000, 128),
            "model.generator.proj.weight": (30522, 128), "model.generator.proj.bias": (30522,),
            "model.decoder.norm.a_2": (128,), "model.decoder.norm.b_2": (128,)}
    for i in range(6):
        p = f"model.decoder.layers.{i}"
        for att in ("self_attn", "src_attn"):
            for j in range(4):
                want[f"{p}.{att}.linears.{j}.weight"] = (128, 128)
                want[f"{p}.{att}.linears.{j}.bias"] = (128,)
        want[f"{p}.feed_forward.w_1.weight"] = (512, 128)
        want[f"{p}.feed_forward.w_1.bias"] = (512,)
        want[f"{p}.feed_forward.w_2.weight"] = (128, 512)
        want[f"{p}.feed_forward.w_2.bias"] = (128,)
        for j in range(3):
            want[f"{p}.sublayer.{j}.norm.a_2"] = (128,)
            want[f"{p}.sublayer.{j}.norm.b_2"] = (128,)
    assert {k: tuple(v.shape) for k, v in sd.items()} == want
    assert sum(p.numel() for p in m.parameters()) == 9431866
    with pytest.raises(NotImplementedError):
        cap.TransformerDecoderModel(30522, use_transformer_encoder=True)
    with pytest.raises(NotImplementedError):
        cap.TransformerDecoderModel(30522, early_guide=False)   # cannot run as shipped (mask / input length mismatch)


@pytest.mark.gpu
def test_forward_train_eval_mode_matches_oracle():
    """Eval-mode forward == the restatement: the materialised log-probabilities, and the fused generator's per-token
    nll / arg-max against the same oracle tensor."""
    m = make_model(2, True).cuda().eval()
    m.materialize = True
    e = make_endpoints(2, 3, 16, 12, device="cuda")
    out = m(dict(e))
    want, idx = ocap.forward_train(sd64(m), {k: (v.double().cpu() if v.is_floating_point() else v.cpu())
                                             for k, v in e.items()}, N=2, early_guide=True)
    assert torch.equal(out["match_idx"].cpu(), idx)
    assert out["lang_cap"].shape == (6, 11, V)
    np.testing.assert_allclose(out["lang_cap"].detach().cpu().numpy(), want.numpy(), rtol=1e-4, atol=2e-5)
    tgt = e["input_ids"].view(6, -1)[:, 1:12].cpu()
    np.testing.assert_allclose(out["lang_cap_nll"].detach().cpu().numpy(), -torch.gather(want, 2, tgt.unsqueeze(-1)).squeeze(-1).numpy(),
                               rtol=1e-4, atol=2e-5)
    assert torch.equal(out["lang_cap_argmax"].cpu().long(), want.argmax(-1))
    loss, acc = cap.compute_cap_loss({**out, "input_ids": e["input_ids"]})
    want_loss = ocap.cap_loss(want, e["input_ids"].cpu(), out["good_bbox_masks"].cpu())
    assert abs(float(loss) - float(want_loss)) < 1e-4 * abs(float(want_loss))
    out.pop("lang_cap_nll")   # a caller that only has the reference's materialised tensor gets the same loss
    loss2, acc2 = cap.compute_cap_loss({**out, "input_ids": e["input_ids"]})
    assert abs(float(loss2) - float(want_loss)) < 1e-4 * abs(float(want_loss)) and float(acc2) == float(acc)
    assert 0.0 <= float(acc) <= 1.0


def test_mask_statistics():
    m = cap.TransformerDecoderModel(V, N=1, caption_mlm=True)
    ids = torch.randint(1, V, (64, 40))
    ids[:, 0] = 101
    ids[:, 30:] = 0
    out, masked = m.mask(ids, V)
    assert not masked[:, 0].any() and not masked[:, 30:].any()
    frac = masked[:, 1:30].float().mean().item()
    assert 0.05 < frac < 0.15
    assert (out[~masked] == ids[~masked]).all()
    assert ((out == 103) & masked).sum() > 0.6 * masked.sum()


@pytest.mark.gpu
def test_mlm_paths_run():
    m = cap.TransformerDecoderModel(V, N=1, caption_mlm=True).cuda().eval()
    e = make_endpoints(2, 2, 8, 10, device="cuda")
    r = m.forward_mlm(dict(e))
    assert r["lang_mlm_nll"].shape == (4, 9) and "lang_mlm" not in r and torch.isfinite(r["mlm_loss"])
    assert torch.isfinite(m(dict(e))["lang_cap_nll"]).all()   # forward_train with caption_mlm=True (raises as shipped)


def test_no_cpu_path():
    m = make_model(1, True).eval()
    with pytest.raises(RuntimeError, match="CPU not supported"):
        m(dict(make_endpoints(1, 1, 4, 6)), is_eval=True)


@pytest.mark.gpu
def test_greedy_eval_first_tokens_match_oracle():
    m = make_model(2, True).cuda().eval()
    m.max_des_len = 3
    e = make_endpoints(1, 1, 4, 6, device="cuda")
    out = m(dict(e), is_eval=True)["lang_cap"].cpu()
    e = {k: v.cpu() for k, v in e.items()}
    assert out.shape == (1, 4, 5) and (out[..., 0] == 101).all()
    # first generated token == argmax of the restatement on the [CLS] prefix
    sd = sd64(m)
    feats = e["aggregated_vote_features"].double().view(4, 1, -1)
    ys = torch.full((4, 1), 101)
    o = ocap.decode(sd, ys, feats, torch.tril(torch.ones(1, 2, 2, dtype=torch.bool)), N=2)
    logits = torch.nn.functional.linear(o[:, -1], sd["model.generator.proj.weight"], sd["model.generator.proj.bias"])
    assert torch.equal(out[0, :, 1], logits.argmax(-1))


# ---- GPU -------------------------------------------------------------------------------------------------------------
def _ref_sum_norm(x, y, keep, p, a, b, eps, std_mode):
    s = x if y is None else x + (y * keep / (1 - p) if p > 0 else y)
    mean = s.mean(-1, keepdim=True)
    if std_mode:
        n = a * (s - mean) / (s.std(-1, keepdim=True) + eps) + b
    else:
        n = a * (s - mean) / torch.sqrt(s.var(-1, unbiased=False, keepdim=True) + eps) + b
    return s, n


@pytest.mark.gpu
@pytest.mark.parametrize("D", [64, 128, 256])
@pytest.mark.parametrize("std_mode", [True, False])
@pytest.mark.parametrize("with_y,p", [(False, 0.0), (True, 0.0), (True, 0.1)])
def test_sum_norm_kernel_vs_fp64(D, std_mode, with_y, p):
    an = importlib.import_module("3dvlp_amd.add_norm")
    torch.manual_seed(D + int(std_mode))
    R = 203
    x = torch.randn(R, D, device="cuda") * 1.7 + 0.3
    y = torch.randn(R, D, device="cuda") if with_y else None
    a = (1 + 0.2 * torch.randn(D, device="cuda")).requires_grad_(True)
    b = (0.2 * torch.randn(D, device="cuda")).requires_grad_(True)
    x.requires_grad_(True)
    if y is not None:
        y.requires_grad_(True)
    eps = 1e-6 if std_mode else 1e-5
    keep = torch.empty(R, D, dtype=torch.uint8, device="cuda") if p > 0 else None
    s, n = an.sum_norm(x, y, a, b, eps, p, True, std_mode, keep)
    ws, wn = torch.randn(R, D, device="cuda"), torch.randn(R, D, device="cuda")
    ins = [t for t in (x, y, a, b) if t is not None]
    grads = torch.autograd.grad((s * ws).sum() + (n * wn).sum(), ins)
    x6, a6, b6 = (t.detach().double().requires_grad_(True) for t in (x, a, b))
    y6 = y.detach().double().requires_grad_(True) if y is not None else None
    k6 = keep.double() if keep is not None else None
    if keep is not None:
        assert 0.85 < keep.float().mean().item() < 0.95
    s6, n6 = _ref_sum_norm(x6, y6, k6, p, a6, b6, eps, std_mode)
    ins6 = [t for t in (x6, y6, a6, b6) if t is not None]
    want = torch.autograd.grad((s6 * ws.double()).sum() + (n6 * wn.double()).sum(), ins6)
    np.testing.assert_allclose(s.detach().cpu().numpy(), s6.detach().cpu().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(n.detach().cpu().numpy(), n6.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
    for got, w in zip(grads, want):
        scale = float(w.abs().max())
        np.testing.assert_allclose(got.cpu().numpy(), w.cpu().numpy(), rtol=1e-4, atol=2e-5 * max(scale, 1.0))


@pytest.mark.gpu
@pytest.mark.parametrize("early_guide,T", [(True, 16), (True, 17)])  # 16 x 17 rows: not a multiple of 32
def test_forward_train_gpu_matches_oracle_fwd_and_grads(early_guide, T):
    """Fused residual stream + MFMA linears (exact fp32 configuration) against the fp64 restatement: outputs 1e-4, every
    parameter gradient 1e-3 of its largest entry (train mode, dropout 0)."""
    m = make_model(3, early_guide).cuda().train()
    m.p_attn = 0.0   # the attention-probability dropout is 0.1 whatever transformer_dropout says (:297)
    # 16 sequences x T positions: T = 16 -> 256 rows, every projection on the MFMA kernels; T = 17 -> rows not a multiple
    # of 32, the projections take the library path (the residual-stream kernel has no row constraint)
    e = make_endpoints(4, 4, 32, T, device="cuda")
    feats = e["aggregated_vote_features"].requires_grad_(True)
    m.materialize = True
    out = m(dict(e))
    loss, _ = cap.compute_cap_loss({**out, "input_ids": e["input_ids"]})
    loss.backward()
    sd = {k: v.requires_grad_(v.is_floating_point() and not k.endswith(".pe")) for k, v in sd64(m).items()}
    e6 = {k: (v.detach().double().cpu() if v.is_floating_point() else v.cpu()) for k, v in e.items()}
    e6["aggregated_vote_features"].requires_grad_(True)
    want, idx = ocap.forward_train(sd, e6, N=3, early_guide=early_guide)
    assert torch.equal(out["match_idx"].cpu(), idx)
    np.testing.assert_allclose(out["lang_cap"].detach().cpu().numpy(), want.detach().numpy(), rtol=1e-4, atol=2e-5)
    wl = ocap.cap_loss(want, e6["input_ids"], out["good_bbox_masks"].cpu())
    assert abs(float(loss) - float(wl)) < 1e-4 * abs(float(wl))
    wl.backward()
    checked = 0
    got = m.reference_view({n: p.grad for n, p in m.named_parameters()})   # gradients under the reference's key names
    for k, v in sd.items():
        w = v.grad
        if k not in got:
            assert w is None or float(w.abs().max()) == 0.0, k
            continue
        scale = max(float(w.abs().max()), 1e-8)
        assert float((got[k].cpu().double() - w).abs().max()) <= 1e-3 * scale + 1e-7, k
        checked += 1
    assert checked >= 3 * 14 + 4
    gf = e6["aggregated_vote_features"].grad
    assert float((feats.grad.cpu().double() - gf).abs().max()) <= 1e-3 * float(gf.abs().max())


@pytest.mark.gpu
def test_caption_head_full_size_step_and_greedy():
    """jointnet's size: 30 522 words, 6 layers, 8 scenes x 8 sentences x 32 tokens, 256 proposals — one training step
    (dropout on, MLM corruption on) is finite and changes the weights; greedy decoding of 2 x 256 proposals runs."""
    torch.manual_seed(0)
    m = cap.TransformerDecoderModel(30522).cuda().train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    e = make_endpoints(8, 8, 256, 32, device="cuda")
    before = m.gen_w.detach().clone()
    out = m(dict(e))
    assert out["lang_cap_nll"].shape == (64, 31) and "lang_cap" not in out   # the 242 MB log-probability tensor is never built
    loss, acc = cap.compute_cap_loss({**out, "input_ids": e["input_ids"]})
    loss.backward()
    opt.step()
    # ~ln(30522) = 10.3 per real token at initialisation, averaged over ALL positions of good boxes (pads count 0)
    assert torch.isfinite(loss) and 2.0 < float(loss) < 12.0
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    assert not torch.equal(before, m.gen_w)
    m.eval()
    m.max_des_len = 4
    e2 = make_endpoints(2, 1, 256, 8, device="cuda")
    ys = m(dict(e2), is_eval=True)["lang_cap"]
    assert ys.shape == (2, 256, 6) and (ys[..., 0] == 101).all() and int(ys.max()) < 30522


@pytest.mark.gpu
def test_caption_head_trains_on_the_grounding_steps_proposal_features():
    """cfg4's composition: the caption head reads the SHARED proposal features of the detection + grounding forward
    (jointnet.py:214-215); one joint loss, one backward — gradients reach both the caption decoder and the backbone."""
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    synth = importlib.import_module("3dvlp_amd.synth")
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    step = gs.GroundingStep(dev, use_graph=False, pipeline=False, sa_dtype=torch.bfloat16)
    head = cap.TransformerDecoderModel(30522).to(dev).train()
    batch = gs.batch_to_device(synth.make_batch(0, 2, 40000, 8), dev)
    ids = torch.randint(1000, 30000, (2, 8, 32), device=dev)
    ids[..., 0] = 101
    ids[..., 20:] = 0
    batch["input_ids"] = ids
    step.bucket.zero()
    loss, d = step.forward_loss(batch)
    for k in ("aggregated_vote_features", "aggregated_vote_xyz", "objectness_scores"):
        assert k in d, k
    d = head(d)
    cap_loss, cap_acc = cap.compute_cap_loss(d)
    assert d["lang_cap_nll"].shape == (16, 31) and d["match_idx"].shape == (16,)
    total = loss + cap_loss
    step._backward(total)
    assert torch.isfinite(total)
    idle = {f"norm_{c}.{3 * i + 1}" for c in "ab" for i in range(6)}   # sublayer 1 = source attention: late guide only
    bad = [n for n, p in head.named_parameters()
           if not n.startswith("src_") and n not in idle and (p.grad is None or not torch.isfinite(p.grad).all())]
    assert not bad, bad
    assert float(head.gen_w.grad.abs().max()) > 0
    sa1 = step.model.backbone_net.sa1.mlp_module.layer0.conv.weight
    g = sa1.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0


# ---- csrc/caption.hip against torch fp64 ------------------------------------------------------------------------------
def _ref_attention(qkv, valid, n, T, H, causal):
    D = H * 16
    q, k, v = [t.reshape(n, T, H, 16).transpose(1, 2) for t in qkv.double().split(D, dim=1)]
    s = q @ k.transpose(-1, -2) / 4.0
    keep = valid.bool()[:, None, None, :].expand(n, H, T, T)
    if causal:
        keep = keep & torch.tril(torch.ones(T, T, dtype=torch.bool, device=qkv.device))
    s = s.masked_fill(~keep, -1e9)   # transformer_captioner.py:37-38
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(n * T, D)


@pytest.mark.gpu
@pytest.mark.parametrize("n,T,causal", [(5, 32, True), (3, 38, True), (4, 17, False), (2, 1, True), (2, 64, True)])
def test_cap_attention_kernel_vs_fp64(n, T, causal):
    torch.manual_seed(T)
    H = 8
    qkv = torch.randn(n * T, 3 * H * 16, device="cuda", requires_grad=True)
    valid = torch.rand(n, T, device="cuda") > 0.3
    valid[:, 0] = True   # the object-indicator position is always a key
    out = cap.cap_attention(qkv, valid.to(torch.uint8), n, T, H, causal, 0.0, True)
    g = torch.randn_like(out)
    (dq,) = torch.autograd.grad((out * g).sum(), qkv)
    q6 = qkv.detach().double().requires_grad_(True)
    want = _ref_attention(q6, valid, n, T, H, causal)
    (dw,) = torch.autograd.grad((want * g.double()).sum(), q6)
    np.testing.assert_allclose(out.detach().cpu().numpy(), want.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dq.cpu().numpy(), dw.cpu().numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.gpu
def test_cap_attention_dropout_is_a_fixed_mask_and_backward_is_its_derivative():
    """p = 0.1: the mask is a pure function of (seed word, call id, element) — same call, same output; roughly 10 % of the
    probability mass is dropped (outputs differ from p = 0, means agree); backward == central difference of forward."""
    an = importlib.import_module("3dvlp_amd.add_norm")
    torch.manual_seed(0)
    n, T, H = 6, 32, 8
    qkv = torch.randn(n * T, 3 * H * 16, device="cuda")
    valid = torch.ones(n, T, dtype=torch.uint8, device="cuda")
    seed = an.state(qkv.device)
    f = lambda x, p: cap._CapAttention.apply(x, valid, n, T, H, True, p, 77, seed)
    a, b, c = f(qkv, 0.1), f(qkv, 0.1), f(qkv, 0.0)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert abs(float(a.mean()) - float(c.mean())) < 0.02 and 0.02 < float((a - c).abs().mean() / c.abs().mean()) < 0.6
    x = qkv.clone().requires_grad_(True)
    g = torch.randn_like(a)
    (dx,) = torch.autograd.grad((f(x, 0.1) * g).sum(), x)
    u = torch.randn_like(qkv)
    u /= u.norm()
    h = 5e-2
    fd = float(((f(qkv + h * u, 0.1).double() - f(qkv - h * u, 0.1).double()) * g.double()).sum() / (2 * h))
    an_ = float((dx.double() * u.double()).sum())
    assert abs(fd - an_) < 2e-2 * abs(an_) + 1e-3, (fd, an_)


@pytest.mark.gpu
@pytest.mark.parametrize("R,Vv,bf", [(66, 1000, False), (240, 517, False), (1984, 30522, False), (1984, 30522, True), (32, 31, True)])
def test_vocab_ce_kernel_vs_fp64(R, Vv, bf):
    """Fused generator + log-softmax + cross entropy: per-row nll, arg-max, and the gradients of x, W, b against torch in
    fp64 (exact-fp32 products: 1e-4; bf16 operands: the rounding of the operands, 2^-8)."""
    ml = importlib.import_module("3dvlp_amd.mfma_linear")
    torch.manual_seed(R + Vv)
    x = (torch.randn(R, 128, device="cuda") * 0.7).requires_grad_(True)
    W = (torch.randn(Vv, 128, device="cuda") * 0.2).requires_grad_(True)
    b = (torch.randn(Vv, device="cuda") * 0.3).requires_grad_(True)
    tgt = torch.randint(0, Vv, (R,), device="cuda")
    coef = torch.rand(R, device="cuda") * (torch.rand(R, device="cuda") > 0.2)
    with ml.bf16_mma(bf):
        nll, arg = cap.vocab_nll(x, W, b, tgt)
    gx, gW, gb = torch.autograd.grad((nll * coef).sum(), (x, W, b))
    x6, W6, b6 = (t.detach().double().requires_grad_(True) for t in (x, W, b))
    logits = x6 @ W6.t() + b6
    want = torch.nn.functional.cross_entropy(logits, tgt, reduction="none")
    wx, wW, wb = torch.autograd.grad((want * coef.double()).sum(), (x6, W6, b6))
    tol = 2e-2 if bf else 1e-4
    np.testing.assert_allclose(nll.detach().cpu().numpy(), want.detach().cpu().numpy(), rtol=tol, atol=tol)
    if not bf:
        top2 = logits.topk(2, dim=-1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-4   # rows whose arg-max is not a round-off coin flip
        assert torch.equal(arg.long()[clear], logits.argmax(-1)[clear])
    else:
        assert float((arg.long() == logits.argmax(-1)).float().mean()) > 0.9
    for got, w in ((gx, wx), (gW, wW), (gb, wb)):
        scale = float(w.abs().max())
        assert float((got.double() - w).abs().max()) <= tol * scale + 1e-7, (float((got.double() - w).abs().max()), scale)
