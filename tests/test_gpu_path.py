"""GPU parity of the composed path: fused attention core, SA / FP modules, one full grounding step."""
import importlib

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _ref_sdpa(q, k, v, H, bias, mode, mask):
    """Unfused formulation (models/transformer/attention.py:63-75) in fp64 on the GPU."""
    B, nq, HD = q.shape
    nk = k.shape[1]
    D = HD // H
    qh = q.double().view(B, nq, H, D).permute(0, 2, 1, 3)
    kh = k.double().view(B, nk, H, D).permute(0, 2, 3, 1)
    vh = v.double().view(B, nk, H, D).permute(0, 2, 1, 3)
    att = torch.matmul(qh, kh) / np.sqrt(D)
    if bias is not None:
        att = att + bias.double() if mode == "add" else att * bias.double()
    if mask is not None:
        att = att.masked_fill(mask.view(B, 1, 1, nk) == 0, -10000)
    att = torch.softmax(att, -1)
    return torch.matmul(att, vh).permute(0, 2, 1, 3).reshape(B, nq, HD)


@pytest.mark.parametrize("nq,nk", [(256, 49), (256, 256), (40, 33), (1, 1), (70, 100)])
@pytest.mark.parametrize("mode", [None, "add", "mul"])
def test_fused_sdpa_forward_backward(nq, nk, mode):
    fa = importlib.import_module("3dvlp_amd.fused_attention")
    torch.manual_seed(nq * 1000 + nk)
    B, H, D = 3, 4, 32
    q = torch.randn(B, nq, H * D, device="cuda", requires_grad=True)
    k = torch.randn(B, nk, H * D, device="cuda", requires_grad=True)
    v = torch.randn(B, nk, H * D, device="cuda", requires_grad=True)
    bias = None
    if mode is not None:
        bias = (torch.randn(B, H, nq, nk, device="cuda") * (1.0 if mode == "add" else 0.5)).requires_grad_(True)
    mask = (torch.rand(B, nk, device="cuda") > 0.3).float()
    mask[:, 0] = 1
    for m in (None, mask):
        out = fa.sdpa(q, k, v, H, bias, mode or "add", None if m is None else m.view(B, 1, 1, nk))
        ref = _ref_sdpa(q, k, v, H, bias, mode, m)
        # north_star tolerance: 1e-4 relative for float features
        torch.testing.assert_close(out.double(), ref, rtol=1e-4, atol=2e-5)
        g = torch.randn_like(out)
        ins = [q, k, v] + ([bias] if bias is not None else [])
        got = torch.autograd.grad(out, ins, g)
        exp = torch.autograd.grad(ref, ins, g.double())
        for a, b in zip(got, exp):
            scale = b.abs().max().item()
            assert (a.double() - b.double()).abs().max().item() < 2e-4 * scale + 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("nq,nk,mode", [(256, 256, "add"), (256, 49, None), (70, 33, "mul")])
def test_fused_sdpa_bf16_mma_close_to_fp32(nq, nk, mode):
    """bf16 MFMA operands (timing configuration) vs the exact-fp32 core: forward and all gradients within bf16
    round-off (2^-8 relative to each tensor's scale), never bit-exact by construction."""
    fa = importlib.import_module("3dvlp_amd.fused_attention")
    g = torch.Generator(device="cpu").manual_seed(5)
    b, h = 3, 4
    q0 = torch.randn(b, nq, h * 32, generator=g).cuda()
    k0 = torch.randn(b, nk, h * 32, generator=g).cuda()
    v0 = torch.randn(b, nk, h * 32, generator=g).cuda()
    w0 = (torch.rand(b, h, nq, nk, generator=g) + 0.5).cuda() if mode else None
    mask = (torch.rand(b, nk, generator=g) > 0.2).float().cuda().view(b, 1, 1, nk)
    go = torch.randn(b, nq, h * 32, generator=g).cuda()
    res = []
    for bf in (False, True):
        q, k, v = (t.clone().requires_grad_(True) for t in (q0, k0, v0))
        w = w0.clone().requires_grad_(True) if mode else None
        out = fa.sdpa(q, k, v, h, w, mode or "add", mask, bf16_mma=bf)
        out.backward(go)
        res.append([out.detach(), q.grad, k.grad, v.grad] + ([w.grad] if mode else []))
    for a, c in zip(*res):
        scale = float(a.abs().max())
        assert float((a - c).abs().max()) < 3e-2 * scale + 1e-6
        assert float((a - c).norm()) < 1e-2 * float(a.norm()) + 1e-6
    assert not torch.equal(res[0][0], res[1][0])


@pytest.mark.parametrize("bf", [False, True])
@pytest.mark.parametrize("layout", ["qkv", "kv"])
def test_fused_sdpa_row_strided_views_equal_contiguous(layout, bf):
    """q / k / v as column blocks of ONE merged projection output (no copies in, ONE merged gradient buffer out) give
    the same result and gradients as contiguous tensors."""
    fa = importlib.import_module("3dvlp_amd.fused_attention")
    torch.manual_seed(11)
    B, n, H, HD = 4, 256, 4, 128
    nk = n if layout == "qkv" else 49
    merged = torch.randn(B, nk, (3 if layout == "qkv" else 2) * HD, device="cuda", requires_grad=True)
    qsep = torch.randn(B, n, HD, device="cuda", requires_grad=True)
    go = torch.randn(B, n, HD, device="cuda")
    if layout == "qkv":
        q, k, v = merged.split(HD, dim=-1)
    else:
        q = qsep
        k, v = merged.split(HD, dim=-1)
    out = fa.sdpa(q, k, v, H, bf16_mma=bf)
    out.backward(go)
    g_merged, g_q = merged.grad.clone(), (None if layout == "qkv" else qsep.grad.clone())
    qc, kc, vc = (t.detach().contiguous().requires_grad_(True) for t in (q, k, v))
    ref = fa.sdpa(qc, kc, vc, H, bf16_mma=bf)
    ref.backward(go)
    torch.testing.assert_close(out, ref, rtol=0, atol=0)
    exp = torch.cat([qc.grad, kc.grad, vc.grad], -1) if layout == "qkv" else torch.cat([kc.grad, vc.grad], -1)
    torch.testing.assert_close(g_merged, exp, rtol=0, atol=0)
    if g_q is not None:
        torch.testing.assert_close(g_q, qc.grad, rtol=0, atol=0)


def test_fused_sdpa_matches_oracle_numpy():
    fa = importlib.import_module("3dvlp_amd.fused_attention")
    rng = np.random.default_rng(0)
    B, H, D, nq, nk = 2, 4, 32, 64, 49
    q, k, v = (rng.normal(size=(B, n, H * D)).astype(np.float32) for n in (nq, nk, nk))
    out = fa.sdpa(dev(q), dev(k), dev(v), H).cpu().numpy()
    heads = lambda t, n: t.reshape(B, n, H, D).transpose(0, 2, 1, 3)
    ref, _ = orc.sdpa_core(heads(q, nq), heads(k, nk), heads(v, nk))
    np.testing.assert_allclose(out, ref.transpose(0, 2, 1, 3).reshape(B, nq, H * D), rtol=1e-4, atol=2e-5)


def _mlp_layers(mlp):
    out = []
    for layer in mlp:
        bn = layer.bn.bn
        out.append(dict(w=layer.conv.weight.detach().cpu().numpy()[:, :, 0, 0], gamma=bn.weight.detach().cpu().numpy(),
                        beta=bn.bias.detach().cpu().numpy(), mean=bn.running_mean.cpu().numpy(),
                        var=bn.running_var.cpu().numpy()))
    return out


@pytest.mark.parametrize("training", [True, False])
def test_sa_module_votes_forward_vs_oracle(training):
    pm = importlib.import_module("3dvlp_amd.pointnet2_modules")
    synth = importlib.import_module("3dvlp_amd.synth")
    torch.manual_seed(0)
    sc = [synth.make_scene(1000 + i, 4096) for i in range(2)]
    xyz = np.stack([s["xyz"] for s in sc])
    feat = np.stack([s["features"][:, :13].T for s in sc]).copy()
    sa = pm.PointnetSAModuleVotes(npoint=256, radius=0.4, nsample=32, mlp=[13, 32, 32, 64], use_xyz=True,
                                  normalize_xyz=True).cuda().train(training)
    layers = _mlp_layers(sa.mlp_module)  # running stats BEFORE the forward (eval) / unused (train)
    with torch.no_grad():
        new_xyz, new_feat, inds = sa(dev(xyz), dev(feat))
    r_xyz, r_feat, r_inds = orc.sa_module_votes(xyz, feat, layers, 256, 0.4, 32, training, normalize_xyz=True)
    assert (inds.cpu().numpy() == r_inds).all()
    assert (new_xyz.cpu().numpy() == r_xyz).all()
    np.testing.assert_allclose(new_feat.cpu().numpy(), r_feat, rtol=1e-4, atol=5e-5)


def test_fp_module_forward_vs_oracle():
    pm = importlib.import_module("3dvlp_amd.pointnet2_modules")
    rng = np.random.default_rng(1)
    torch.manual_seed(0)
    B, n, m, C1, C2 = 2, 200, 60, 16, 24
    unknown = rng.uniform(0, 3, (B, n, 3)).astype(np.float32)
    known = rng.uniform(0, 3, (B, m, 3)).astype(np.float32)
    uf = rng.normal(size=(B, C1, n)).astype(np.float32)
    kf = rng.normal(size=(B, C2, m)).astype(np.float32)
    fp = pm.PointnetFPModule(mlp=[C1 + C2, 32, 32]).cuda().train()
    layers = _mlp_layers(fp.mlp)
    with torch.no_grad():
        out = fp(dev(unknown), dev(known), dev(uf), dev(kf))
    ref = orc.fp_module(unknown, known, uf, kf, layers, training=True)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=5e-5)


def test_grounding_step_forward_backward_small():
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    synth = importlib.import_module("3dvlp_amd.synth")
    devc = torch.device("cuda:0")
    step = gs.GroundingStep(devc)
    batch = gs.batch_to_device(synth.make_batch(0, 2, num_points=8192, lang_num_max=2), devc)
    loss, d = step.forward_loss(batch)
    assert torch.isfinite(loss)
    # geometry of the first SA layer equals the oracle (bit-exact indices inside the composed model)
    xyz = batch["point_clouds"][..., :3].cpu().numpy()
    assert (d["sa1_inds"].cpu().numpy() == orc.furthest_point_sampling(xyz, 2048)).all()
    assert d["cluster_ref"].shape == (4, 256) and d["pred_bbox_corner"].shape == (2, 256, 8, 3)
    step.bucket.zero()
    loss.backward()
    step.bucket.collect()
    assert torch.isfinite(step.bucket.flat).all()
    touched = [n for n, p in step.model.named_parameters() if p.grad is not None and p.grad.abs().sum() > 0]
    # parameters outside the step's graph keep .grad None: the optimiser skips them like the reference's does
    untouched = {n for n, p in step.model.named_parameters() if p.grad is None}
    assert "match.box_con_proj.weight" in untouched and "constrast.nce_loss.tau" in untouched
    # relation / match are in the graph (gradient tensors exist); whether they are non-zero at random init depends on a
    # proposal reaching IoU >= 0.25 with a referred box (reference semantics: loss_grounding.py:258) — tests/test_losses.py
    # checks that gradient flow on a case built for it
    for in_graph in ("relation.self_attn.0.attention.fc_q.weight",
                     "match.grounding_cross_attn.1.enc_dec_attention.attention.fc_k.weight", "match.match.6.weight"):
        assert in_graph not in untouched, in_graph
    for must in ("backbone_net.sa1.mlp_module.layer0.conv.weight", "vgen.conv3.weight",
                 "proposal.vote_aggregation.mlp_module.layer2.conv.weight",
                 # heads that only the full loss reaches (box / size-distance, heading residual, semantic class)
                 "proposal.proposal.box_predictor.weight", "proposal.proposal.heading_reg_predictor.weight",
                 "proposal.proposal.sem_cls_predictor.weight"):
        assert must in touched, must
    # OCC/OSC are active (epoch 50); whether they carry gradient at random init depends on IoU>0.25 hits
    assert torch.isfinite(d["lang_con_loss"]) and torch.isfinite(d["iou_con_loss"])
    l0 = float(loss.detach())
    counters = {n: int(b) for n, b in step.model.named_buffers() if n.endswith("num_batches_tracked")}
    idle = {n for n in counters if n.startswith("match.lang_emb_proj.")}  # never runs (use_lang_emb off): stays 0 like the reference's
    assert len(idle) == 2 and all(counters[n] == 0 for n in idle)
    counters = {n: v for n, v in counters.items() if n not in idle}
    assert len(counters) > 20 and set(counters.values()) == {1}  # one forward so far, every BatchNorm that ran counted once
    for _ in range(3):
        l1 = float(step.run(batch).detach())
    assert np.isfinite(l1) and l1 < l0  # three AdamW steps on a fixed batch reduce the loss
    after = {n: int(b) for n, b in step.model.named_buffers() if n.endswith("num_batches_tracked") and n not in idle}
    ran = {v - counters[n] for n, v in after.items()}
    assert len(ran) == 1 and ran.pop() >= 3  # all counters advance together (the fused increment, graph replays included)


def test_group_rows_forward_backward_vs_oracle():
    pu = importlib.import_module("3dvlp_amd.pointnet2_utils")
    rng = np.random.default_rng(21)
    B, N, M, S, C, r = 2, 600, 40, 8, 12, 0.5
    xyz = rng.uniform(0, 2, (B, N, 3)).astype(np.float32)
    new_xyz = xyz[:, :M].copy()
    feat = rng.normal(size=(B, C, N)).astype(np.float32)
    g_feat, g_xyz, idx = orc.query_and_group(xyz, new_xyz, feat, r, S, use_xyz=True, normalize_xyz=True)
    exp = np.concatenate([g_feat[:, 3:], g_feat[:, :3], np.zeros((B, 1, M, S), np.float32)], 1)  # [feat|xyz|0]
    exp = exp.transpose(0, 2, 3, 1).reshape(B * M * S, C + 4)
    X, NX = dev(xyz).requires_grad_(True), dev(new_xyz).requires_grad_(True)
    F_pm = dev(feat.transpose(0, 2, 1)).requires_grad_(True)
    rows = pu.group_rows(X, NX, dev(idx), F_pm, r, torch.float32)
    assert (rows.detach().cpu().numpy() == exp).all()  # gather + the same fp32 (q-c)/r
    rows_bf = pu.group_rows(X, NX, dev(idx), F_pm, r, torch.bfloat16)
    np.testing.assert_allclose(rows_bf.detach().float().cpu().numpy(), exp, rtol=8e-3, atol=1e-6)
    g = rng.normal(size=exp.shape).astype(np.float32)
    rows.backward(dev(g))
    g4 = g.reshape(B, M, S, C + 4).transpose(0, 3, 1, 2)
    np.testing.assert_allclose(F_pm.grad.cpu().numpy().transpose(0, 2, 1),
                               orc.group_points_grad(np.ascontiguousarray(g4[:, :C]), idx, N), rtol=1e-5, atol=1e-5)
    gx = np.ascontiguousarray(g4[:, C:C + 3]) / np.float32(r)
    np.testing.assert_allclose(X.grad.cpu().numpy().transpose(0, 2, 1), orc.group_points_grad(gx, idx, N),
                               rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(NX.grad.cpu().numpy(), -gx.sum(3).transpose(0, 2, 1), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("mode", ["mfma", "rows"])
@pytest.mark.parametrize("training", [True, False])
def test_sa_module_fused_equals_reference_sequence(training, mode):
    """The fused SA paths (hand-written MFMA kernels / row-major torch) == the literal reference op sequence
    (same weights, fp32): forward within 1e-4, gradients norm-wise."""
    pm = importlib.import_module("3dvlp_amd.pointnet2_modules")
    synth = importlib.import_module("3dvlp_amd.synth")
    torch.manual_seed(0)
    sc = [synth.make_scene(1000 + i, 4096) for i in range(2)]
    xyz = dev(np.stack([s["xyz"] for s in sc]))
    feat = dev(np.stack([s["features"][:, :12].T for s in sc]).copy())
    sa = pm.PointnetSAModuleVotes(npoint=256, radius=0.4, nsample=32, mlp=[12, 64, 64, 128], use_xyz=True,
                                  normalize_xyz=True).cuda().train(training)
    import copy
    ref = copy.deepcopy(sa)
    ref.fused = False
    assert sa.fused == "mfma"
    sa.fused = mode
    sf = importlib.import_module("3dvlp_amd.sa_fused")
    assert sf.supported(12, [64, 64, 128], 32, 2 * 256 * 32)  # the MFMA kernels really cover this shape
    x1, f1 = xyz.clone().requires_grad_(True), feat.clone().requires_grad_(True)
    x2, f2 = xyz.clone().requires_grad_(True), feat.clone().requires_grad_(True)
    nx1, nf1, i1 = sa(x1, f1)
    nx2, nf2, i2 = ref(x2, f2)
    assert torch.equal(i1, i2) and torch.equal(nx1, nx2)
    torch.testing.assert_close(nf1, nf2, rtol=1e-4, atol=2e-5)
    g = torch.randn_like(nf2)
    (nf1 * g).sum().backward()
    (nf2 * g).sum().backward()
    torch.testing.assert_close(f1.grad, f2.grad, rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(x1.grad, x2.grad, rtol=1e-3, atol=1e-4)
    for (n, p), (_, q) in zip(sa.named_parameters(), ref.named_parameters()):
        err = (p.grad - q.grad).abs().max().item()
        scale = q.grad.abs().max().item()
        # train-mode BN makes weight gradients differences of large cancelling sums: compare norm-wise
        assert err <= 2e-3 * scale + 1e-5, (n, err, scale)
    if training:
        for (n, p), (_, q) in zip(sa.named_buffers(), ref.named_buffers()):
            torch.testing.assert_close(p.float(), q.float(), rtol=1e-4, atol=1e-6, msg=n)


def test_sa_module_mfma_bf16_close_to_fp32():
    """bf16 storage + bf16 MFMA variant of one SA module against its fp32 evaluation.  Yardstick (round 3, replaces the flat
    15 %): the reference's LITERAL op sequence (`fused=False`: group -> 1x1 conv -> BatchNorm2d -> ReLU -> max-pool) under
    torch.autocast(bfloat16) on the same weights and inputs — output and every gradient of the fused kernels must be within
    1.5 x that sequence's own error (+ 5e-3)."""
    import copy
    pm = importlib.import_module("3dvlp_amd.pointnet2_modules")
    synth = importlib.import_module("3dvlp_amd.synth")
    torch.manual_seed(0)
    sc = [synth.make_scene(1000 + i, 4096) for i in range(2)]
    xyz = dev(np.stack([s["xyz"] for s in sc]))
    feat = dev(np.stack([s["features"][:, :12].T for s in sc]).copy())
    sa = pm.PointnetSAModuleVotes(npoint=256, radius=0.4, nsample=32, mlp=[12, 64, 64, 128], use_xyz=True,
                                  normalize_xyz=True).cuda().train()
    lit = copy.deepcopy(sa)
    lit.fused = False
    g = None

    def run(mod, amp):
        nonlocal g
        f = feat.clone().requires_grad_(True)
        mod.zero_grad()
        if amp:
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                _, out, _ = mod(xyz, f)
        else:
            _, out, _ = mod(xyz, f)
        out = out.float()
        if g is None:
            g = torch.randn_like(out)
        (out * g).sum().backward()
        r = {"out": out.detach(), "dfeat": f.grad.clone()}
        r.update({n: p.grad.clone() for n, p in mod.named_parameters()})
        return r

    def rel(x, y):
        return ((x - y).norm() / (y.norm() + 1e-20)).item()

    ref = run(sa, False)
    fused, literal = run(sa, True), run(lit, True)
    ef = {k: rel(fused[k], ref[k]) for k in ref}
    el = {k: rel(literal[k], ref[k]) for k in ref}
    print("bf16 vs fp32 relative (Frobenius) errors, fused kernels | autocast literal sequence:",
          {k: (round(ef[k], 4), round(el[k], 4)) for k in ref})
    assert ef["out"] < 2e-2, ef
    for k in ref:
        assert ef[k] <= 1.5 * el[k] + 5e-3, (k, ef[k], el[k])



def test_relation_bias_fused_forward_backward():
    """Fused pairwise-geometry MLP kernel == the reference op sequence (torch, fp64) on the same weights."""
    det = importlib.import_module("3dvlp_amd.detection")
    torch.manual_seed(3)
    B, K = 3, 70
    m = det.RelationModule(num_proposals=K, det_channel=128).cuda()
    fc = m.self_attn_fc[0]
    with torch.no_grad():
        for p in fc.parameters():
            p.add_(0.05 * torch.randn_like(p))
    centre = torch.rand(B, K, 3, device="cuda") * 4
    out = det.relation_bias(centre, fc)
    import copy
    fc64 = copy.deepcopy(fc).double()
    c64 = centre.double()
    delta = c64[:, None, :, :] - c64[:, :, None, :]
    pair = torch.cat([delta, delta.pow(2).sum(-1, keepdim=True).sqrt()], dim=-1)
    ref = fc64(pair).permute(0, 3, 1, 2)
    torch.testing.assert_close(out.double(), ref, rtol=1e-4, atol=2e-5)
    g = torch.randn_like(out)
    got = torch.autograd.grad(out, list(fc.parameters()), g)
    exp = torch.autograd.grad(ref, list(fc64.parameters()), g.double())
    for (n, _), a, b in zip(fc.named_parameters(), got, exp):
        scale = b.abs().max().item()
        assert (a.double() - b).abs().max().item() < 2e-4 * scale + 1e-5, n


def test_relation_bias_backward_bf16_mma_close_to_fp64():
    """The bf16-MFMA form of the relation-bias backward (the step's timing configuration: mfma_linear.bf16_mma) against the
    fp64 op sequence: operands of the two 32 x 32 products and of the rank-32 updates are rounded to bf16 (2^-9 relative each),
    the sums over B*K*K pairs are fp32; the recomputed forward stays exact — parameter gradients within 1 % of their scale
    (0.2-0.5 % measured, tools/relbias_bf16_err.py; a bf16 layer-2 forward product gave 3-5 %)."""
    det = importlib.import_module("3dvlp_amd.detection")
    ml = importlib.import_module("3dvlp_amd.mfma_linear")
    torch.manual_seed(4)
    B, K = 3, 70
    m = det.RelationModule(num_proposals=K, det_channel=128).cuda()
    fc = m.self_attn_fc[1]
    with torch.no_grad():
        for p in fc.parameters():
            p.add_(0.05 * torch.randn_like(p))
    centre = torch.rand(B, K, 3, device="cuda") * 4
    with ml.bf16_mma(True):
        out = det.relation_bias(centre, fc)
    import copy
    fc64 = copy.deepcopy(fc).double()
    c64 = centre.double()
    delta = c64[:, None, :, :] - c64[:, :, None, :]
    pair = torch.cat([delta, delta.pow(2).sum(-1, keepdim=True).sqrt()], dim=-1)
    ref = fc64(pair).permute(0, 3, 1, 2)
    g = torch.randn_like(out)
    got = torch.autograd.grad(out, list(fc.parameters()), g)
    exp = torch.autograd.grad(ref, list(fc64.parameters()), g.double())
    for (n, _), a, b in zip(fc.named_parameters(), got, exp):
        scale = b.abs().max().item()
        assert (a.double() - b).abs().max().item() < 1e-2 * scale + 1e-4, (n, (a.double() - b).abs().max().item(), scale)


def _eval_dropout_train_bn(step):
    step.model.eval()  # no dropout: execution variants must agree numerically
    for m in step.model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.train()


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def test_split_backward_gradients_do_not_depend_on_side_stream_timing():
    """The captured, pipelined step finishes the weight-gradient-only work on the side stream while the main stream runs the
    backward of SA2 / SA1 (GroundingStep._capture).  Every consumer of those gradients must be ordered after the side
    stream's graph — round 3 found the flat-bucket copy of ALL parameters on the main stream beside it (stale head gradients
    whenever the side stream was late; it surfaced as a NaN at cfg5's size once in a full test run).  Two identical step
    objects with learning rate 0 (the weights never move, so the second step's gradient is a function of its batch alone)
    run batch A, then batch B; in one of them the side stream's deferred graph is held back by 30 ms in the second step.
    Both must leave batch B's gradient; a consumer that does not wait would leave batch A's in the head parameters."""
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    synth = importlib.import_module("3dvlp_amd.synth")
    devc = torch.device("cuda:0")
    batches = [gs.batch_to_device(synth.make_batch(4 * i, 2, num_points=8192, lang_num_max=2), devc) for i in range(2)]
    for b in batches:
        b["random"] = torch.tensor(0.25, device=devc)
    flats = []
    for delayed in (False, True):
        step = gs.GroundingStep(devc, sa_dtype=None, use_graph=True, pipeline=True, lr=0.0)
        _eval_dropout_train_bn(step)
        step.run(batches[0], batches[1])
        torch.cuda.synchronize()
        first = step.bucket.flat.clone()
        assert step._gD is not None, "the split backward is what this test is about"
        if delayed:
            real = step._gD.replay

            def late_replay():
                torch.cuda._sleep(int(30e-3 * 2.0e9))  # on the side stream (replay() is called under it)
                real()
            step._gD.replay = late_replay
        step.run(batches[1], batches[1])
        torch.cuda.synchronize()
        if delayed:
            del step._gD.replay  # (the instance attribute made a reference cycle through the graph object)
        flats.append(step.bucket.flat.clone())
        assert torch.isfinite(flats[-1]).all()
        assert _rel(flats[-1], first) > 1e-2, "the two batches must give different gradients for this test to see anything"
    assert _rel(flats[1], flats[0]) < 1e-4, _rel(flats[1], flats[0])


def test_geometry_pipeline_and_graph_equal_inline_step():
    """Precomputed backbone geometry (side stream) and hipGraph replay give the same STEP as the inline path: the flat
    gradient, the updated parameters and every persistent buffer (BatchNorm running statistics and counters) after
    step 1 — compared tightly, before any discrete decision of an updated model can differ."""
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    synth = importlib.import_module("3dvlp_amd.synth")
    devc = torch.device("cuda:0")
    batch = gs.batch_to_device(synth.make_batch(0, 2, num_points=8192, lang_num_max=2), devc)
    batch["random"] = torch.tensor(0.25, device=devc)  # fix the copy-paste coin
    res = {}
    for name, kw in (("inline", {}), ("pipeline", {"pipeline": True}), ("graph", {"pipeline": True, "use_graph": True}),
                     ("graph1", {"use_graph": True})):
        step = gs.GroundingStep(devc, **kw)
        _eval_dropout_train_bn(step)
        loss = float(step.run(batch))
        torch.cuda.synchronize()
        res[name] = dict(loss=loss, grad=step.bucket.flat.clone(),
                         params=torch.cat([p.detach().reshape(-1) for p in step.model.parameters()]),
                         bufs={n: b.clone() for n, b in step.model.named_buffers()})
        losses = [loss] + [float(step.run(batch)) for _ in range(2)]
        assert losses[2] < losses[0], (name, losses)
    ref = res["inline"]
    for name in ("pipeline", "graph", "graph1"):
        r = res[name]
        assert abs(r["loss"] - ref["loss"]) <= 1e-5 * abs(ref["loss"]), name
        # float atomics (scatter epilogues, loss partial sums) reorder additions: round-off level, not bitwise
        assert _rel(r["grad"], ref["grad"]) < 1e-4, (name, _rel(r["grad"], ref["grad"]))
        # first AdamW step = lr * g / (|g| + eps): elements whose gradient is at round-off level move by up to 2 lr in
        # either direction, the rest agree to the gradient's accuracy
        assert _rel(r["params"], ref["params"]) < 1e-4, name
        for n, b in ref["bufs"].items():
            if b.dtype.is_floating_point:
                torch.testing.assert_close(r["bufs"][n], b, rtol=1e-4, atol=1e-6, msg=f"{name}:{n}")
            else:
                assert torch.equal(r["bufs"][n], b), (name, n, r["bufs"][n], b)  # counters advanced exactly once


@pytest.mark.parametrize("use_graph,bf16", [(False, False), (True, False), (True, True)])
def test_pipeline_never_uses_stale_geometry(use_graph, bf16):
    """Alternating two DIFFERENT batches, with and without announcing the next one: every step must use the geometry
    of the batch it runs (`run(A)` then `run(B)` used A's FPS / ball-query indices for B).  The bf16 case also
    replays the captured graphs with a DIFFERENT number of distinct rows per batch (compact row map, csrc/sa_compact.hip:
    the row count is read on the device)."""
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    synth = importlib.import_module("3dvlp_amd.synth")
    devc = torch.device("cuda:0")
    A = gs.batch_to_device(synth.make_batch(0, 2, num_points=8192, lang_num_max=2), devc)
    Bb = gs.batch_to_device(synth.make_batch(2, 2, num_points=8192, lang_num_max=2), devc)
    for b in (A, Bb):
        b["random"] = torch.tensor(0.25, device=devc)
    seq = [(A, None), (Bb, None), (A, Bb), (Bb, A), (A, A), (Bb, None)]
    out = {}
    for name, kw in (("inline", {}), ("pipe", {"pipeline": True, "use_graph": use_graph})):
        step = gs.GroundingStep(devc, lr=0.0, sa_dtype=torch.bfloat16 if bf16 else None, **kw)   # lr 0: the model stays put
        _eval_dropout_train_bn(step)
        out[name] = []
        for cur, nxt in seq:
            loss = float(step.run(cur, nxt))
            torch.cuda.synchronize()
            out[name].append((loss, step.bucket.flat.clone()))
    for i, ((l0, g0), (l1, g1)) in enumerate(zip(out["inline"], out["pipe"])):
        # bf16: storage rounding turns the atomics' ordering noise into last-bit flips of stored activations
        assert abs(l0 - l1) <= (2e-3 if bf16 else 1e-5) * abs(l0), (i, l0, l1)
        assert _rel(g1, g0) < (3e-2 if bf16 else 1e-4), (i, _rel(g1, g0))
    # and the two batches really differ (so stale geometry would have been visible)
    assert abs(out["inline"][0][0] - out["inline"][1][0]) > 1e-3 * abs(out["inline"][0][0])


@pytest.mark.parametrize("use_graph", [False, True])
def test_pipeline_refills_when_a_freed_batch_address_is_reused(use_graph):
    """batches uploaded as temporaries — each one freed before the next upload, so the caching allocator hands
    out the SAME device address with in-place version 0 again.  An (address, version, shape) identity then took the new
    batch for the previous one: the captured graph kept its stale static inputs / the eager pipeline applied the previous
    batch's FPS and ball-query indices.  Identity is the live tensor object now; every step must equal the inline step."""
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    synth = importlib.import_module("3dvlp_amd.synth")
    devc = torch.device("cuda:0")
    hosts = [synth.make_batch(2 * i, 2, num_points=8192, lang_num_max=2) for i in range(3)]
    seq = [0, 1, 2, 1, 0]
    out = {}
    for name, kw in (("inline", {}), ("pipe", {"pipeline": True, "use_graph": use_graph})):
        step = gs.GroundingStep(devc, lr=0.0, **kw)
        _eval_dropout_train_bn(step)
        out[name], ptrs = [], []
        for i in seq:
            batch = gs.batch_to_device(hosts[i], devc)
            batch["random"] = torch.tensor(0.25, device=devc)
            ptrs.append(batch["point_clouds"].data_ptr())
            loss = float(step.run(batch))
            torch.cuda.synchronize()
            out[name].append((loss, step.bucket.flat.clone()))
            del batch
        out[name + "/ptrs"] = ptrs
    for i, ((l0, g0), (l1, g1)) in enumerate(zip(out["inline"], out["pipe"])):
        assert abs(l0 - l1) <= 1e-5 * abs(l0), (i, l0, l1)
        assert _rel(g1, g0) < 1e-4, (i, _rel(g1, g0))
    assert abs(out["inline"][0][0] - out["inline"][1][0]) > 1e-3 * abs(out["inline"][0][0])  # the batches do differ


@pytest.mark.parametrize("R,K,N", [(16384, 128, 128), (3136, 128, 128), (2048, 128, 256), (2048, 256, 128), (64, 64, 64), (16384, 128, 384), (2048, 128, 512), (8192, 256, 256)])
def test_mfma_linear_equals_f_linear(R, K, N):
    ml = importlib.import_module("3dvlp_amd.mfma_linear")
    torch.manual_seed(R + K)
    x = torch.randn(R // 32, 32, K, device="cuda", requires_grad=True)
    w = (torch.randn(N, K, device="cuda") * 0.1).requires_grad_(True)
    b = torch.randn(N, device="cuda", requires_grad=True)
    assert ml.supported(x, w)
    y = ml.linear(x, w, b)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    torch.testing.assert_close(y.double(), ref, rtol=1e-5, atol=1e-5)
    g = torch.randn_like(y)
    got = torch.autograd.grad(y, [x, w, b], g)
    exp = torch.autograd.grad(ref, [x, w, b], g.double())
    for a, e in zip(got, exp):
        assert (a.double() - e.double()).abs().max().item() < 1e-4 * e.abs().max().item() + 1e-5


@pytest.mark.parametrize("R,K,N", [(16384, 128, 128), (2048, 128, 256), (2048, 256, 128), (16384, 128, 384), (64, 64, 64)])
def test_mfma_linear_bf16_mma_close_to_fp64(R, K, N):
    """The timing-configuration form (bf16 MFMA operands, fp32 I/O and accumulation): within bf16 operand rounding of the
    fp64 result — norm-wise 2^-8 — and EQUAL to the exact kernels evaluated on bf16-rounded operands (same products)."""
    ml = importlib.import_module("3dvlp_amd.mfma_linear")
    torch.manual_seed(R + K)
    x = torch.randn(R // 32, 32, K, device="cuda", requires_grad=True)
    w = (torch.randn(N, K, device="cuda") * 0.1).requires_grad_(True)
    b = torch.randn(N, device="cuda", requires_grad=True)
    with ml.bf16_mma(True):
        y = ml.linear(x, w, b)
    assert not ml.BF16_MMA
    g = torch.randn_like(y)
    got = torch.autograd.grad(y, [x, w, b], g)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    exp = torch.autograd.grad(ref, [x, w, b], g.double())
    assert _rel(y, ref) < 4e-3
    for a, e in zip(got, exp):
        assert _rel(a, e) < 4e-3, _rel(a, e)
    # same products on pre-rounded operands through the exact-fp32 kernels
    r = lambda t: t.detach().bfloat16().float()
    y2 = torch.nn.functional.linear(r(x).double(), r(w).double(), b.double())
    assert _rel(y, y2) < 1e-5
    dx2 = r(g).double().reshape(-1, N) @ r(w).double()
    assert _rel(got[0].reshape(-1, K), dx2) < 1e-5
    dw2 = r(g).double().reshape(-1, N).t() @ r(x).double().reshape(-1, K)
    assert _rel(got[1], dw2) < 1e-5
    assert _rel(got[2], r(g).double().reshape(-1, N).sum(0)) < 1e-5


@pytest.mark.parametrize("NH", [1, 12])
def test_box_decode_fused_equals_op_sequence(NH):
    """csrc/box_decode.hip vs the op-by-op restatement of decode_pred_box + get_3d_box_batch: same values (fp32
    round-off of cos/sin only) and the same gradients w.r.t. rois, heading residuals and vote centres."""
    det = importlib.import_module("3dvlp_amd.detection")
    g = torch.Generator(device="cpu").manual_seed(3 + NH)
    B, K = 3, 256
    base = {"aggregated_vote_xyz": torch.randn(B, K, 3, generator=g) * 2,
            "heading_scores": torch.randn(B, K, NH, generator=g),
            "heading_residuals": torch.randn(B, K, NH, generator=g) * 0.2,
            "rois": torch.randn(B, K, 6, generator=g).mul(0.5).exp()}
    base["heading_scores"][0, 0, :] = 0.25  # a tie: the first maximum wins
    weights = [torch.randn(B, K, generator=g), torch.randn(B, K, 3, generator=g), torch.randn(B, K, 3, generator=g)]
    res = []
    for fused in (False, True):
        pm = det.ProposalModule(18, NH, 18, np.ones((18, 3), np.float32), K, "vote_fps").cuda()
        pm.fused_decode = fused
        d = {k: v.clone().cuda().requires_grad_(k != "heading_scores") for k, v in base.items()}
        out = pm.decode_pred_box(dict(d))
        loss = sum((out[k] * w.cuda()).sum() for k, w in zip(("pred_heading", "pred_size", "pred_center"), weights))
        loss.backward()
        assert not out["pred_bbox_corner"].requires_grad
        res.append(([out[k].detach() for k in ("pred_heading", "pred_size", "pred_center", "pred_bbox_corner")],
                    [d[k].grad for k in ("aggregated_vote_xyz", "heading_residuals", "rois")]))
    for a, b in zip(res[0][0] + res[0][1], res[1][0] + res[1][1]):
        torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("R,D,p", [(2048, 128, 0.1), (16384, 128, 0.1), (70, 64, 0.5), (33, 256, 0.0)])
def test_add_norm_fused_equals_torch_with_same_mask(R, D, p):
    """csrc/add_norm.hip vs LayerNorm(x + y*mask/(1-p)) evaluated by torch with the kernel's own keep mask: outputs and
    all four gradients; the backward kernel must regenerate exactly that mask; advancing the seed redraws it."""
    an = importlib.import_module("3dvlp_amd.add_norm")
    torch.manual_seed(R + D)
    norm = torch.nn.LayerNorm(D).cuda()
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.5, 0.5)
    x = torch.randn(R // 1 if R < 100 else R // 32, 1 if R < 100 else 32, D, device="cuda", requires_grad=True)
    y = torch.randn_like(x, requires_grad=True)
    mask = torch.empty(x.shape, dtype=torch.uint8, device="cuda")
    out = an.add_norm(x, y, norm, p, True, mask_out=mask)
    go = torch.randn_like(out)
    got = torch.autograd.grad(out, [x, y, norm.weight, norm.bias], go)
    keep = mask.double() if p > 0 else torch.ones_like(x, dtype=torch.double)
    xd, yd = x.detach().double().requires_grad_(True), y.detach().double().requires_grad_(True)
    wd, bd = norm.weight.detach().double().requires_grad_(True), norm.bias.detach().double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd + yd * keep / (1.0 - p), (D,), wd, bd, norm.eps)
    exp = torch.autograd.grad(ref, [xd, yd, wd, bd], go.double())
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=1e-5)
    for a, e in zip(got, exp):
        assert (a.double() - e).abs().max().item() < 1e-4 * e.abs().max().item() + 1e-5
    if p > 0:
        frac = mask.float().mean().item()
        assert abs(frac - (1 - p)) < 4 * (p * (1 - p) / mask.numel()) ** 0.5 + 1e-3  # Bernoulli(1-p) keep rate
        m2 = torch.empty_like(mask)
        an.add_norm(x, y, norm, p, True, mask_out=m2)      # another call id: another mask
        assert not torch.equal(mask, m2)
    # eval mode: no dropout at all
    ev = an.add_norm(x, y, norm, p, False)
    torch.testing.assert_close(ev, torch.nn.functional.layer_norm(x + y, (D,), norm.weight, norm.bias, norm.eps),
                               rtol=1e-5, atol=1e-5)


def _grads_close(a, b, tol=5e-3):
    """Module-level comparison against an fp64 formula, norm-wise.  The tolerance is NOT the kernels' accuracy (that is
    test_row_stack_exact_vs_fp64: 1e-5): an activation within fp32 round-off of zero takes the other side of a ReLU than
    in fp64, which changes one row's whole contribution to a weight gradient (measured: 1.7e-3 of the norm for conv1 of the
    eval-mode voting module, identically for the rows kernels and for the library Conv1d/BatchNorm1d sequence in fp32)."""
    a, b = a.double(), b.double()
    assert (a - b).norm().item() <= tol * b.norm().item() + 1e-9, ((a - b).norm().item(), b.norm().item())


@pytest.mark.parametrize("bf16", [False, True])
def test_row_stack_with_prepared_k_major_weights_equals_row_major(bf16):
    """Inside `with row_mlp.PreparedWeights()` the forward product of a rows stack reads K-major weight copies made by ONE
    batched transpose at the start of the pass (vlp3d_transpose_batch, vlp3d_rows_fwd_wt): the first pass registers the
    weights (and runs the row-major kernel), the second is served the copies — same outputs and gradients; a weight changed
    between the passes is transposed again; outside the context nothing is served."""
    rm = importlib.import_module("3dvlp_amd.row_mlp")
    ml = importlib.import_module("3dvlp_amd.mfma_linear")
    torch.manual_seed(3)
    dims = [512, 256, 128, 64]
    Ws = [(torch.randn(dims[i + 1], dims[i], device="cuda") * 0.1).requires_grad_(True) for i in range(3)]
    bns = [torch.nn.BatchNorm1d(d).cuda().train() for d in dims[1:]]
    x = torch.randn(4096, 512, device="cuda")
    go = torch.randn(4096, 64, device="cuda")
    prep = rm.PreparedWeights()

    def run(ctx):
        for w in Ws:
            w.grad = None
        with ml.bf16_mma(bf16):
            if ctx:
                with prep:
                    y = rm.row_stack(x, [(w, None, bn) for w, bn in zip(Ws, bns)])
            else:
                y = rm.row_stack(x, [(w, None, bn) for w, bn in zip(Ws, bns)])
        (y * go).sum().backward()
        return [y.detach().clone()] + [w.grad.clone() for w in Ws]
    plain = run(False)
    first = run(True)                      # registers: row-major kernels
    assert len(prep.entries) == 3 and len(prep.fresh) == 3
    served = run(True)                     # K-major copies
    assert len(prep.fresh) == 0
    tol = 2e-3 if bf16 else 1e-6           # bf16: the two kernels round the same operands but sum in different orders
    for a, b, c in zip(plain, first, served):
        assert torch.equal(a, b)
        assert _rel(c, a) <= tol, _rel(c, a)
    with torch.no_grad():
        Ws[1].add_(0.3 * torch.randn_like(Ws[1]))  # (a pure scaling would vanish in the train-mode BatchNorm behind it)
    changed_plain = run(False)
    changed = run(True)
    assert _rel(changed[0], changed_plain[0]) <= tol and _rel(changed[0], plain[0]) > 1e-2
    assert rm._ACTIVE is None


def test_row_stack_declines_widths_its_kernels_do_not_stage():
    """A BatchNorm layer 192 wide passed `supported` (n % 64 == 0) although the BatchNorm loaders of the rows products stage
    four columns per thread and need n / 4 to divide 256 — the stack then failed loudly inside backward (round 3)."""
    rm = importlib.import_module("3dvlp_amd.row_mlp")
    x = torch.randn(2048, 128, device="cuda")
    for n, ok in ((64, True), (128, True), (192, False), (256, True), (320, False), (512, True)):
        w, b, bn = torch.randn(n, 128, device="cuda"), torch.randn(n, device="cuda"), torch.nn.BatchNorm1d(n).cuda()
        assert rm.supported(x, [(w, b, bn)]) == ok, n


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("R,dims,last_plain", [(4096, [512, 256, 256], False), (8192, [256, 256, 256, 259], True),
                                               (2048, [128, 128, 128, 28], True), (4096, [256, 128, 256], False),
                                               (64, [64, 64], False),
                                               # 96 rows x 128: three partial workgroups of rows_act_bwd's row-lane form
                                               (96, [64, 128], False)])
def test_row_stack_exact_vs_fp64(R, dims, last_plain, training):
    """row_mlp.row_stack (csrc/rows_mlp.hip + the weight-gradient kernel of csrc/sa_mlp.hip) against Linear(+bias) ->
    BatchNorm1d -> ReLU in fp64 on random data: output, input gradient and every parameter gradient to 1e-5 of their
    scale (up to an isolated ReLU flip)."""
    rm = importlib.import_module("3dvlp_amd.row_mlp")
    torch.manual_seed(R + len(dims))
    L = len(dims) - 1
    Ws = [torch.randn(dims[i + 1], dims[i], device="cuda") * 0.1 for i in range(L)]
    bs = [torch.randn(dims[i + 1], device="cuda") * 0.1 for i in range(L)]
    bns = [None if (last_plain and i == L - 1) else torch.nn.BatchNorm1d(dims[i + 1]).cuda().train(training) for i in range(L)]
    for bn in bns:
        if bn is not None:
            _randomise_bn(bn)
    x0, go = torch.randn(R, dims[0], device="cuda"), torch.randn(R, dims[-1], device="cuda")
    x = x0.clone().requires_grad_(True)
    W = [w.clone().requires_grad_(True) for w in Ws]
    b = [t.clone().requires_grad_(True) for t in bs]
    assert rm.supported(x, [(W[i], b[i], bns[i]) for i in range(L)])
    y = rm.row_stack(x, [(W[i], b[i], bns[i]) for i in range(L)])
    (y * go).sum().backward()
    got = [y.detach(), x.grad] + [w.grad for w in W] + [t.grad for t in b] + \
          [bn.weight.grad for bn in bns if bn is not None] + [bn.bias.grad for bn in bns if bn is not None]
    xd = x0.double().requires_grad_(True)
    Wd = [w.double().requires_grad_(True) for w in Ws]
    bd = [t.double().requires_grad_(True) for t in bs]
    gd = [bn.weight.detach().double().requires_grad_(True) for bn in bns if bn is not None]
    ed = [bn.bias.detach().double().requires_grad_(True) for bn in bns if bn is not None]
    h, j = xd, 0
    for i in range(L):
        h = h @ Wd[i].t() + bd[i]
        if bns[i] is not None:
            mean, var = (h.mean(0), h.var(0, unbiased=False)) if training else (bns[i].running_mean.double(), bns[i].running_var.double())
            h = torch.relu((h - mean) / torch.sqrt(var + bns[i].eps) * gd[j] + ed[j])
            j += 1
    (h * go.double()).sum().backward()
    exp = [h.detach(), xd.grad] + [w.grad for w in Wd] + [t.grad for t in bd] + [t.grad for t in gd] + [t.grad for t in ed]
    for k, (a_, e_) in enumerate(zip(got, exp)):
        pre_bn_bias = training and 2 + L <= k < 2 + 2 * L and bns[k - 2 - L] is not None
        if pre_bn_bias:
            assert a_.abs().max().item() == 0 and e_.abs().max().item() < 1e-8   # cancels inside a train-mode BatchNorm
            continue
        # element-wise 1e-5 of the tensor's scale, except where ONE activation sat within fp32 round-off of a ReLU
        # threshold and took the other side than in fp64: that changes one row of dx / one row of a dW, nothing else
        scale = e_.abs().max().item()
        bad = ((a_.double() - e_).abs() > 1e-5 * scale + 1e-7).double().mean().item()
        assert bad < 1e-2, (k, bad, (a_.double() - e_).abs().max().item(), scale)
        assert ((a_.double() - e_).norm() <= 2e-3 * e_.norm() + 1e-9).item(), k


def _bn_rows64(x, bn, training):
    """BatchNorm over the rows of a (R, C) fp64 matrix with the module's parameters (batch statistics when training)."""
    if training:
        mean, var = x.mean(0), x.var(0, unbiased=False)
    else:
        mean, var = bn.running_mean.double(), bn.running_var.double()
    return (x - mean) / torch.sqrt(var + bn.eps) * bn.weight.double() + bn.bias.double()


def _randomise_bn(mod):
    with torch.no_grad():
        for m_ in mod.modules():
            if isinstance(m_, torch.nn.modules.batchnorm._BatchNorm):
                m_.weight.uniform_(0.5, 1.5); m_.bias.uniform_(-0.3, 0.3)
                m_.running_mean.uniform_(-0.2, 0.2); m_.running_var.uniform_(0.5, 1.5)


@pytest.mark.parametrize("training", [True, False])
def test_fp_module_rows_equals_fp64_formula(training):
    """PointnetFPModule on point-major rows (csrc/rows_mlp.hip: interpolate + concat, MFMA products with BatchNorm / ReLU
    folded in) == pointnet2_modules.py:393-416 evaluated in fp64 with torch ops (inverse-distance blend of the three
    neighbours, concat, 1x1 conv, BatchNorm over B*n, ReLU): output, input gradients, parameter gradients, running
    statistics — FP2 shape of cfg2.  (The library BatchNorm backward of the literal NCHW sequence is NOT used as the
    reference: with non-default running statistics it returned gradients 10 % off the fp64 formula on this stack.)"""
    pm = importlib.import_module("3dvlp_amd.pointnet2_modules")
    torch.manual_seed(4)
    B, n, m = 4, 1024, 512
    unknown = torch.rand(B, n, 3, device="cuda") * 4
    known = unknown[:, torch.randperm(n)[:m]].contiguous() + 0.01
    fp = pm.PointnetFPModule(mlp=[512, 256, 256]).cuda().train(training)
    _randomise_bn(fp)
    rm0 = [b.clone() for b in fp.buffers()]
    uf0, kf0 = torch.randn(B, n, 256, device="cuda"), torch.randn(B, m, 256, device="cuda")
    g = torch.randn(B, 256, n, device="cuda")
    uf, kf = uf0.clone().requires_grad_(True), kf0.clone().requires_grad_(True)
    out = fp(unknown, known, uf.transpose(1, 2), kf.transpose(1, 2))   # (B,C,n) views of point-major data
    assert not out.is_contiguous()                                      # the rows path ran: (B,C,n) is a view of (B,n,C)
    (out * g).sum().backward()
    got = [out.detach(), uf.grad, kf.grad] + [p.grad.clone() for p in fp.parameters()]
    # fp64 reference
    geo = pm.PointnetFPModule.compute_geometry(unknown, known)
    idx, w = geo[:2]
    assert len(geo) == 4, "the inverse three_nn map travels with the geometry (atomic-free adjoint)"
    # the LDS-atomic adjoint (no inverse map) agrees with the CSR form the module just used
    rm = importlib.import_module("3dvlp_amd.row_mlp")
    kf2 = kf0.clone().requires_grad_(True)
    X2 = rm.fp_rows(kf2, uf0, idx, w, None)
    gX = torch.randn_like(X2)
    X2.backward(gX)
    kf3 = kf0.clone().requires_grad_(True)
    rm.fp_rows(kf3, uf0, idx, w, tuple(geo[2:4])).backward(gX)
    _grads_close(kf3.grad, kf2.grad.double(), 1e-5)
    ufd, kfd = uf0.double().requires_grad_(True), kf0.double().requires_grad_(True)
    base = (torch.arange(B, device="cuda") * m)[:, None, None]
    nb = kfd.reshape(B * m, 256)[(idx.long() + base).reshape(-1, 3)]
    x = torch.cat([(nb * w.double().reshape(-1, 3, 1)).sum(1), ufd.reshape(B * n, 256)], 1)
    params64 = []
    for layer in fp.mlp:
        W = layer.conv.weight.detach().double()[:, :, 0, 0].requires_grad_(True)
        bn = layer.bn.bn
        gam, bet = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
        y = x @ W.t()
        mean, var = (y.mean(0), y.var(0, unbiased=False)) if training else (bn.running_mean.double(), bn.running_var.double())
        x = torch.relu((y - mean) / torch.sqrt(var + bn.eps) * gam + bet)
        params64 += [(W, y.detach()), gam, bet]
    ref = x.view(B, n, 256).transpose(1, 2)
    (ref * g.double()).sum().backward()
    torch.testing.assert_close(got[0].double(), ref.detach(), rtol=1e-4, atol=2e-5)
    _grads_close(got[1], ufd.grad, 2e-4); _grads_close(got[2], kfd.grad, 2e-4)
    exp = []
    for W, gam, bet in zip(params64[0::3], params64[1::3], params64[2::3]):
        exp += [W[0].grad[:, :, None, None], gam.grad, bet.grad]
    for a, b in zip(got[3:], exp):
        _grads_close(a, b)
    if training:  # running statistics: momentum update with the batch mean / unbiased variance
        R = B * n
        for layer, (W, y) in zip(fp.mlp, params64[0::3]):
            bn = layer.bn.bn
        bufs = dict(fp.named_buffers())
        i = 0
        for li, layer in enumerate(fp.mlp):
            y = params64[3 * li][1]
            mom = layer.bn.bn.momentum
            torch.testing.assert_close(bufs[f"mlp.layer{li}.bn.bn.running_mean"].double(),
                                       (1 - mom) * rm0[3 * li].double() + mom * y.mean(0), rtol=1e-4, atol=1e-6)
            torch.testing.assert_close(bufs[f"mlp.layer{li}.bn.bn.running_var"].double(),
                                       (1 - mom) * rm0[3 * li + 1].double() + mom * y.var(0, unbiased=True), rtol=1e-4, atol=1e-6)
            assert int(bufs[f"mlp.layer{li}.bn.bn.num_batches_tracked"]) == int(rm0[3 * li + 2]) + 1


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("kind", ["voting", "roi"])
def test_voting_and_roi_heads_rows_equal_fp64_formula(kind, training):
    """VotingModule (voting_module.py:33-60) / StandardROIHeads (roi_heads.py:127-147) on csrc/rows_mlp.hip vs the same
    Conv1d(+bias) -> BatchNorm1d -> ReLU chain evaluated in fp64: outputs, input gradient, every parameter gradient (a
    bias in front of a train-mode BatchNorm: exactly zero), running statistics including the bias shift."""
    det = importlib.import_module("3dvlp_amd.detection")
    torch.manual_seed(6)
    B, S = (4, 1024) if kind == "voting" else (8, 256)
    C = 256 if kind == "voting" else 128
    mod = (det.VotingModule(1, 256) if kind == "voting" else det.StandardROIHeads(1, 18)).cuda().train(training)
    _randomise_bn(mod)
    with torch.no_grad():
        for n_, p in mod.named_parameters():
            if n_.endswith("bias") and p.dim() == 1:
                p.uniform_(-0.3, 0.3)
    buf0 = {n_: b.clone() for n_, b in mod.named_buffers()}
    f = torch.randn(B, S, C, device="cuda").requires_grad_(True)
    xyz = torch.rand(B, S, 3, device="cuda")
    if kind == "voting":
        convs, bns = [mod.conv1, mod.conv2, mod.conv3], [mod.bn1, mod.bn2, None]
        vx, vf = mod(xyz, f.transpose(1, 2))
        outs = [vx, vf.transpose(1, 2)]
    else:
        heads = [mod.heading_reg_predictor, mod.heading_cls_predictor, mod.box_predictor, mod.objectness_predictor,
                 mod.sem_cls_predictor]
        convs, bns = [mod.convs[0], mod.convs[3], None], [mod.convs[1], mod.convs[4], None]
        d = mod(f.transpose(1, 2), {})
        outs = [d["heading_residuals_normalized"], d["heading_scores"], d["rois"], d["objectness_scores"], d["sem_cls_scores"]]
    gs_ = [torch.randn_like(o) for o in outs]
    sum((o * g).sum() for o, g in zip(outs, gs_)).backward()
    got = {n_: p.grad.clone() for n_, p in mod.named_parameters() if p.grad is not None}
    # fp64 formula
    fd = f.detach().double().requires_grad_(True)
    P = {n_: p.detach().double().requires_grad_(True) for n_, p in mod.named_parameters()}
    names = dict((id(m_), n_) for n_, m_ in mod.named_modules())
    x = fd.reshape(B * S, C)
    pre = []
    for conv, bn in zip(convs, bns):
        if conv is None:   # merged ROI predictors
            W = torch.cat([P[names[id(h)] + ".weight"][:, :, 0] for h in heads], 0)
            bb = torch.cat([P[names[id(h)] + ".bias"] for h in heads], 0)
        else:
            W, bb = P[names[id(conv)] + ".weight"][:, :, 0], P[names[id(conv)] + ".bias"]
        y = x @ W.t() + bb
        pre.append(y.detach())
        if bn is None:
            x = y
            break
        mean, var = (y.mean(0), y.var(0, unbiased=False)) if training else (bn.running_mean.double(), bn.running_var.double())
        x = torch.relu((y - mean) / torch.sqrt(var + bn.eps) * P[names[id(bn)] + ".weight"] + P[names[id(bn)] + ".bias"])
    if kind == "voting":
        net = x.view(B, S, 1, 3 + C)
        ref = [(xyz.double().unsqueeze(2) + net[..., 0:3]).reshape(B, S, 3), (fd.unsqueeze(2) + net[..., 3:]).reshape(B, S, C)]
    else:
        parts = torch.split(x.view(B, S, -1), [h.weight.shape[0] for h in heads], dim=-1)
        ref = [parts[0], parts[1], parts[2].exp(), parts[3], parts[4]]
    sum((o * g.double()).sum() for o, g in zip(ref, gs_)).backward()
    for o, r in zip(outs, ref):
        torch.testing.assert_close(o.double(), r.detach(), rtol=1e-4, atol=2e-5)
    _grads_close(f.grad, fd.grad)
    for n_, p64 in P.items():
        if p64.grad is None:
            assert n_ not in got, n_
            continue
        pre_bn_bias = training and n_ in ("conv1.bias", "conv2.bias", "convs.0.bias", "convs.3.bias")
        if pre_bn_bias:  # cancels inside the BatchNorm: exactly zero here, round-off in any op-by-op evaluation
            assert got[n_].abs().max().item() == 0 and p64.grad.abs().max().item() < 1e-8
        else:
            _grads_close(got[n_], p64.grad)
    if training:
        now = dict(mod.named_buffers())
        for (conv, bn), y in zip(zip(convs, bns), pre):
            if bn is None:
                continue
            nm = names[id(bn)]
            mom = bn.momentum
            torch.testing.assert_close(now[nm + ".running_mean"].double(), (1 - mom) * buf0[nm + ".running_mean"].double() + mom * y.mean(0),
                                       rtol=1e-4, atol=1e-6)
            torch.testing.assert_close(now[nm + ".running_var"].double(),
                                       (1 - mom) * buf0[nm + ".running_var"].double() + mom * y.var(0, unbiased=True), rtol=1e-4, atol=1e-6)


def test_glue_kernels_equal_torch_ops():
    """csrc/glue.hip vs the op sequences they replace: ROI split (+exp, scale, arg-max), vote epilogue (+L2 norm), row
    L2 normalisation, copy-paste augmentation — values and gradients."""
    glue = importlib.import_module("3dvlp_amd.glue")
    grd = importlib.import_module("3dvlp_amd.grounding")
    torch.manual_seed(12)
    dev = "cuda"
    # --- roi_split
    B, K, NH, NC, ld = 4, 256, 1, 18, 64
    out0 = torch.randn(B, K, ld, device=dev)
    ws = [torch.randn(B, K, n, device=dev) for n in (NH, NH, NH, 6, 2, NC)]
    a = out0.clone().requires_grad_(True)
    got = glue.roi_split(a, NH, NC)
    sum((g * w).sum() for g, w in zip(got[:6], ws)).backward()
    b = out0.clone().requires_grad_(True)
    parts = torch.split(b[..., :2 * NH + 8 + NC], [NH, NH, 6, 2, NC], dim=-1)
    exp = [parts[0], parts[0] * (np.pi / NH), parts[1], parts[2].exp(), parts[3], parts[4]]
    sum((g * w).sum() for g, w in zip(exp, ws)).backward()
    for x, y in zip(got[:6], exp):
        torch.testing.assert_close(x, y, rtol=1e-6, atol=1e-6)
    assert torch.equal(got[6], exp[4].argmax(-1)) and torch.equal(got[7], exp[5].argmax(-1))
    torch.testing.assert_close(a.grad, b.grad, rtol=1e-6, atol=1e-6)
    # --- vote epilogue
    S, C, ldn = 1024, 256, 320
    sx, sf = torch.rand(B, S, 3, device=dev), torch.randn(B, S, C, device=dev)
    net0 = torch.randn(B * S, ldn, device=dev)
    w1, w2 = torch.randn(B, S, 3, device=dev), torch.randn(B, S, C, device=dev)
    res = []
    for fused in (True, False):
        f = sf.clone().requires_grad_(True)
        n = net0.clone().requires_grad_(True)
        if fused:
            vx, vf = glue.vote_epilogue(sx, f, n)
        else:
            net = n[:, :3 + C].view(B, S, 3 + C)
            vx = sx + net[..., :3]
            v = f + net[..., 3:]
            vf = v / torch.norm(v, p=2, dim=-1, keepdim=True)
        ((vx * w1).sum() + (vf * w2).sum()).backward()
        res.append((vx.detach(), vf.detach(), f.grad, n.grad))
    for x, y in zip(*res):
        torch.testing.assert_close(x, y, rtol=1e-5, atol=1e-6)
    # --- l2norm rows
    x0 = torch.randn(2112, 128, device=dev)
    x0[5] = 0
    g = torch.randn_like(x0)
    x1, x2 = x0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
    y1 = glue.l2norm_rows(x1)
    y2 = torch.nn.functional.normalize(x2, dim=-1)
    (y1 * g).sum().backward(); (y2 * g).sum().backward()
    torch.testing.assert_close(y1, y2, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(x1.grad[6:], x2.grad[6:], rtol=1e-5, atol=1e-6)
    assert torch.isfinite(x1.grad).all()
    # --- copy-paste (random objectness patterns, both coin sides, degenerate: no objects / all objects)
    for trial in range(12):
        Bc, Kc, D = 8, 256, 128
        feats = torch.randn(Bc, Kc, D, device=dev)
        p = [0.0, 1.0, 0.1, 0.5][trial % 4] if trial < 8 else torch.rand(()).item()
        mask = (torch.rand(Bc, Kc, device=dev) < p).long()
        for coin in (0.2, 0.8):
            c = torch.tensor(coin, device=dev)
            fa = feats.clone().requires_grad_(True)
            oa = glue.copy_paste(fa, mask, c)
            fb = feats.clone().requires_grad_(True)
            ob = torch.where(c < 0.5, grd.MatchModule._copy_paste(fb, mask.float().unsqueeze(2)), fb)
            assert torch.equal(oa, ob), (trial, coin)
            w = torch.randn_like(oa)
            (oa * w).sum().backward(); (ob * w).sum().backward()
            torch.testing.assert_close(fa.grad, fb.grad, rtol=1e-5, atol=1e-5)


def test_flat_params_merge_views_and_flat_adamw_equal_torch():
    """ddp.FlatParams / merge_adjacent / FlatAdamW: the merged q|k|v weight is a VIEW of the flat parameter buffer (no
    copy), its gradient splits back into the three parameters, and three flat AdamW steps equal torch.optim.AdamW
    (including 'no gradient -> untouched, no weight decay')."""
    import copy
    ddp = importlib.import_module("3dvlp_amd.ddp")
    tr = importlib.import_module("3dvlp_amd.transformer")
    torch.manual_seed(2)
    net = torch.nn.ModuleDict({"att": tr.MultiHeadAttention(128, 32, 32, 4, dropout=0.0), "unused": torch.nn.Linear(8, 8)}).cuda()
    ref = copy.deepcopy(net)
    layout = ddp.FlatParams(net)
    bucket = ddp.FlatGradBucket(net, layout=layout)
    opt = ddp.FlatAdamW(layout, bucket, lr=1e-2, weight_decay=0.1)
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=0.1)
    a = net["att"].attention
    w = ddp.merge_adjacent([a.fc_q.weight, a.fc_k.weight, a.fc_v.weight])
    assert w.data_ptr() == a.fc_q.weight.data_ptr() and w.shape == (384, 128)       # a view, not a copy
    assert torch.equal(w.detach(), torch.cat([a.fc_q.weight, a.fc_k.weight, a.fc_v.weight]).detach())
    for n_, p in net.named_parameters():                                              # values survived the re-homing
        assert torch.equal(p.detach(), dict(ref.named_parameters())[n_].detach()), n_
    x = torch.randn(4, 256, 128, device="cuda")
    for step in range(3):
        bucket.zero()
        net["att"](x, x, x).pow(2).mean().backward()
        bucket.collect()
        opt.step()
        ropt.zero_grad(set_to_none=True)
        ref["att"](x, x, x).pow(2).mean().backward()
        ropt.step()
        for (n_, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
            # step 1 is exact to round-off; later steps amplify 1e-5 gradient differences of near-zero elements (the
            # update is ~ lr * sign(g) there): a few 1e-5 absolute at lr = 1e-2
            torch.testing.assert_close(p, q, rtol=0, atol=1e-6 if step == 0 else 2e-4, msg=f"{n_} step {step}")
    assert net["unused"].weight.grad is None and torch.equal(net["unused"].weight, ref["unused"].weight)


def test_flat_adamw_late_parameters_get_their_own_step_count_and_state_round_trips():
    """torch.optim.AdamW keeps a step count PER PARAMETER, starting when the parameter first receives a gradient.
    A parameter that joins later (here: the second layer is switched on after three steps) must get bias corrections of ITS
    OWN first steps, not the global ones; and state_dict() / load_state_dict() carry m, v and the counts."""
    import copy
    ddp = importlib.import_module("3dvlp_amd.ddp")
    torch.manual_seed(4)
    net = torch.nn.ModuleDict({"a": torch.nn.Linear(16, 16), "b": torch.nn.Linear(16, 16), "c": torch.nn.Linear(16, 16)}).cuda()
    ref = copy.deepcopy(net)
    layout = ddp.FlatParams(net)
    bucket = ddp.FlatGradBucket(net, layout=layout)
    opt = ddp.FlatAdamW(layout, bucket, lr=1e-2, weight_decay=0.1)
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=0.1)
    x = torch.randn(64, 16, device="cuda")

    def loss_of(m, with_b):
        h = m["a"](x)
        if with_b:
            h = m["b"](torch.relu(h))       # "b" sits BETWEEN "a" and "c" in the flat buffer: the launch splits into runs
        return m["c"](h).pow(2).mean()

    saved = None
    for it in range(6):
        with_b = it >= 3
        bucket.zero()
        loss_of(net, with_b).backward()
        bucket.collect()
        opt.step()
        ropt.zero_grad(set_to_none=True)
        loss_of(ref, with_b).backward()
        ropt.step()
        for (n_, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
            torch.testing.assert_close(p, q, rtol=0, atol=2e-5, msg=f"{n_} step {it}")
        if it == 3:
            assert sorted(set(opt.steps)) == [1, 4] and len(opt._runs) == 3
            saved = (opt.state_dict(), layout.flat.clone())
    # resume from the checkpoint taken after step 3: the same two further steps give the same parameters
    end = layout.flat.clone()
    layout.flat.copy_(saved[1])
    opt2 = ddp.FlatAdamW(layout, bucket, lr=1.0)      # (hyper-parameters come back from the state dict)
    opt2.load_state_dict(saved[0])
    for it in (4, 5):
        bucket.zero()
        loss_of(net, True).backward()
        bucket.collect()
        opt2.step()
    torch.testing.assert_close(layout.flat, end, rtol=0, atol=1e-6)


def test_captured_step_recaptures_when_the_epoch_crosses_50():
    """the loss configuration (reference-loss weight, label smoothing, OCC / OSC from epoch 50) is baked into a
    captured graph.  Changing `step.epoch` across the boundary must recapture: the replayed loss then equals an eager
    step's at the new epoch, and the contrast losses appear."""
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    synth = importlib.import_module("3dvlp_amd.synth")
    devc = torch.device("cuda:0")
    batch = gs.batch_to_device(synth.make_batch(0, 2, num_points=8192, lang_num_max=2), devc)
    batch["random"] = torch.tensor(0.75, device=devc)
    res = {}
    for name, kw in (("graph", {"use_graph": True, "pipeline": True}), ("eager", {})):
        step = gs.GroundingStep(devc, epoch=49, lr=0.0, **kw)
        _eval_dropout_train_bn(step)
        l49 = float(step.run(batch))
        out = step._static_out if name == "graph" else step._last_out
        had = "iou_con_loss" in out
        step.epoch = 50
        l50 = float(step.run(batch))
        out = step._static_out if name == "graph" else step._last_out
        res[name] = (l49, l50, had, "iou_con_loss" in out)
    # OCC / OSC are evaluated (and their projections enter the graph) only from epoch 50 on: a stale graph would keep
    # replaying the epoch-49 step, without them
    assert res["graph"][2:] == (False, True) and res["eager"][2:] == (False, True), res
    assert abs(res["graph"][0] - res["eager"][0]) <= 1e-5 * abs(res["eager"][0])
    assert abs(res["graph"][1] - res["eager"][1]) <= 1e-5 * abs(res["eager"][1])


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_decoder_layer_tiled_queries_equal_explicit_tiling(p):
    """CrossAttentionDecoderLayer.forward_tiled(query (B,K,C), rep, ...) == forward(query tiled rep times, ...): the first
    decoder layer of the match module runs its self-attention block once per scene instead of once per (scene, sentence).
    p = 0: outputs and every gradient equal (reassociated sums only).  p = 0.1: the replicas draw INDEPENDENT masks (rows of
    different replicas differ) at the expected keep rate, and the backward is the derivative of that forward."""
    tr = importlib.import_module("3dvlp_amd.transformer")
    an = importlib.import_module("3dvlp_amd.add_norm")
    torch.manual_seed(3)
    B, L, K, C, T = 3, 4, 64, 128, 17
    layer = tr.CrossAttentionDecoderLayer(hidden_size=C).cuda().train()
    for m in layer.modules():
        if hasattr(m, "fused_norm"):
            m.fused_norm = True
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    layer.self_attention.dropout.p = p
    q = torch.randn(B, K, C, device="cuda")
    kv = torch.randn(B * L, T, C, device="cuda")
    g = torch.randn(B * L, K, C, device="cuda")
    res = []
    for tiled in (False, True):
        layer.zero_grad()
        x = q.clone().requires_grad_(True)
        an._CALLS[0] = 100   # same call ids -> same masks for the layers that follow in both evaluations
        out = layer.forward_tiled(x, L, kv, kv) if tiled else layer(x[:, None].expand(B, L, K, C).reshape(B * L, K, C), kv, kv)
        (out * g).sum().backward()
        res.append((out.detach(), x.grad.clone(), {n: p_.grad.clone() for n, p_ in layer.named_parameters()}))
    (oa, ga, pa), (ob, gb, pb) = res
    if p == 0.0:
        assert _rel(ob, oa) < 1e-5 and _rel(gb, ga) < 1e-4
        scale = max(float(v.norm()) for v in pa.values())
        for n in pa:  # (the key bias has an exactly zero gradient — softmax is shift invariant: absolute floor)
            assert float((pb[n] - pa[n]).norm()) < 2e-4 * float(pa[n].norm()) + 1e-6 * scale, (n, _rel(pb[n], pa[n]))
    else:
        o4 = ob.view(B, L, K, C)
        assert float((o4[:, 0] - o4[:, 1]).abs().max()) > 1e-3            # replicas differ: independent masks
        layer.zero_grad()
        x = q.clone().requires_grad_(True)
        f = lambda v: (layer.forward_tiled(v, L, kv, kv) * g).sum()
        an._CALLS[0] = 100
        f(x).backward()
        u = torch.randn_like(q)
        u /= u.norm()
        h = 1e-2
        an._CALLS[0] = 100
        lp = float(f(q + h * u))
        an._CALLS[0] = 100
        lm = float(f(q - h * u))
        fd, an_ = (lp - lm) / (2 * h), float((x.grad * u).sum())
        assert abs(fd - an_) < 2e-2 * abs(an_) + 1e-2, (fd, an_)


@pytest.mark.parametrize("bf16", [False, True])
def test_deferred_slab_reduce_equals_immediate(bf16):
    """bf16=True additionally queues the weight-gradient LAUNCHES of the plain linear layers (vlp3d_linear_wgrad_batch; same
    kernel body, run at the flush).
    Weight gradients with every slab sum of the backward pass batched into one launch at its end
    (_lib.deferred_slab_reduce, vlp3d_slab_reduce_batch) equal the per-layer reduce launches: SA stacks incl. the
    [xyz | features] column rotation of the gather layer, rows stacks with K slices and bias sums, nn.Linear [dW | db]."""
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    synth = importlib.import_module("3dvlp_amd.synth")
    ext = importlib.import_module("3dvlp_amd._lib")
    devc = torch.device("cuda:0")
    batch = gs.batch_to_device(synth.make_batch(0, 2, num_points=8192, lang_num_max=2), devc)
    batch["random"] = torch.tensor(0.25, device=devc)
    step = gs.GroundingStep(devc, sa_dtype=torch.bfloat16) if bf16 else gs.GroundingStep(devc)
    _eval_dropout_train_bn(step)
    state = [b.clone() for b in step.model.buffers()]
    grads = []
    for deferred in (False, False, False, True):  # the first three runs give the run-to-run noise of the float atomics upstream
        for b, s0 in zip(step.model.buffers(), state):
            b.copy_(s0)
        step.bucket.zero()
        loss, _ = step.forward_loss(batch, None)
        if deferred:
            with ext.deferred_slab_reduce() as q:
                q.FLUSH_BYTES = 1 << 40  # everything in one batch (more than 40 entries: two launches)
                loss.backward()
                assert q.added > 20 and (len(q.wjobs) >= 20) == bf16
        else:
            loss.backward()
        step.bucket.collect()
        torch.cuda.synchronize()
        grads.append({n: p.grad.clone() for n, p in step.model.named_parameters() if p.grad is not None})
    assert grads[0].keys() == grads[3].keys()
    # Run-to-run noise: the fp32 path differs by the summation order of float atomics (1e-6).  With bf16 storage such an
    # ulp-level difference flips the rounding of a few stored gradient elements (4e-3 each), which the layers below amplify:
    # identical backbone inputs give weight gradients 1e-4 .. 5e-4 apart (Frobenius) from one pass to the next, and two
    # passes are often bit-identical while the third is not (tools/sa_determinism.py) — hence three baseline runs and a floor.
    # (the floor is a bound on that noise, not on the queue: a slab missing from the batched sum is an O(1) error, and the
    # bit-equality of batched and per-layer launches on identical inputs is test_linear_wgrad_batch_equals_single_launches'.
    # One run in ~8 of the full suite exceeded 2e-3 once the atomic-ordered inverse maps fed the SA backward.)
    floor = 5e-3 if bf16 else 2e-5
    for n, g0 in grads[0].items():
        noise = max(_rel(grads[1][n], g0), _rel(grads[2][n], g0), _rel(grads[2][n], grads[1][n]))
        g1 = grads[3][n]
        assert _rel(g1, g0) <= max(floor, 5 * noise) or float((g1 - g0).abs().max()) < 1e-7, (n, _rel(g1, g0), noise)


def test_linear_wgrad_batch_equals_single_launches():
    """Weight gradients of many plain linear layers queued during backward and run as ONE launch at the flush
    (vlp3d_linear_wgrad_batch: every register bucket, with / without bias, row counts from 64 to 16 384) are bit-equal to the
    per-layer launches (same body, same slabs), and the immediate mode is untouched by the queue code."""
    ml = importlib.import_module("3dvlp_amd.mfma_linear")
    ext = importlib.import_module("3dvlp_amd._lib")
    torch.manual_seed(11)
    shapes = [(2048, 128, 128, True), (2048, 128, 384, True), (16384, 128, 256, True), (3200, 256, 128, False),
              (64, 32, 64, True), (2048, 64, 128, True), (4096, 256, 512, True), (2048, 128, 64, False)] * 7  # 56 > 48 jobs
    xs = [torch.randn(R, K, device="cuda", requires_grad=True) for R, K, N, b in shapes]
    ws = [(torch.randn(N, K, device="cuda") * 0.1).requires_grad_(True) for R, K, N, b in shapes]
    bs = [torch.randn(N, device="cuda").requires_grad_(True) if b else None for R, K, N, b in shapes]
    gy = [torch.randn(R, N, device="cuda") for R, K, N, b in shapes]
    assert all(ml.supported(x, w) for x, w in zip(xs, ws))

    def run(batched, deferred, blocks=None):
        for t in xs + ws + [b for b in bs if b is not None]:
            t.grad = None
        old = ml.BATCH_WGRAD, ml.BATCH_WGRAD_BLOCKS
        ml.BATCH_WGRAD = batched
        ml.BATCH_WGRAD_BLOCKS = ml.WGRAD_BLOCKS if blocks is None else blocks  # same row groups: same summation order
        try:
            with ml.bf16_mma(True):
                loss = sum((ml.linear(x, w, b) * g).sum() for x, w, b, g in zip(xs, ws, bs, gy))
            if deferred:
                with ext.deferred_slab_reduce() as q:
                    loss.backward()
                    assert (len(q.wjobs) > 0) == batched
            else:
                loss.backward()
        finally:
            ml.BATCH_WGRAD, ml.BATCH_WGRAD_BLOCKS = old
        torch.cuda.synchronize()
        return [t.grad.clone() for t in ws + [b for b in bs if b is not None] + xs]

    ref = run(False, False)
    for batched, deferred in ((False, True), (True, True), (True, False)):
        got = run(batched, deferred)
        for a, b in zip(got, ref):
            assert torch.equal(a, b)
    for a, b in zip(run(True, True, blocks=32), ref):  # the default inside a batch: fewer, longer row groups per layer
        assert _rel(a, b) < 1e-5
    want = (gy[0].double().t() @ xs[0].detach().double())
    assert _rel(ref[0].double(), want) < 2e-2  # bf16 operands


@pytest.mark.parametrize("kind,p", [("relu", 0.1), ("gelu", 0.5), ("gelu", 0.0)])
def test_act_dropout_equals_torch_with_same_mask(kind, p):
    """dropout(relu(.)) (attention.py:104-112) / Dropout(GELU(.)) (match_module.py:40-47) as one launch each way: values and
    gradient against torch evaluating the same formula with the kernel's own keep mask; keep rate; mask redraw."""
    an = importlib.import_module("3dvlp_amd.add_norm")
    torch.manual_seed(4)
    z = torch.randn(4096, 128, device="cuda", requires_grad=True)
    mask = torch.empty(z.shape, dtype=torch.uint8, device="cuda")
    out = an.act_dropout(z, kind, p, True, mask_out=mask)
    g = torch.randn_like(out)
    out.backward(g)
    z2 = z.detach().clone().requires_grad_(True)
    act = torch.relu(z2) if kind == "relu" else torch.nn.functional.gelu(z2)
    ref = act * mask.float() / (1.0 - p)
    ref.backward(g)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(z.grad, z2.grad, rtol=1e-4, atol=1e-6)
    if p > 0:
        assert abs(mask.float().mean().item() - (1 - p)) < 0.01
        an.advance(z.device)
        mask2 = torch.empty_like(mask)
        an.act_dropout(z.detach(), kind, p, True, mask_out=mask2)
        assert (mask2 != mask).float().mean().item() > 0.05
    else:
        assert bool(mask.all())
    torch.testing.assert_close(an.act_dropout(z.detach(), kind, p, False), torch.relu(z.detach()) if kind == "relu"
                               else torch.nn.functional.gelu(z.detach()), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("N,rows", [(64, 250), (128, 256), (256, 250)])
@pytest.mark.parametrize("deferred", [False, True])
def test_small_linears_equal_f_linear(deferred, N, rows):
    """Linear(27,128) added onto a base (relation_module.py:64,118) and Linear(128,1) (match_module.py:47) on the glue
    kernels vs F.linear in fp64: values, input / base gradients, weight and bias gradients (immediate and queued sums)."""
    glue = importlib.import_module("3dvlp_amd.glue")
    ext = importlib.import_module("3dvlp_amd._lib")
    torch.manual_seed(5)
    dev = "cuda"
    x = torch.randn(8, rows, 27, device=dev)   # (8 x 250 rows: the last 64-row block of the backward kernel is partial)
    W = (torch.randn(N, 27, device=dev) * 0.2).requires_grad_(True)
    b = torch.randn(N, device=dev).requires_grad_(True)
    base = torch.randn(8, rows, N, device=dev).requires_grad_(True)
    xr = torch.randn(3000, 128, device=dev).requires_grad_(True)
    wr = (torch.randn(1, 128, device=dev) * 0.2).requires_grad_(True)
    br = torch.randn(1, device=dev).requires_grad_(True)
    assert glue.smallk_supported(x, W) and glue.rowdot_supported(xr, wr)

    def run():
        o1 = glue.small_linear(x, W, b, base=base)
        o2 = glue.rowdot(xr, wr, br)
        return o1, o2
    o1, o2 = run()
    g1, g2 = torch.randn_like(o1), torch.randn_like(o2)
    if deferred:
        with ext.deferred_slab_reduce():
            torch.autograd.backward([o1, o2], [g1, g2])
    else:
        torch.autograd.backward([o1, o2], [g1, g2])
    got = [t.grad.clone() for t in (W, b, base, xr, wr, br)]
    d = lambda t: t.detach().double().requires_grad_(True)
    W64, b64, base64, xr64, wr64, br64 = (d(t) for t in (W, b, base, xr, wr, br))
    r1 = base64 + torch.nn.functional.linear(x.double(), W64, b64)
    r2 = torch.nn.functional.linear(xr64, wr64, br64)
    torch.autograd.backward([r1, r2], [g1.double(), g2.double()])
    torch.testing.assert_close(o1.double(), r1.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(o2.double(), r2.detach(), rtol=1e-5, atol=1e-5)
    for a, r in zip(got, (W64, b64, base64, xr64, wr64, br64)):
        assert _rel(a, r.grad) < 1e-5, _rel(a, r.grad)


@pytest.mark.parametrize("R,dims,last_plain", [(4096, [512, 256, 256], False), (2048, [128, 128, 128, 28], True)])
def test_row_stack_bf16_mma_close_to_fp32(R, dims, last_plain):
    """The timing-configuration form of the rows stacks (bf16 MFMA operands, fp32 I/O / statistics / accumulation) stays
    within bf16 operand rounding of the exact-fp32 form: the output norm-wise at 1e-2; gradients at 0.12 — on random
    Gaussian rows the BatchNorm backward subtracts two nearly equal column means, which amplifies the operand rounding
    (the bf16 grouped MLPs show the same against their autocast yardstick, test_sa_module_mfma_bf16_close_to_fp32)."""
    rm = importlib.import_module("3dvlp_amd.row_mlp")
    ml = importlib.import_module("3dvlp_amd.mfma_linear")
    torch.manual_seed(R)
    L = len(dims) - 1
    Ws = [torch.randn(dims[i + 1], dims[i], device="cuda") * 0.1 for i in range(L)]
    bs = [torch.randn(dims[i + 1], device="cuda") * 0.1 for i in range(L)]
    x0, go = torch.randn(R, dims[0], device="cuda"), torch.randn(R, dims[-1], device="cuda")
    res = []
    for bf in (False, True):
        torch.manual_seed(1)
        bns = [None if (last_plain and i == L - 1) else torch.nn.BatchNorm1d(dims[i + 1]).cuda().train() for i in range(L)]
        x = x0.clone().requires_grad_(True)
        W = [w.clone().requires_grad_(True) for w in Ws]
        b = [t.clone().requires_grad_(True) for t in bs]
        with ml.bf16_mma(bf):
            y = rm.row_stack(x, [(W[i], b[i], bns[i]) for i in range(L)])
        (y * go).sum().backward()
        res.append([y.detach(), x.grad] + [w.grad for w in W] + [bn.weight.grad for bn in bns if bn is not None])
    assert 1e-5 < _rel(res[1][0], res[0][0]) < 1e-2  # close, and the switch did something
    for a_, e_ in zip(res[1][1:], res[0][1:]):
        assert _rel(a_, e_) < 0.12, _rel(a_, e_)


@pytest.mark.parametrize("training", [True, False])
def test_row_stack_final_prelu_equals_fp64_formula(training):
    """Conv1d -> BatchNorm1d -> PReLU(C) (relation_module.py:47, features_concat) on the rows kernels: output, input gradient,
    weight / bias / BatchNorm / slope gradients against the same chain in fp64."""
    rm = importlib.import_module("3dvlp_amd.row_mlp")
    torch.manual_seed(11)
    R, K, N = 2048, 128, 128
    x0 = torch.randn(R, K, device="cuda")
    W0, b0 = torch.randn(N, K, device="cuda") * 0.1, torch.randn(N, device="cuda") * 0.1
    a0 = torch.rand(N, device="cuda") * 0.5
    bn = torch.nn.BatchNorm1d(N).cuda().train(training)
    _randomise_bn(bn)
    go = torch.randn(R, N, device="cuda")
    x, W, b, a = (t.clone().requires_grad_(True) for t in (x0, W0, b0, a0))
    y = rm.row_stack(x, [(W, b, bn)], final_slope=a)
    (y * go).sum().backward()
    xd, Wd, bd, ad = (t.double().requires_grad_(True) for t in (x0, W0, b0, a0))
    gd, ed = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    h = xd @ Wd.t() + bd
    mean, var = (h.mean(0), h.var(0, unbiased=False)) if training else (bn.running_mean.double(), bn.running_var.double())
    v = (h - mean) / torch.sqrt(var + bn.eps) * gd + ed
    ref = torch.where(v > 0, v, ad * v)
    (ref * go.double()).sum().backward()
    assert _rel(y, ref) < 1e-5
    for got, exp in ((x.grad, xd.grad), (W.grad, Wd.grad), (bn.weight.grad, gd.grad), (bn.bias.grad, ed.grad), (a.grad, ad.grad)):
        assert _rel(got, exp) < 2e-4, _rel(got, exp)
    if not training:
        assert _rel(b.grad, bd.grad) < 2e-4


@pytest.mark.parametrize("S,need_xyz_grad,C", [(64, False, 60), (32, True, 60), (16, True, 256), (16, False, 256), (32, False, 128)])
def test_sa_compact_rows_equal_dense_rows(S, need_xyz_grad, C):
    """The grouped MLP on the DISTINCT rows of every ball (csrc/sa_compact.hip: ball-query padding removed, multiplicities
    in the BatchNorm sums and the BatchNorm-backward term) equals the padded evaluation: pooled output, running
    statistics, every parameter gradient and the input gradients — at the tolerance of two bf16 runs that differ only in
    the order of their float additions."""
    pm = importlib.import_module("3dvlp_amd.pointnet2_modules")
    synth = importlib.import_module("3dvlp_amd.synth")
    ext = importlib.import_module("3dvlp_amd._lib")
    torch.manual_seed(21)
    B, N = 2, 8192  # C + 3 padded to 64 / 288 columns: widths the scatter kernels are instantiated for
    xyz = torch.from_numpy(np.stack([synth.make_scene(30 + i, N)["xyz"] for i in range(B)]).astype(np.float32)).cuda()
    feats0 = torch.randn(B, C, N, device="cuda")
    res = []
    for compact in (False, True):
        torch.manual_seed(5)
        m = pm.PointnetSAModuleVotes(npoint=512, radius=0.3, nsample=S, mlp=[C, 64, 64, 128] if C < 128 else [C, 128, 128, 256], use_xyz=True,
                                     normalize_xyz=True).cuda().train()
        m.mlp_dtype, m.compact = torch.bfloat16, compact
        x = xyz.clone().requires_grad_(need_xyz_grad)
        f = feats0.clone().requires_grad_(True)
        geo = m.compute_geometry(x.detach())
        assert (len(geo) == 5) == compact
        if compact:
            rowptr, crow = geo[3], geo[4]
            P = int(rowptr[-1])
            assert 0 < P < B * 512 * S and int(crow[:P, 2].view(torch.float32).sum().round()) == B * 512 * S  # multiplicities
        new_xyz, out, _ = m(x, f, geometry=geo)
        g = torch.randn_like(out)
        out.backward(g)
        res.append(dict(out=out.detach(), df=f.grad, dx=x.grad if need_xyz_grad else None,
                        params=[p.grad for p in m.parameters()], bufs=[b.clone() for b in m.buffers()]))
    a, b = res
    if not need_xyz_grad:
        # ... and the atomic-free backward of the gather layer (csrc/sa_gather_sum.hip: per-point gather of the layer-1 rows
        # through the inverse map + one product) against the scatter form, same compact rows: d(features) to bf16 rounding
        torch.manual_seed(5)
        m = pm.PointnetSAModuleVotes(npoint=512, radius=0.3, nsample=S, mlp=[C, 64, 64, 128] if C < 128 else [C, 128, 128, 256], use_xyz=True,
                                     normalize_xyz=True).cuda().train()
        m.mlp_dtype, m.compact, m.csr_backward = torch.bfloat16, True, True
        f = feats0.clone().requires_grad_(True)
        geo = m.compute_geometry(xyz)
        assert len(geo) == 7 and int(geo[5][-1]) == int(geo[3][-1])      # every compact row is listed under exactly one point
        _, out, _ = m(xyz, f, geometry=geo)
        out.backward(g)
        assert _rel(out.detach(), b["out"]) < 1e-6
        assert _rel(f.grad, b["df"]) < 6e-3, _rel(f.grad, b["df"])
        for pc, pb in zip([p.grad for p in m.parameters()], b["params"]):
            assert _rel(pc, pb) < 1e-5
    assert _rel(b["out"], a["out"]) < 2e-3
    assert _rel(b["df"], a["df"]) < 2e-2
    if need_xyz_grad:
        assert _rel(b["dx"], a["dx"]) < 2e-2
    for pa, pb in zip(a["params"], b["params"]):
        assert _rel(pb, pa) < 2e-2, _rel(pb, pa)
    for ba, bb in zip(a["bufs"], b["bufs"]):
        if ba.dtype.is_floating_point:
            assert _rel(bb, ba) < 1e-4


@pytest.mark.parametrize("compact", [True, False])
@pytest.mark.parametrize("mlp,S", [([60, 64, 64, 128], 64), ([128, 128, 128, 256], 32), ([256, 128, 128, 128], 16)])
def test_sa_last_layer_backward_without_its_preactivation(mlp, S, compact):
    """csrc/sa_last.hip (round 4): the last layer's weight gradient and the masked gradient of the layer below from Y2 and the
    balls' pooled rows alone — dY3 = k1 G - w (alpha + beta y3) with y3 = a2 W3^T, so dA2 = (k1 G) W3 - w (W3^T alpha + a2 Q) and
    dW3 = (k1 G)^T a2 - alpha (x) s - diag(beta) W3 M — against (i) the round-3 kernels that read Y3 (VLP3D_SA_LAST=0: the same bf16
    configuration, so the two differ by bf16 rounding of different intermediates) and (ii) the exact-fp32 configuration: every
    parameter gradient and the feature gradient of the new form are as close to fp32 as the old form's (x 1.25 + 2e-3), on
    compact and on padded rows, for the three (C2, C3) shapes of the path."""
    pm = importlib.import_module("3dvlp_amd.pointnet2_modules")
    sf = importlib.import_module("3dvlp_amd.sa_fused")
    synth = importlib.import_module("3dvlp_amd.synth")
    B, N, C = 2, 8192, mlp[0]
    xyz = torch.from_numpy(np.stack([synth.make_scene(40 + i, N)["xyz"] for i in range(B)]).astype(np.float32)).cuda()
    torch.manual_seed(3)
    feats0 = torch.randn(B, C, N, device="cuda")
    runs = {}
    g = None
    for name, dtype, last in (("fp32", None, False), ("old", torch.bfloat16, False), ("new", torch.bfloat16, True)):
        torch.manual_seed(5)
        m = pm.PointnetSAModuleVotes(npoint=512, radius=0.3, nsample=S, mlp=list(mlp), use_xyz=True, normalize_xyz=True).cuda().train()
        m.mlp_dtype, m.compact = dtype, compact
        with torch.no_grad():   # BatchNorm parameters away from (1, 0): alpha / beta / k1 all matter
            for bn in [l.bn.bn for l in m.mlp_module]:
                bn.weight.uniform_(0.5, 1.5)
                bn.bias.uniform_(-0.3, 0.3)
        f = feats0.clone().requires_grad_(True)
        old = sf.SA_LAST, sf.SA_LAST_WGRAD_MIN_ROWS, sf.SA_LAST_DGRAD_MIN_ROWS
        sf.SA_LAST, sf.SA_LAST_WGRAD_MIN_ROWS, sf.SA_LAST_DGRAD_MIN_ROWS = last, 0, 0   # both kernels whatever the module's size
        try:
            _, out, _ = m(xyz, f)
            if g is None:
                g = torch.randn_like(out)
            out.backward(g)
        finally:
            sf.SA_LAST, sf.SA_LAST_WGRAD_MIN_ROWS, sf.SA_LAST_DGRAD_MIN_ROWS = old
        runs[name] = dict(out=out.detach(), df=f.grad, params=[p.grad for p in m.parameters()])
    assert _rel(runs["new"]["out"], runs["old"]["out"]) < 1e-6          # same forward
    names = [n for n, _ in m.named_parameters()]
    worst = 0.0
    for n, pf, po, pn in zip(names, runs["fp32"]["params"], runs["old"]["params"], runs["new"]["params"]):
        eo, en = _rel(po, pf), _rel(pn, pf)
        worst = max(worst, en)
        assert en <= 1.25 * eo + 2e-3, (n, en, eo)
        assert _rel(pn, po) < 3e-2, (n, _rel(pn, po))
    eo, en = _rel(runs["old"]["df"], runs["fp32"]["df"]), _rel(runs["new"]["df"], runs["fp32"]["df"])
    assert en <= 1.25 * eo + 2e-3, (en, eo)
    diff = max(_rel(pn, po) for po, pn in zip(runs["old"]["params"], runs["new"]["params"]))
    print(f"mlp {mlp} S {S} compact {compact}: worst parameter-gradient error vs fp32 {worst:.2e}; d(features) old {eo:.2e} new {en:.2e}; "
          f"new vs old: parameters {diff:.2e}, d(features) {_rel(runs['new']['df'], runs['old']['df']):.2e}")


@pytest.mark.parametrize("compact", [True, False])
@pytest.mark.parametrize("C", [132, 128, 12])
def test_sa_gather_layer_reads_bf16_feature_rows(C, compact):
    """include/vlp3d.h bf16_io bit 1 (round 4): the first grouped-MLP layer and its weight gradient read the loader's BF16 copy
    of the feature channels ((B, N, round_up(C, 8)) zero-padded rows) instead of the fp32 rows.  The fp32 rows are rounded to
    bf16 (nearest even) on their way into LDS, so with features that ARE bf16 values both forms must agree to the bit: pooled
    output and every parameter gradient (the weight-gradient slabs are summed in the same order)."""
    pm = importlib.import_module("3dvlp_amd.pointnet2_modules")
    synth = importlib.import_module("3dvlp_amd.synth")
    B, N = 2, 8192
    xyz = torch.from_numpy(np.stack([synth.make_scene(60 + i, N)["xyz"] for i in range(B)]).astype(np.float32)).cuda()
    torch.manual_seed(C)
    feats = torch.randn(B, N, C, device="cuda").to(torch.bfloat16)          # representable: the rounding is the identity
    rows = torch.zeros(B, N, (C + 7) // 8 * 8, dtype=torch.bfloat16, device="cuda")
    rows[..., :C] = feats
    res, g = [], None
    for form in ("fp32", "bf16"):
        torch.manual_seed(11)
        m = pm.PointnetSAModuleVotes(npoint=512, radius=0.3, nsample=32, mlp=[C, 64, 64, 128], use_xyz=True, normalize_xyz=True).cuda().train()
        m.mlp_dtype, m.compact = torch.bfloat16, compact
        if form == "fp32":
            _, out, _ = m(xyz, feats.float().transpose(1, 2))
        else:
            _, out, _ = m(xyz, None, feat_rows_bf16=(rows, C))
        if g is None:
            g = torch.randn_like(out)
        out.backward(g)
        res.append((out.detach().clone(), [p.grad.clone() for p in m.parameters()], [n for n, _ in m.named_parameters()]))
    assert torch.equal(res[0][0], res[1][0])
    for n, a, b in zip(res[0][2], res[0][1], res[1][1]):
        assert torch.equal(a, b), (n, float((a - b).abs().max()))
    # the exact-fp32 configuration takes the same rows through the widening fall-back
    m = pm.PointnetSAModuleVotes(npoint=512, radius=0.3, nsample=32, mlp=[C, 64, 64, 128], use_xyz=True, normalize_xyz=True).cuda().train()
    torch.manual_seed(11)
    m2 = pm.PointnetSAModuleVotes(npoint=512, radius=0.3, nsample=32, mlp=[C, 64, 64, 128], use_xyz=True, normalize_xyz=True).cuda().train()
    m2.load_state_dict(m.state_dict())
    _, o1, _ = m(xyz, feats.float().transpose(1, 2))
    _, o2, _ = m2(xyz, None, feat_rows_bf16=(rows, C))
    assert torch.equal(o1, o2)


@pytest.mark.parametrize("compact", [True, False])
def test_sa_levels_hand_their_pooled_rows_on_as_bf16(compact):
    """vlp3d_sa_pool_rows + bf16_io bit 1 between levels (round 4): a level also writes its pooled output as bf16 rows, the next
    level's gather layer (K = 144: the fast path; K = 272: the wide instantiation) and its weight gradient read those instead
    of rounding the fp32 rows themselves.  Three chained levels with and without the hand-over: the same bits everywhere —
    the final features and the last level's parameter gradients; everything upstream of the last level's float-atomic
    feature gradient to the run-to-run noise of the backward pass."""
    pm = importlib.import_module("3dvlp_amd.pointnet2_modules")
    synth = importlib.import_module("3dvlp_amd.synth")
    B, N, C = 2, 8192, 12
    xyz = torch.from_numpy(np.stack([synth.make_scene(70 + i, N)["xyz"] for i in range(B)]).astype(np.float32)).cuda()
    torch.manual_seed(2)
    feats0 = torch.randn(B, C, N, device="cuda")
    res, g = [], None
    for hand_over in (False, True):
        torch.manual_seed(21)
        levels = [pm.PointnetSAModuleVotes(npoint=2048, radius=0.2, nsample=32, mlp=[C, 64, 64, 128], use_xyz=True, normalize_xyz=True),
                  pm.PointnetSAModuleVotes(npoint=1024, radius=0.4, nsample=32, mlp=[128, 128, 128, 256], use_xyz=True, normalize_xyz=True),
                  pm.PointnetSAModuleVotes(npoint=512, radius=0.8, nsample=16, mlp=[256, 128, 128, 256], use_xyz=True, normalize_xyz=True)]
        for m in levels:
            m.cuda().train()
            m.mlp_dtype, m.compact, m.pass_rows_bf16 = torch.bfloat16, compact, hand_over
        f = feats0.clone().requires_grad_(True)
        x, h = xyz, f
        carried = []
        for m in levels:
            x, h, _ = m(x, h)
            carried.append(getattr(h, "_vlp3d_rows_bf16", None) is not None)
        assert carried == [hand_over] * 3
        if g is None:
            g = torch.randn_like(h)
        h.backward(g)
        res.append((h.detach().clone(), f.grad.clone(), [p.grad.clone() for m in levels for p in m.parameters()]))
    assert torch.equal(res[0][0], res[1][0])
    n3 = len(list(levels[2].parameters()))
    for i, (a, b) in enumerate(zip(res[0][2], res[1][2])):
        if i >= len(res[0][2]) - n3:   # the last level's gradients depend on the forward tensors and g alone: bit for bit
            assert torch.equal(a, b), (i, float((a - b).abs().max()))
        else:   # upstream of the last level's float-atomic feature gradient: run-to-run noise, amplified by the
            assert _rel(a, b) < 2e-2, (i, _rel(a, b))   # bf16 storage of the layer gradients
    assert _rel(res[0][1], res[1][1]) < 2e-2


@pytest.mark.parametrize("name,B,N", [("cfg3: 32 scenes per GPU", 32, 40000), ("cfg5: 80 000-point scenes", 4, 80000)])
def test_step_runs_at_other_baseline_shapes(name, B, N):
    """BASELINE.json cfg3 (batch 32 per GPU, epoch >= 50: OCC/OSC active) and cfg5 (80k-point scenes: pruned FPS with two slot
    states per lane, grid ball query) through the captured, pipelined bf16 step: finite losses, and the loss goes down on a
    repeated batch."""
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    synth = importlib.import_module("3dvlp_amd.synth")
    devc = torch.device("cuda:0")
    batch = gs.batch_to_device(synth.make_batch(0, B, num_points=N, lang_num_max=8), devc)
    step = gs.GroundingStep(devc, epoch=50, sa_dtype=torch.bfloat16, use_graph=True, pipeline=True)
    losses = [float(step.run(batch)) for _ in range(6)]
    torch.cuda.synchronize()
    assert all(np.isfinite(losses)), (name, losses)
    assert min(losses[3:]) < losses[0], (name, losses)


def _chain_parts(R, seed=0):
    torch.manual_seed(seed)
    lin = lambda n, k: torch.nn.Linear(k, n).cuda()
    mods = {"fo": lin(128, 128), "l1": lin(256, 128), "l2": lin(128, 256), "nx": lin(384, 128), "m1": lin(128, 128),
            "n1": torch.nn.LayerNorm(128).cuda(), "n2": torch.nn.LayerNorm(128).cuda()}
    for n in ("n1", "n2"):
        mods[n].weight.data.uniform_(0.5, 1.5)
        mods[n].bias.data.uniform_(-0.5, 0.5)
    return mods, torch.randn(R, 128, device="cuda"), torch.randn(R, 128, device="cuda")


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_row_chain_equals_unfused_modules(p):
    """[fc_o -> add & norm -> FFN linear1 -> ReLU -> dropout -> linear2 -> add & norm -> q|k|v projection -> Linear/GELU/Dropout]
    as ONE launch (csrc/rows_chain.hip) against the same modules as separate launches with the same dropout call ids (= the
    same masks): the first add & norm equals to fp32 round-off (only the order of the row statistics differs); after it a
    1e-7 difference can flip a bf16 operand rounding, so later tensors agree on all but a sprinkle of elements.  Backward of
    the chain = the q|k|v product's own launch, then ONE launch for the rest (vlp3d_rows_chain_bwd), which has to regenerate the
    forward launch's dropout masks (p = 0.1: a wrong mask is an O(1) error): every gradient agrees to bf16-flip noise."""
    an = importlib.import_module("3dvlp_amd.add_norm")
    ml = importlib.import_module("3dvlp_amd.mfma_linear")
    rc = importlib.import_module("3dvlp_amd.row_chain")
    R = 64 * 37
    m, a0, x0 = _chain_parts(R)
    g3, gq, gm = torch.randn(R, 128, device="cuda"), torch.randn(R, 384, device="cuda"), torch.randn(R, 128, device="cuda")
    params = [q for mod in m.values() for q in mod.parameters()]

    def unfused(a, x):
        y = ml.linear(a, m["fo"].weight, m["fo"].bias)
        x1 = an.add_norm(x, y, m["n1"], p, True)
        h = an.act_dropout(ml.linear(x1, m["l1"].weight, m["l1"].bias), "relu", p, True)
        x3 = an.add_norm(x1, ml.linear(h, m["l2"].weight, m["l2"].bias), m["n2"], p, True)
        qkv = ml.linear(x3, m["nx"].weight, m["nx"].bias)
        return x1, x3, qkv

    def chained(a, x):
        st = [rc.linear_add_norm(m["fo"].weight, m["fo"].bias, m["n1"], x, p), rc.linear(m["l1"].weight, m["l1"].bias, "relu", p),
              rc.linear_add_norm(m["l2"].weight, m["l2"].bias, m["n2"], ("tile", 1), p), rc.linear(m["nx"].weight, m["nx"].bias)]
        assert rc.supported(a, st)
        t = rc.run(a, st)
        return t[0], t[2], t[3]

    res = []
    for fn in (unfused, chained):
        an._CALLS[0] = 500
        a, x = a0.clone().requires_grad_(), x0.clone().requires_grad_()
        for q in params:
            q.grad = None
        with ml.bf16_mma(True):
            x1, x3, qkv = fn(a, x)
            gl = an.act_dropout(ml.linear(x3, m["m1"].weight, m["m1"].bias), "gelu", 0.5 if p else 0.0, True)
            ((x3 * g3).sum() + (qkv * gq).sum() + (gl * gm).sum()).backward()
        res.append(([x1.detach(), x3.detach(), qkv.detach()], [a.grad, x.grad] + [q.grad for q in params if q.grad is not None]))
    (fu, gu), (fc, gc) = res
    assert float((fu[0] - fc[0]).abs().max()) < 1e-5
    for u, c in zip(fu[1:], fc[1:]):
        d = (u - c).abs()
        assert float(d.max()) < 5e-2 and float((d > 1e-4).float().mean()) < 2e-3 and float(d.mean()) < 1e-5
    assert len(gu) == len(gc)
    for u, c in zip(gu, gc):
        assert float((u - c).norm()) < 1e-3 * float(u.norm()) + 1e-6, (u.shape, _rel(c, u))


@pytest.mark.parametrize("R", [100, 32, 1])
def test_row_chain_kernel_ragged_rows_gelu_and_wide_input(R):
    """The kernel itself on row counts that fill no tile, starting from a 256-column input, with a GELU stage and no dropout:
    against fp64 products of the bf16-rounded operands (what the MFMA computes), each stage checked from the stored tensors
    of the stage before it."""
    ext = importlib.import_module("3dvlp_amd._lib")
    torch.manual_seed(R)
    X = torch.randn(R, 256, device="cuda")
    W1, b1 = torch.randn(128, 256, device="cuda") * 0.1, torch.randn(128, device="cuda")
    W2, b2 = torch.randn(256, 128, device="cuda") * 0.1, torch.randn(256, device="cuda")
    W3 = torch.randn(128, 256, device="cuda") * 0.1
    gam, bet = torch.rand(128, device="cuda") + 0.5, torch.randn(128, device="cuda")
    res = torch.randn(R, 128, device="cuda")
    z1, h1 = torch.empty(R, 128, device="cuda"), torch.empty(R, 128, device="cuda")
    v2 = torch.empty(R, 256, device="cuda")
    out, xhat, rstd = torch.empty(R, 128, device="cuda"), torch.empty(R, 128, device="cuda"), torch.empty(R, device="cuda")
    ext.rows_chain(X, [dict(W=W1, bias=b1, N=128, K=256, v_out=z1, act_kind=1, h_out=h1),
                       dict(W=W2, bias=b2, N=256, K=128, v_out=v2),
                       dict(W=W3, bias=None, N=128, K=256, has_ln=1, res=res, gamma=gam, beta=bet, eps=1e-5, ln_out=out,
                            xhat=xhat, rstd=rstd)], None)
    r16 = lambda t: t.bfloat16().double()
    e1 = r16(X) @ r16(W1).t() + b1.double()
    torch.testing.assert_close(z1.double(), e1, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(h1.double(), torch.nn.functional.gelu(z1.double()), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(v2.double(), r16(h1) @ r16(W2).t() + b2.double(), rtol=1e-5, atol=1e-5)
    s = res.double() + r16(v2) @ r16(W3).t()
    mu, var = s.mean(1, keepdim=True), s.var(1, unbiased=False, keepdim=True)
    eh = (s - mu) / torch.sqrt(var + 1e-5)
    torch.testing.assert_close(xhat.double(), eh, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(out.double(), eh * gam.double() + bet.double(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rstd.double(), 1 / torch.sqrt(var + 1e-5).squeeze(1), rtol=1e-5, atol=0)


def test_row_chain_rejects_shapes_it_does_not_stage():
    ext = importlib.import_module("3dvlp_amd._lib")
    X = torch.randn(64, 128, device="cuda")
    y = torch.empty(64, 192, device="cuda")
    with pytest.raises(ext.Vlp3dError):   # 192 columns: not a multiple of the 128-column weight block
        ext.rows_chain(X, [dict(W=torch.randn(192, 128, device="cuda"), N=192, K=128, v_out=y)], None)
    with pytest.raises(ext.Vlp3dError):   # stage 1 reads 256 columns, stage 0 produced 128
        ext.rows_chain(X, [dict(W=torch.randn(128, 128, device="cuda"), N=128, K=128),
                           dict(W=torch.randn(128, 256, device="cuda"), N=128, K=256)], None)
    with pytest.raises(ext.Vlp3dError):   # dropout without a seed word
        ext.rows_chain(X, [dict(W=torch.randn(128, 128, device="cuda"), N=128, K=128, act_kind=0, act_p=0.1,
                                h_out=torch.empty(64, 128, device="cuda"))], None)
    with pytest.raises(ext.Vlp3dError):   # 512 columns only as the last stage's 384
        ext.rows_chain(X, [dict(W=torch.randn(512, 128, device="cuda"), N=512, K=128)], None)


def test_match_module_chained_decoder_equals_layer_modules():
    """MatchModule.forward with the decoder stack on the row chains (transformer.decoder_stack_chained) against the same module
    with the chains off (the layer modules' own launches), all dropout rates 0 so that neither draws masks.  Both are bf16-MFMA
    evaluations whose 1e-7 differences flip operand roundings and ReLU gates, so the yardstick is what bf16 itself costs: every
    gradient of the chained form is within half the distance between the module form and the exact-fp32 MFMA form (measured:
    a tenth for most, a quarter for the first layer's query / key projections, which sit behind both layers' noise)."""
    gr = importlib.import_module("3dvlp_amd.grounding")
    ml = importlib.import_module("3dvlp_amd.mfma_linear")
    rc = importlib.import_module("3dvlp_amd.row_chain")
    torch.manual_seed(5)
    B, L, K, C, T = 2, 4, 256, 128, 20
    mm = gr.MatchModule(num_proposals=K, hidden_size=C).cuda().train()
    for mod in mm.modules():
        if hasattr(mod, "fused_norm"):
            mod.fused_norm = True
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    feats = torch.randn(B, K, C, device="cuda")
    lang = torch.randn(B * L, T + 1, C, device="cuda")
    gconf, gfeat = torch.randn(B * L, K, device="cuda"), torch.randn(B * L, K, C, device="cuda")

    def run(bf, chain):
        rc.ENABLED = chain
        try:
            for mod in mm.modules():
                if hasattr(mod, "bf16_mma"):
                    mod.bf16_mma = bf
            mm.zero_grad()
            x = feats.clone().requires_grad_()
            dd = {"bbox_feature": x, "input_ids": torch.zeros(B, L, T + 1), "istrain": [0], "lang_fea": lang}
            launches = []
            orig = ext_mod.rows_chain
            ext_mod.rows_chain = lambda *a, **k: (launches.append(1), orig(*a, **k))[1]
            try:
                with ml.bf16_mma(bf):
                    dd = mm(dd)
                    ((dd["cluster_ref"] * gconf).sum() + (dd["cross_box_feature"] * gfeat).sum()).backward()
            finally:
                ext_mod.rows_chain = orig
            out = {"cluster_ref": dd["cluster_ref"].detach().double(), "cross_box_feature": dd["cross_box_feature"].detach().double(),
                   "d bbox_feature": x.grad.double()}
            out.update({n: p_.grad.double() for n, p_ in mm.named_parameters() if p_.grad is not None})
            return out, len(launches)
        finally:
            rc.ENABLED = True

    ext_mod = importlib.import_module("3dvlp_amd._lib")
    (exact, n0), (mods, n1), (chain, n2) = run(False, False), run(True, False), run(True, True)
    assert (n0, n1, n2) == (0, 0, 3)   # [layer 0 tail + layer 1 q|k|v], [layer 1 fc_o + fc_q], [layer 1 tail + match MLP]
    assert exact.keys() == mods.keys() == chain.keys()
    scale = max(float(v.norm()) for n, v in exact.items() if "." in n)
    for n in exact:
        if n.endswith("fc_k.bias"):  # exactly zero in exact arithmetic (softmax is shift invariant): all sides are round-off
            assert float(chain[n].norm()) < 1e-3 * scale
            continue
        bf_cost = float((mods[n] - exact[n]).norm())
        assert float((chain[n] - mods[n]).norm()) < 0.5 * bf_cost + 1e-6 * scale, (n, _rel(chain[n], mods[n]), _rel(mods[n], exact[n]))


@pytest.mark.parametrize("R", [100, 32, 7])
def test_row_chain_backward_kernel_against_fp64_arithmetic(R):
    """vlp3d_rows_chain_bwd on row counts that fill no tile: add & norm backward (residual gradient kept in registers) ->
    product (128 -> 256) -> ReLU backward -> product (256 -> 128) -> + base + kept -> add & norm backward (residual gradient
    stored) -> product -> result.  Every stage against fp64 arithmetic on the bf16-rounded operands it actually multiplied (the
    stored gradient of the stage before it), and the [dgamma | dbeta] slabs against the column sums."""
    ext = importlib.import_module("3dvlp_amd._lib")
    torch.manual_seed(R)
    cu = lambda *s: torch.randn(*s, device="cuda")
    G, xh0, xh2, z, base = cu(R, 128), cu(R, 128), cu(R, 128), cu(R, 256), cu(R, 128)
    rs0, rs2 = torch.rand(R, device="cuda") + 0.5, torch.rand(R, device="cuda") + 0.5
    gam0, gam2 = torch.rand(128, device="cuda") + 0.5, torch.rand(128, device="cuda") + 0.5
    Wt0, Wt1, Wt2 = cu(256, 128) * 0.1, cu(128, 256) * 0.1, cu(128, 128) * 0.1
    nblk = ext.rows_chain_bwd_blocks(R)
    e = lambda *s: torch.full(s, float("nan"), device="cuda")
    dy0, dz, dy2, dres2, gX = e(R, 128), e(R, 256), e(R, 128), e(R, 128), e(R, 128)
    part0, part2 = e(nblk, 2, 128), e(nblk, 2, 128)
    ext.rows_chain_bwd(G, [dict(op=1, aux=xh0, rstd=rs0, gamma=gam0, g_out=dy0, keep=1, part=part0),
                           dict(op=2, aux=z, act_kind=0, g_out=dz),
                           dict(base=base, add_kept=1, op=1, aux=xh2, rstd=rs2, gamma=gam2, g_out=dy2, dres_out=dres2, part=part2),
                           dict(op=0, g_out=gX)],
                       [dict(Wt=Wt0, N=256, K=128), dict(Wt=Wt1, N=128, K=256), dict(Wt=Wt2, N=128, K=128)], None)
    r16 = lambda t: t.bfloat16().double()

    def ln_bwd(v, xhat, rstd, gamma):
        gg = v * gamma.double()
        m1, m2 = gg.mean(1, keepdim=True), (gg * xhat.double()).mean(1, keepdim=True)
        return rstd.double()[:, None] * (gg - m1 - xhat.double() * m2)

    close = lambda a, b: torch.testing.assert_close(a.double(), b, rtol=1e-4, atol=1e-5)
    dx0 = ln_bwd(G.double(), xh0, rs0, gam0)
    close(dy0, dx0)
    close(part0.sum(0)[0], (G.double() * xh0.double()).sum(0))
    close(part0.sum(0)[1], G.double().sum(0))
    close(dz, (r16(dy0) @ r16(Wt0).t()) * (z > 0).double())
    v2 = r16(dz) @ r16(Wt1).t() + base.double() + dx0
    dx2 = ln_bwd(v2, xh2, rs2, gam2)
    close(dres2, dx2)
    close(dy2, dx2)
    close(part2.sum(0)[0], (v2 * xh2.double()).sum(0))
    close(part2.sum(0)[1], v2.sum(0))
    close(gX, r16(dy2) @ r16(Wt2).t())


def test_multi_head_attention_fc_o_add_norm_chain_equals_separate_launches():
    """MultiHeadAttention.forward in the bf16 configuration runs fc_o -> dropout -> add -> LayerNorm as ONE row-chain launch
    each way (the relation module's two self-attention layers, 2048 rows): against the separate launches (row chains off) with
    dropout 0, on the bf16 yardstick of test_match_module_chained_decoder_equals_layer_modules, with an additive attention bias
    that needs its gradient."""
    tr = importlib.import_module("3dvlp_amd.transformer")
    ml = importlib.import_module("3dvlp_amd.mfma_linear")
    rc = importlib.import_module("3dvlp_amd.row_chain")
    ext_mod = importlib.import_module("3dvlp_amd._lib")
    torch.manual_seed(21)
    B, K, C = 8, 256, 128
    mha = tr.MultiHeadAttention(d_model=C, d_k=32, d_v=32, h=4, dropout=0.0).cuda().train()
    mha.fused_norm = True
    mha.layer_norm.weight.data.uniform_(0.5, 1.5)
    mha.layer_norm.bias.data.uniform_(-0.5, 0.5)
    x0 = torch.randn(B, K, C, device="cuda")
    bias0 = torch.randn(B, 4, K, K, device="cuda") * 0.5
    g = torch.randn(B, K, C, device="cuda")

    def run(bf, chain):
        rc.ENABLED = chain
        try:
            mha.attention.bf16_mma = bf
            mha.zero_grad()
            x, bias = x0.clone().requires_grad_(), bias0.clone().requires_grad_()
            launches = []
            orig = ext_mod.rows_chain
            ext_mod.rows_chain = lambda *a, **k: (launches.append(1), orig(*a, **k))[1]
            try:
                with ml.bf16_mma(bf):
                    out = mha(x, x, x, attention_weights=bias, way="add")
                    (out * g).sum().backward()
            finally:
                ext_mod.rows_chain = orig
            res = {"out": out.detach().double(), "dx": x.grad.double(), "dbias": bias.grad.double()}
            res.update({n: p.grad.double() for n, p in mha.named_parameters() if p.grad is not None})
            return res, len(launches)
        finally:
            rc.ENABLED = True

    (exact, n0), (mods, n1), (chain, n2) = run(False, False), run(True, False), run(True, True)
    assert (n0, n1, n2) == (0, 0, 1)
    assert exact.keys() == mods.keys() == chain.keys()
    scale = max(float(v.norm()) for v in exact.values())
    for n in exact:
        if n.endswith("fc_k.bias"):
            continue
        bf_cost = float((mods[n] - exact[n]).norm())
        assert float((chain[n] - mods[n]).norm()) < 0.5 * bf_cost + 1e-6 * scale, (n, _rel(chain[n], mods[n]), _rel(mods[n], exact[n]))
