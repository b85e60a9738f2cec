"""Module-level parity against fixtures produced by the reference's own Python modules
(tests/golden/make_golden.py): same weights (loaded through load_state_dict, which also pins the
state_dict key contract), same inputs -> same outputs.

CPU variants exercise the host logic with the explicit impl='torch' attention; the GPU variants run the
product path (fused HIP attention, HIP pointnet2 ops)."""
import importlib

import numpy as np
import pytest
import torch


def _mods():
    return (importlib.import_module("3dvlp_amd.transformer"), importlib.import_module("3dvlp_amd.detection"),
            importlib.import_module("3dvlp_amd.grounding"), importlib.import_module("3dvlp_amd.pytorch_utils"))


def _load(module, g, prefix="w/"):
    sd = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in g.weights(prefix).items()}
    res = module.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all(k.endswith("num_batches_tracked") for k in res.missing_keys), res.missing_keys
    return module


DEVICES = [pytest.param("cpu", id="cpu-host-logic"), pytest.param("cuda", id="gpu", marks=pytest.mark.gpu)]


@pytest.fixture
def env(request, monkeypatch):
    device = request.param
    tr = importlib.import_module("3dvlp_amd.transformer")
    monkeypatch.setattr(tr, "DEFAULT_IMPL", "torch" if device == "cpu" else "hip")
    return torch.device(device)


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("env", DEVICES, indirect=True)
def test_multi_head_attention(env, golden):
    tr, _, _, _ = _mods()
    g = golden("attention")
    mha = _load(tr.MultiHeadAttention(d_model=128, d_k=32, d_v=32, h=4), g).to(env).eval()
    q, k = T(g["in/q"], env), T(g["in/k"], env)
    tol = dict(rtol=1e-4, atol=3e-5)
    with torch.no_grad():
        np.testing.assert_allclose(mha(q, k, k).cpu().numpy(), g["out/cross"], **tol)
        np.testing.assert_allclose(mha(q, q, q).cpu().numpy(), g["out/self"], **tol)
        out = mha(q, k, k, attention_weights=T(g["in/bias"], env), way="add")
        np.testing.assert_allclose(out.cpu().numpy(), g["out/add"], **tol)
        out = mha(q, k, k, attention_weights=T(g["in/wts"], env), way="mul")
        np.testing.assert_allclose(out.cpu().numpy(), g["out/mul"], **tol)
        out = mha(q, k, k, attention_mask=T(g["in/mask"], env))
        np.testing.assert_allclose(out.cpu().numpy(), g["out/mask"], **tol)
        out, att = mha(q, k, k, output_attn=True)  # the att-returning (unfused) form
        np.testing.assert_allclose(att.cpu().numpy(), g["out/cross_att"], rtol=1e-4, atol=1e-6)
        out, _ = mha.attention(q, k, k, need_att=False)
        np.testing.assert_allclose(out.cpu().numpy(), g["out/sdpa"], **tol)


@pytest.mark.parametrize("env", DEVICES, indirect=True)
def test_cross_attention_decoder_layers(env, golden):
    tr, _, _, _ = _mods()
    g = golden("decoder_layer")
    layers = _load(torch.nn.ModuleList(tr.CrossAttentionDecoderLayer(hidden_size=128) for _ in range(2)), g)
    layers = layers.to(env).eval()
    x, k = T(g["in/q"], env), T(g["in/k"], env)
    with torch.no_grad():
        for i in range(2):
            x = layers[i](x, k, k)
            np.testing.assert_allclose(x.cpu().numpy(), g[f"out/layer{i}"], rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize("env", DEVICES, indirect=True)
def test_match_module(env, golden):
    _, _, gr, _ = _mods()
    g = golden("match_module")
    m = _load(gr.MatchModule(num_proposals=32, lang_size=256, det_channel=128), g).to(env).eval()
    d = {"objectness_scores": T(g["in/objectness_scores"], env), "bbox_feature": T(g["in/bbox_feature"], env),
         "input_ids": torch.zeros(2, 2, 50, dtype=torch.long), "istrain": [0], "lang_fea": T(g["in/lang_fea"], env)}
    with torch.no_grad():
        d = m(d)
    np.testing.assert_allclose(d["cross_box_feature"].cpu().numpy(), g["out/cross_box_feature"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(d["cluster_ref"].cpu().numpy(), g["out/cluster_ref"], rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize("env", DEVICES, indirect=True)
def test_relation_module(env, golden):
    _, det, _, _ = _mods()
    g = golden("relation_module")
    m = _load(det.RelationModule(num_proposals=32, det_channel=128), g).to(env).eval()
    d = {k: T(g["in/" + k], env) for k in ("pred_bbox_feature", "pred_bbox_corner", "point_clouds", "seed_inds",
                                           "aggregated_vote_inds")}
    with torch.no_grad():
        d = m(d)
    np.testing.assert_allclose(d["dist_weights"].cpu().numpy(), g["out/dist_weights"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(d["bbox_feature"].cpu().numpy(), g["out/bbox_feature"], rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize("env", DEVICES, indirect=True)
def test_voting_module_train_and_eval(env, golden):
    _, det, _, _ = _mods()
    g = golden("voting_module")
    m = _load(det.VotingModule(1, 256), g).to(env)
    xyz, feat = T(g["in/seed_xyz"], env), T(g["in/seed_features"], env)
    with torch.no_grad():
        for mode in ("eval", "train"):
            m.train(mode == "train")
            vx, vf = m(xyz, feat)
            np.testing.assert_allclose(vx.cpu().numpy(), g[f"out/{mode}/vote_xyz"], rtol=1e-4, atol=3e-5)
            np.testing.assert_allclose(vf.cpu().numpy(), g[f"out/{mode}/vote_features"], rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize("env", DEVICES, indirect=True)
def test_shared_mlp_train_and_eval(env, golden):
    _, _, _, pt = _mods()
    g = golden("shared_mlp")
    m = _load(pt.SharedMLP([135, 64, 64, 128], bn=True), g).to(env)
    x = T(g["in/x"], env)
    with torch.no_grad():
        m.eval()
        np.testing.assert_allclose(m(x).cpu().numpy(), g["out/eval"], rtol=1e-4, atol=3e-5)
        m.train()
        np.testing.assert_allclose(m(x).cpu().numpy(), g["out/train"], rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize("env", DEVICES, indirect=True)
def test_roi_heads_and_box_decode(env, golden):
    _, det, gr, _ = _mods()
    g = golden("roi_heads")
    m = _load(det.StandardROIHeads(num_heading_bin=1, num_class=18, seed_feat_dim=256), g).to(env).eval()
    with torch.no_grad():
        d = m(T(g["in/features"], env), {})
    for k in ("objectness_scores", "rois", "heading_scores", "heading_residuals_normalized", "heading_residuals",
              "sem_cls_scores"):
        np.testing.assert_allclose(d[k].cpu().numpy(), g["out/" + k], rtol=1e-4, atol=2e-5)
    b = golden("boxes")
    corners = det.box_corners(T(b["size"], env), T(b["heading"], env), T(b["center"], env))
    np.testing.assert_allclose(corners.cpu().numpy(), b["corners"], rtol=1e-5, atol=2e-6)
    iou = gr.axis_aligned_iou(T(b["c1"], env), T(b["s1"], env), T(b["c2"], env), T(b["s2"], env))
    np.testing.assert_allclose(iou.cpu().numpy(), b["iou"], rtol=2e-5, atol=1e-7)


def test_contrast_module_matches_oracle_loop():
    """Batched ContrastModule == the oracle's literal restatement of the reference double loop."""
    from oracle import oracle as orc
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    gr = importlib.import_module("3dvlp_amd.grounding")
    rng = np.random.default_rng(3)
    B, K, L = 3, 40, 4
    cfg = gs.GroundingNet(use_con=False).dataset_config
    torch.manual_seed(1)
    cm = gr.ContrastModule(cfg)
    pred_center = rng.uniform(0, 3, (B, K, 3)).astype(np.float32)
    pred_size = rng.uniform(0.4, 1.5, (B, K, 3)).astype(np.float32)
    ref_center = pred_center[:, :L] + rng.normal(0, 0.1, (B, L, 3)).astype(np.float32)
    size_class = rng.integers(0, 18, (B, L))
    size_res = (pred_size[:, :L] - cfg.mean_size_arr[size_class]).astype(np.float32)
    feat = rng.normal(size=(B, K, 128)).astype(np.float32)
    obj = rng.normal(size=(B, K, 2)).astype(np.float32)
    obj[2, :, 1] = -10  # a scene without any positive proposal is skipped
    lang_emb = rng.normal(size=(B * L, 128)).astype(np.float32)
    lang_num = np.array([4, 2, 3])
    d = {"epoch": 50, "pred_center": torch.from_numpy(pred_center), "pred_size": torch.from_numpy(pred_size),
         "bbox_feature": torch.from_numpy(feat), "objectness_scores": torch.from_numpy(obj),
         "ref_center_label_list": torch.from_numpy(ref_center), "ref_size_class_label_list": torch.from_numpy(size_class),
         "ref_size_residual_label_list": torch.from_numpy(size_res), "lang_emb": torch.from_numpy(lang_emb),
         "lang_num": torch.from_numpy(lang_num)}
    d = cm(d)
    (d["lang_con_loss"] + d["iou_con_loss"]).backward()
    for p_ in (cm.pc_proj.weight, cm.text_proj.weight, cm.pc_proj_iou[0].weight):
        assert p_.grad is not None and torch.isfinite(p_.grad).all() and p_.grad.abs().sum() > 0
    W = {"pc_proj": cm.pc_proj.weight.detach().numpy(), "text_proj": cm.text_proj.weight.detach().numpy(),
         "pc_proj_iou": cm.pc_proj_iou[0].weight.detach().numpy()}
    gt_size = cfg.mean_size_arr[size_class] + size_res
    lang_num_eff = np.where(obj.argmax(-1).sum(1) > 0, lang_num, 0)
    occ, osc = orc.contrast_losses(W, pred_center, pred_size, feat, obj, ref_center, gt_size, lang_emb, lang_num_eff)
    assert abs(float(d["lang_con_loss"]) - occ) < 1e-5 * max(1, abs(occ))
    assert abs(float(d["iou_con_loss"]) - osc) < 1e-5 * max(1, abs(osc))
    d0 = cm({"epoch": 3})
    assert float(d0["con_loss"]) == 0.0  # no-op before epoch 50 (constrast_module.py:54-56)


@pytest.mark.gpu
@pytest.mark.parametrize("B,K,L,seed", [(3, 40, 4, 3), (8, 256, 8, 11), (2, 300, 5, 12)])
def test_contrast_fused_kernel_equals_batched_ops(B, K, L, seed):
    """csrc/contrast.hip (3 launches) vs the batched op-by-op ContrastModule (itself tested against the oracle's
    literal loop above): both losses and the gradients of the three projection weights and of the features."""
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    gr = importlib.import_module("3dvlp_amd.grounding")
    rng = np.random.default_rng(seed)
    cfg = gs.GroundingNet(use_con=False).dataset_config
    torch.manual_seed(seed)
    cm = gr.ContrastModule(cfg).cuda()
    pred_center = rng.uniform(0, 3, (B, K, 3)).astype(np.float32)
    pred_size = rng.uniform(0.4, 1.5, (B, K, 3)).astype(np.float32)
    ref_center = pred_center[:, :L] + rng.normal(0, 0.1, (B, L, 3)).astype(np.float32)
    size_class = rng.integers(0, 18, (B, L))
    size_res = (pred_size[:, :L] - cfg.mean_size_arr[size_class]).astype(np.float32)
    obj = rng.normal(size=(B, K, 2)).astype(np.float32)
    obj[B - 1, :, 1] = -10  # a scene without any positive proposal is skipped
    lang_num = rng.integers(1, L + 1, (B,))
    lang_num[0] = L
    base = {"epoch": 50, "pred_center": pred_center, "pred_size": pred_size, "objectness_scores": obj,
            "ref_center_label_list": ref_center, "ref_size_class_label_list": size_class,
            "ref_size_residual_label_list": size_res, "lang_emb": rng.normal(size=(B * L, 128)).astype(np.float32),
            "lang_num": lang_num}
    feat = rng.normal(size=(B, K, 128)).astype(np.float32)
    res = []
    for fused in (False, True):
        cm.fused = fused
        cm.zero_grad()
        d = {k: (torch.from_numpy(np.asarray(v)).cuda() if not isinstance(v, int) else v) for k, v in base.items()}
        f = torch.from_numpy(feat).cuda().requires_grad_(True)
        d["bbox_feature"] = f
        d = cm(d)
        (1.3 * d["lang_con_loss"] + 0.7 * d["iou_con_loss"]).backward()
        res.append([d["lang_con_loss"].detach(), d["iou_con_loss"].detach(), f.grad.clone(),
                    cm.pc_proj.weight.grad.clone(), cm.text_proj.weight.grad.clone(),
                    cm.pc_proj_iou[0].weight.grad.clone()])
    assert float(res[0][0]) > 0 and float(res[0][1]) > 0
    for a, b in zip(res[0], res[1]):
        assert (a - b).abs().max().item() < 1e-4 * a.abs().max().item() + 1e-6
    # ... and the HIP kernels DIRECTLY against the oracle's literal double loop (constrast_module.py:53-131): both losses,
    # and — the oracle is forward only — the feature gradient as a directional derivative of the oracle's loss (fp64
    # central difference along the kernel's own gradient direction and along a random one).
    from oracle import oracle as orc
    W = {"pc_proj": cm.pc_proj.weight.detach().cpu().numpy(), "text_proj": cm.text_proj.weight.detach().cpu().numpy(),
         "pc_proj_iou": cm.pc_proj_iou[0].weight.detach().cpu().numpy()}
    gt_size = cfg.mean_size_arr[size_class] + size_res
    lang_num_eff = np.where(obj.argmax(-1).sum(1) > 0, lang_num, 0)

    def oracle_loss(f):
        occ, osc = orc.contrast_losses(W, pred_center, pred_size, f, obj, ref_center, gt_size, base["lang_emb"],
                                       lang_num_eff)
        return occ, osc

    occ, osc = oracle_loss(feat)
    hip = res[1]
    assert abs(float(hip[0]) - occ) < 1e-5 * max(1, abs(occ)), (float(hip[0]), occ)
    assert abs(float(hip[1]) - osc) < 1e-5 * max(1, abs(osc)), (float(hip[1]), osc)
    g = hip[2].cpu().numpy().astype(np.float64)
    if K <= 64:  # the literal loop is O(B·L·K²) python: derivative check at the small shape only
        for direction in (g / np.linalg.norm(g), rng.normal(size=g.shape) / np.sqrt(g.size)):
            h = 1e-2
            lp = oracle_loss((feat + h * direction).astype(np.float64))
            lm = oracle_loss((feat - h * direction).astype(np.float64))
            fd = (1.3 * (lp[0] - lm[0]) + 0.7 * (lp[1] - lm[1])) / (2 * h)
            an = float((g * direction).sum())
            assert abs(fd - an) < 2e-3 * max(abs(an), 1e-3) + 1e-5, (fd, an)


def test_copy_paste_device_formulation_equals_reference_loop():
    """MatchModule._copy_paste (fixed-shape gather, no host sync) == the reference's host loop
    (models/refnet/match_module.py:97-121) restated literally."""
    gr = importlib.import_module("3dvlp_amd.grounding")
    torch.manual_seed(0)
    for trial in range(40):
        B, K, D = 4, 12, 5
        feats = torch.randn(B, K, D)
        masks = (torch.rand(B, K, 1) > (0.2 + 0.15 * (trial % 5))).float()
        if trial == 7:
            masks[:] = 0
        if trial == 8:
            masks[:] = 1
        got = gr.MatchModule._copy_paste(feats, masks)
        feature0 = feats.clone()
        obj_masks = masks.bool().squeeze(2)
        obj_lens = [int(obj_masks[i].sum()) for i in range(B)]
        obj_features = feats.reshape(B * K, -1)[obj_masks.reshape(-1)].repeat(2, 1)
        total_len = sum(obj_lens)
        j = 0
        for i in range(B):
            om = torch.where(obj_masks[i] == False)[0]  # noqa: E712
            j += obj_lens[i]
            if om.shape[0] < total_len - obj_lens[i]:
                feature0[i, om, :] = obj_features[j:j + om.shape[0], :]
            else:
                feature0[i, om[:total_len - obj_lens[i]], :] = obj_features[j:j + total_len - obj_lens[i], :]
        assert torch.equal(got, feature0), trial


@pytest.mark.parametrize("env", DEVICES, indirect=True)
def test_answer_module(env, golden):
    """QA head (answer_module.py:10-114, AttFlat of mcan_module.py:74-112): strict state-dict key contract + outputs."""
    ans = importlib.import_module("3dvlp_amd.answer")
    g = golden("answer_module")
    m = ans.AnswerModule(num_answers=24, hidden_size=128)
    sd = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in g.weights("w/").items()}
    m.load_state_dict(sd, strict=True)  # every reference key present, nothing extra
    m = m.to(env).eval()
    with torch.no_grad():
        d = m({"cross_box_feature": T(g["in/cross_box_feature"], env)})
    np.testing.assert_allclose(d["answer_scores"].cpu().numpy(), g["out/answer_scores"], rtol=1e-4, atol=5e-5)


@pytest.mark.gpu
def test_answer_module_train_backward_vs_torch_ops():
    """Train mode (hash dropout off: p = 0 set on the modules) on the MFMA kernels vs the same module on plain torch ops
    (CPU, fp64): every parameter gradient the head uses."""
    ans = importlib.import_module("3dvlp_amd.answer")
    torch.manual_seed(3)
    m = ans.AnswerModule(num_answers=24, hidden_size=128)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, ans.FC):
            mod.pdrop = 0.0
    ref = ans.AnswerModule(num_answers=24, hidden_size=128).double()
    ref.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    for mod in ref.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, ans.FC):
            mod.pdrop = 0.0
    x = torch.randn(64, 256, 128)
    g = torch.randn(64, 24)
    m = m.cuda().train()
    out = m({"cross_box_feature": x.cuda()})["answer_scores"]
    (out * g.cuda()).sum().backward()
    out_r = ref.train()({"cross_box_feature": x.double()})["answer_scores"]
    (out_r * g.double()).sum().backward()
    assert float((out.cpu().double() - out_r).norm() / out_r.norm()) < 1e-5
    for (n, p), (_, pr) in zip(m.named_parameters(), ref.named_parameters()):
        if pr.grad is None:
            assert p.grad is None, n
            continue
        # (the score bias has an exactly zero gradient — softmax is shift invariant — hence the absolute floor)
        assert float((p.grad.cpu().double() - pr.grad).norm()) < 1e-4 * float(pr.grad.norm()) + 1e-6, n
