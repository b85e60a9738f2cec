"""Input pipeline: scene sampling on the host, prefetching uploader on the device."""
import importlib

import numpy as np
import pytest
import torch


def test_sample_scene_follows_dataset_rule():
    """lib/joint/dataset.py:603-612: height = z - percentile(z, 0.99); choice with replacement only for small scenes;
    the same choice applied to the per-point labels."""
    ip = importlib.import_module("3dvlp_amd.input_pipeline")
    rng = np.random.default_rng(0)
    cloud = rng.normal(size=(5000, 6)).astype(np.float32)
    labels = np.arange(5000)
    out, lab = ip.sample_scene(cloud, 4000, np.random.default_rng(1), use_height=True, per_point=(labels,))
    assert out.shape == (4000, 7) and len(np.unique(lab)) == 4000            # no replacement when the scene is large enough
    np.testing.assert_allclose(out[:, :6], cloud[lab])
    np.testing.assert_allclose(out[:, 6], cloud[lab, 2] - np.percentile(cloud[:, 2], 0.99), rtol=1e-6)
    exp = np.random.default_rng(1).choice(5000, 4000, replace=False)
    assert (lab == exp).all()                                                # the reference's rng.choice call
    out2, lab2 = ip.sample_scene(cloud[:100], 4000, np.random.default_rng(2), use_height=False, per_point=(labels[:100],))
    assert out2.shape == (4000, 6) and lab2.max() < 100 and len(np.unique(lab2)) <= 100   # small scene: with replacement


@pytest.mark.gpu
def test_prefetcher_uploads_in_order_and_prepares_on_copy_stream():
    ip = importlib.import_module("3dvlp_amd.input_pipeline")
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    synth = importlib.import_module("3dvlp_amd.synth")
    host = [synth.make_batch(2 * i, 2, 4096, 2) for i in range(3)]
    host[1] = {k: torch.from_numpy(v).pin_memory() for k, v in host[1].items()}   # a pre-pinned batch is used as it is
    pf = ip.Prefetcher(iter(host), device="cuda", prepare=gs.prepare_batch)
    seen = []
    for i in range(3):
        d = pf.next()
        assert d is not None and d["point_clouds"].is_cuda and d["k/lang_num"].dtype == torch.int32
        ref = host[i]["point_clouds"]
        ref = ref.numpy() if torch.is_tensor(ref) else ref
        assert np.array_equal(d["point_clouds"].cpu().numpy(), ref)
        assert torch.equal(d["k/lang_kv"], d["lang_fea"][:, 1:])
        seen.append(d)
    assert pf.next() is None


def test_packed_staging_and_compressed_cloud_on_the_host():
    """Host logic of the uploader (no device needed): tensors below Prefetcher.PACK_BELOW bytes travel in ONE byte buffer and
    come back as views of the right dtype / shape; compress_cloud replaces the cloud by (k/xyz fp32, k/feat_bf bf16 = the
    round-to-nearest-even values of the feature channels) and leaves everything else alone."""
    ip = importlib.import_module("3dvlp_amd.input_pipeline")
    synth = importlib.import_module("3dvlp_amd.synth")
    pf = ip.Prefetcher.__new__(ip.Prefetcher)      # the staging helpers only (no stream, no upload)
    pf._pinned, pf._flip = [{}, {}], 0
    pf._stage = lambda k, v: v
    host = {"a": np.arange(10, dtype=np.int64).reshape(2, 5), "b": torch.randn(3, 4), "empty": torch.zeros(0),
            "big": torch.zeros(600000), "flag": [1], "h": torch.randn(7).half(), "u": torch.arange(5, dtype=torch.uint8)}
    staged, (buf, layout) = pf._stage_all(host)
    assert sorted(staged) == ["big", "empty", "flag"] and [l[0] for l in layout] == ["a", "b", "h", "u"]
    for k, off, n, dt, sh in layout:
        assert off % 256 == 0
        ref = torch.from_numpy(host[k]) if isinstance(host[k], np.ndarray) else host[k]
        assert torch.equal(buf[off:off + n].view(dt).view(sh), ref), k
    b = {k: torch.from_numpy(v) for k, v in synth.make_batch(0, 2, 4096, 2).items()}
    c = ip.compress_cloud(b)
    assert "point_clouds" not in c and c["k/xyz"].dtype == torch.float32 and c["k/feat_bf"].dtype == torch.bfloat16
    assert torch.equal(c["k/xyz"], b["point_clouds"][..., :3])
    C = b["point_clouds"].shape[-1] - 3
    assert c["k/feat_c"] == C and c["k/feat_bf"].shape[-1] == (C + 7) // 8 * 8 and c["k/feat_bf"].is_contiguous()
    assert torch.equal(c["k/feat_bf"][..., :C].float(), b["point_clouds"][..., 3:].to(torch.bfloat16).float())
    assert not c["k/feat_bf"][..., C:].float().any()          # 16-byte rows: the padding columns are zero
    assert all(torch.equal(c[k], b[k]) for k in b if k != "point_clouds")
    xyz_only = {"point_clouds": torch.zeros(2, 16, 3)}
    assert ip.compress_cloud(xyz_only) is xyz_only


@pytest.mark.gpu
def test_compressed_cloud_gives_the_same_bf16_step():
    """A batch uploaded through compress_cloud (the feature channels as bf16 rows on the link AND in the step: the gather
    layer of SA1, its weight gradient and the relation module read them directly), the same batch with the bf16 rows made on
    the device (prepare_batch(feat_bf16=True)) and the same batch as the fp32 cloud: the bf16 configuration rounds the gathered
    features to bf16 in front of the first product, so forward, loss and gradients are the same numbers (the loss to the bit:
    tools/determinism_probe.py)."""
    ip = importlib.import_module("3dvlp_amd.input_pipeline")
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    synth = importlib.import_module("3dvlp_amd.synth")
    hb = {k: torch.from_numpy(v) for k, v in synth.make_batch(0, 2, 20000, 4).items()}
    feeds = [ip.Prefetcher(iter([hb]), device="cuda", prepare=gs.prepare_batch),
             ip.Prefetcher(iter([ip.compress_cloud(hb)]), device="cuda", prepare=gs.prepare_batch),
             ip.Prefetcher(iter([hb]), device="cuda", prepare=lambda d: gs.prepare_batch(d, feat_bf16=True))]
    a, c, e = feeds[0].next(), feeds[1].next(), feeds[2].next()
    assert "point_clouds" not in c and "k/feat_pm" not in c and "k/feat_pm" not in e and torch.equal(c["k/xyz"], a["k/xyz"])
    C = a["k/feat_pm"].shape[-1]
    assert c["k/feat_c"] == C == e["k/feat_c"] and torch.equal(c["k/feat_bf"], e["k/feat_bf"])
    assert torch.equal(c["k/feat_bf"][..., :C].float(), a["k/feat_pm"].to(torch.bfloat16).float())
    for k in a:   # every other prepared tensor arrives unchanged through the packed upload
        if torch.is_tensor(a[k]) and k not in ("point_clouds", "k/feat_pm"):
            assert a[k].dtype == c[k].dtype and torch.equal(a[k], c[k]), k
    res = []
    for batch in (a, c, e):
        torch.manual_seed(0)
        step = gs.GroundingStep(torch.device("cuda:0"), epoch=50, sa_dtype=torch.bfloat16, lr=0.0, seed=0)
        loss = step.run(batch)
        torch.cuda.synchronize()
        res.append((float(loss), step.bucket.flat.clone()))
    for other in res[1:]:
        assert abs(res[0][0] - other[0]) <= 1e-6 * abs(res[0][0]), (res[0][0], other[0])
        d = (res[0][1] - other[1]).abs().max().item()
        assert d <= 2e-2 * res[0][1].abs().max().item(), d     # (backbone gradients carry the run-to-run noise of DESIGN.md 4.20)


# ---- training-time augmentation (lib/joint/dataset.py:653-690, utils/utils_fn.py:28-142) -------------
def test_augment_oracle_invariants():
    """The numpy restatement: identity parameters change nothing except that votes point to the instance's POINT box
    centre (dataset.py:676-679); a pure flip / translation moves points, boxes and votes consistently; the rotated
    aligned box follows model_util_scannet.py's enclosing-box rule (lengths never shrink below the projection)."""
    from oracle import augment as oa
    synth = importlib.import_module("3dvlp_amd.synth")
    ip = importlib.import_module("3dvlp_amd.input_pipeline")
    b = synth.make_batch(0, 2, 4096, 2, instances=True)
    mean = synth.mean_size_arr()
    ident = np.stack([oa.identity_params()] * 2)
    o = oa.augment_batch(b, ident, mean, -1)
    assert np.array_equal(o["point_clouds"], b["point_clouds"]) and np.allclose(o["center_label"], b["center_label"])
    assert np.allclose(o["size_residual_label"], b["size_residual_label"], atol=1e-6)
    assert (o["vote_label_mask"] == b["vote_label_mask"]).all()
    x = b["point_clouds"][0, :, :3]
    for i in range(synth.NUM_BOXES):
        ind = b["instance_labels"][0] == i
        c = 0.5 * (x[ind].min(0) + x[ind].max(0))
        assert np.allclose(o["vote_label"][0][ind][:, :3], c - x[ind], atol=1e-6)
        assert np.allclose(o["vote_label"][0][ind][:, 3:6], o["vote_label"][0][ind][:, :3])
    p = ident.copy()
    p[:, 0] = 1.0            # flip x
    p[:, 8:11] = [0.25, -0.1, 0.05]
    f = oa.augment_batch(b, p, mean, -1)
    assert np.allclose(f["point_clouds"][..., 0], -b["point_clouds"][..., 0] + 0.25, atol=1e-6)
    assert np.allclose(f["center_label"][:, :12, 0], -b["center_label"][:, :12, 0] + 0.25, atol=1e-6)
    assert np.allclose(f["vote_label"][..., 0], -o["vote_label"][..., 0], atol=1e-5)       # votes are translation invariant
    assert np.allclose(f["ref_center_label_list"][..., 1], b["ref_center_label_list"][..., 1] - 0.1, atol=1e-6)
    # the host-side draw of the product is the oracle's draw, call for call
    assert np.allclose(ip.draw_augment_params(np.random.default_rng(3), 1)[0], oa.draw_params(np.random.default_rng(3)), atol=1e-7)
    box = np.array([[1.0, 2.0, 0.5, 2.0, 1.0, 0.6]])
    r = oa.rotate_aligned_boxes_along_axis(box, oa.rotz(0.3), "z")
    assert r[0, 3] >= 2.0 * np.cos(0.3) - 1e-9 and r[0, 5] == 0.6


def test_augment_absent_gt_rows_follow_the_reference_statements():
    """the padded (absent) GT rows.  dataset.py:631 target_bboxes = zeros((MAX_NUM_OBJ, 6)); :649-650 rows
    [0:num_bbox] filled; flip / rotate / scale leave zero rows zero; utils_fn.py:137-139 `bbox[:, :3] += factor` moves ALL
    rows; dataset.py:823 exports target_bboxes[:, 0:3] unmasked; :688 size_residuals only for [0:num_bbox].  Restated here
    statement by statement for the absent rows and compared with the oracle's batch function; and scale_augment multiplies
    point_cloud[:, 3] (utils_fn.py:119-120), whatever that column holds."""
    from oracle import augment as oa
    synth = importlib.import_module("3dvlp_amd.synth")
    ip = importlib.import_module("3dvlp_amd.input_pipeline")
    b = synth.make_batch(2, 2, 2048, 2, instances=True)
    mean = synth.mean_size_arr()
    params = ip.draw_augment_params(np.random.default_rng(5), 2).astype(np.float64)
    params[0, 0] = params[1, 1] = 1.0
    o = oa.augment_batch(b, params, mean, 3)
    for i in range(2):
        nb = int(b["box_label_mask"][i].sum())
        M = b["center_label"].shape[1]
        assert 0 < nb < M
        target_bboxes = np.zeros((M, 6))                                        # dataset.py:631
        fx, fy = params[i, 0], params[i, 1]
        if fx:
            target_bboxes[:, 0] = -1 * target_bboxes[:, 0]                      # utils_fn.py:33
        if fy:
            target_bboxes[:, 1] = -1 * target_bboxes[:, 1]                      # :38
        scale = np.diag(params[i, 5:8])
        target_bboxes[:, 0:3] = np.dot(target_bboxes[:, 0:3], scale)            # :121 (rotations of a zero box: a zero box)
        target_bboxes[:, 3:6] = np.dot(target_bboxes[:, 3:6], scale)            # :122
        target_bboxes[:, :3] += list(params[i, 8:11])                           # :137-139
        assert np.allclose(o["center_label"][i, nb:], target_bboxes[nb:, :3].astype(np.float32), atol=1e-7)
        assert np.abs(o["center_label"][i, nb:]).max() > 1e-3                   # really the translation, not the origin
        assert (o["size_residual_label"][i, nb:] == 0).all()
        # the column scale_augment touches: column 3 (first feature channel here), the true height column stays
        col3 = b["point_clouds"][i, :, 3].astype(np.float64) * params[i, 7]
        assert np.allclose(o["point_clouds"][i, :, 3], col3, atol=1e-6)
        assert np.array_equal(o["point_clouds"][i, :, -1], b["point_clouds"][i, :, -1])


@pytest.mark.gpu
@pytest.mark.parametrize("N", [4096, 40000])
def test_augment_on_device_equals_oracle(N):
    """csrc/augment.hip (points + per-instance point boxes, votes, GT boxes) and the label derivation of
    input_pipeline.augment_on_device against the numpy restatement of the reference's loader code, drawn parameters shared."""
    from oracle import augment as oa
    ip = importlib.import_module("3dvlp_amd.input_pipeline")
    synth = importlib.import_module("3dvlp_amd.synth")
    B = 3
    host = synth.make_batch(4, B, N, 4, instances=True)
    params = ip.draw_augment_params(np.random.default_rng(11), B)
    params[0, 0], params[1, 1], params[2, :2] = 1.0, 1.0, 0.0     # every flip branch taken at least once
    hc = -1 if N == 4096 else 3    # the true height column / the column the reference as shipped scales (utils_fn.py:119-120)
    want = oa.augment_batch(host, params.astype(np.float64), synth.mean_size_arr(), hc)
    dev = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
    got = ip.augment_on_device(dev, params, height_col=hc)
    nb = int(host["box_label_mask"][0].sum())
    assert np.abs(want["center_label"][0, nb:]).max() > 1e-3   # absent GT rows carry the translation (dataset.py:823)
    for k, tol in (("point_clouds", 2e-6), ("center_label", 2e-6), ("size_residual_label", 5e-6), ("ref_center_label_list", 2e-6),
                   ("ref_size_residual_label_list", 5e-6), ("vote_label", 5e-6)):
        np.testing.assert_allclose(got[k].cpu().numpy(), want[k], rtol=0, atol=tol * 10, err_msg=k)
    assert got["vote_label_mask"].dtype == dev["vote_label_mask"].dtype   # int64 like the loader's array
    assert np.array_equal(got["vote_label_mask"].cpu().numpy(), want["vote_label_mask"])
    assert np.abs(got["point_clouds"].cpu().numpy()[..., :3] - host["point_clouds"][..., :3]).max() > 0.05  # it did something


@pytest.mark.gpu
def test_prefetcher_with_augmentation_feeds_the_step():
    """prepare = augmentation + prepare_batch on the copy stream; the step trains on the augmented batches (finite loss,
    the loader-split cloud is the AUGMENTED one)."""
    ip = importlib.import_module("3dvlp_amd.input_pipeline")
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    synth = importlib.import_module("3dvlp_amd.synth")
    rng = np.random.default_rng(0)
    host = [synth.make_batch(2 * i, 2, 8192, 2, instances=True) for i in range(3)]

    def prepare(d):
        return gs.prepare_batch(ip.augment_on_device(d, ip.draw_augment_params(rng, 2), height_col=-1))

    pf = ip.Prefetcher(iter(host), device="cuda", prepare=prepare)
    step = gs.GroundingStep(torch.device("cuda:0"), sa_dtype=torch.bfloat16)
    for i in range(3):
        d = pf.next()
        assert torch.equal(d["k/xyz"], d["point_clouds"][..., :3])
        assert not np.allclose(d["point_clouds"][..., :3].cpu().numpy(), host[i]["point_clouds"][..., :3], atol=1e-3)
        loss = step.run(d)
        assert torch.isfinite(loss)
