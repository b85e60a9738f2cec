"""The numpy part of the oracle against fixtures produced by the reference's own Python
(tests/golden/make_golden.py).  CPU only."""
import numpy as np

from oracle import oracle as orc


def test_nn_distance_matches_reference(golden):
    g = golden("nn_distance")
    for seed in range(4):
        pc1, pc2 = g[f"{seed}/pc1"], g[f"{seed}/pc2"]
        for tag, kw in (("l2", {}), ("l1", {"l1": True}), ("huber", {"l1smooth": True, "delta": 0.5})):
            d1, i1, d2, i2 = orc.nn_distance(pc1, pc2, **kw)
            np.testing.assert_allclose(d1, g[f"{seed}/{tag}/dist1"], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(d2, g[f"{seed}/{tag}/dist2"], rtol=1e-6, atol=1e-7)
            assert i1.dtype == np.int64 and (i1 == g[f"{seed}/{tag}/idx1"]).all()
            assert (i2 == g[f"{seed}/{tag}/idx2"]).all()


def test_nn_distance_demo_numpy_loop(golden):
    """utils/nn_distance.py:95-126 — the reference's only self-checking demo."""
    g = golden("nn_distance")
    pc1, pc2 = g["0/pc1"].astype(np.float64), g["0/pc2"].astype(np.float64)
    dist = ((pc1[0, :, None, :] - pc2[0, None, :, :]) ** 2).sum(-1)
    np.testing.assert_allclose(g["0/l2/dist1"][0], dist.min(1), rtol=1e-5)
    assert (g["0/l2/idx1"][0] == dist.argmin(1)).all()


def test_mha_variants(golden):
    g = golden("attention")
    W = g.weights()
    q, k = g["in/q"], g["in/k"]
    out, att = orc.multi_head_attention(W, q, k, k, 4)
    np.testing.assert_allclose(out, g["out/cross"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(att, g["out/cross_att"], rtol=1e-4, atol=1e-6)
    out, _ = orc.multi_head_attention(W, q, q, q, 4)
    np.testing.assert_allclose(out, g["out/self"], rtol=1e-4, atol=2e-5)
    out, att = orc.multi_head_attention(W, q, k, k, 4, attention_weights=g["in/bias"], way="add")
    np.testing.assert_allclose(out, g["out/add"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(att, g["out/add_att"], rtol=1e-4, atol=1e-6)
    out, _ = orc.multi_head_attention(W, q, k, k, 4, attention_weights=g["in/wts"], way="mul")
    np.testing.assert_allclose(out, g["out/mul"], rtol=1e-4, atol=2e-5)
    out, _ = orc.multi_head_attention(W, q, k, k, 4, attention_mask=g["in/mask"])
    np.testing.assert_allclose(out, g["out/mask"], rtol=1e-4, atol=2e-5)
    out, _ = orc.scaled_dot_product_attention(orc._sub(W, "attention."), q, k, k, 4)
    np.testing.assert_allclose(out, g["out/sdpa"], rtol=1e-4, atol=2e-5)


def test_decoder_layers(golden):
    g = golden("decoder_layer")
    W = g.weights()
    x, k = g["in/q"], g["in/k"]
    for i in range(2):
        x = orc.cross_attention_decoder_layer(orc._sub(W, f"{i}."), x, k, k)
        np.testing.assert_allclose(x, g[f"out/layer{i}"], rtol=1e-4, atol=3e-5)


def _mlp_layers(W, n):
    return [dict(w=W[f"layer{i}.conv.weight"][:, :, 0, 0], gamma=W[f"layer{i}.bn.bn.weight"],
                 beta=W[f"layer{i}.bn.bn.bias"], mean=W[f"layer{i}.bn.bn.running_mean"],
                 var=W[f"layer{i}.bn.bn.running_var"]) for i in range(n)]


def test_shared_mlp_train_and_eval(golden):
    g = golden("shared_mlp")
    layers = _mlp_layers(g.weights(), 3)
    x = g["in/x"]
    np.testing.assert_allclose(orc.shared_mlp(x, layers, training=False), g["out/eval"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(orc.shared_mlp(x, layers, training=True), g["out/train"], rtol=1e-4, atol=3e-5)
    # the state_dict key contract (SURVEY.md §5 checkpoint row)
    assert "layer0.conv.weight" in g["meta/keys"] and "layer0.bn.bn.running_var" in g["meta/keys"]


def test_axis_aligned_iou_matches_box_util(golden):
    g = golden("boxes")
    iou = orc.box3d_iou_axis_aligned(g["c1"], g["s1"], g["c2"], g["s2"])
    np.testing.assert_allclose(iou, g["iou"], rtol=2e-5, atol=1e-7)


def test_nce_loss_occ_row_has_zero_transpose_term():
    """SURVEY.md §7 hard part 4: for (1,P) logits loss_t == 0, so OCC == loss_v / 2."""
    rng = np.random.default_rng(0)
    logits = rng.normal(size=(1, 17))
    target = (rng.random((1, 17)) > 0.5).astype(np.float64)
    assert abs(orc.nce_loss(logits, target) - orc.soft_cross_entropy(logits, target) / 2) < 1e-12


def test_contract_vectors_file_is_what_the_oracle_returns():
    """tests/golden/contract_vectors.npz holds the oracle's indices under the three fp32 evaluation orders
    (tests/golden/make_contract_vectors.py); the orders disagree on the lattice cases, so the file can tell a build's mode."""
    import os
    V = np.load(os.path.join(os.path.dirname(__file__), "golden", "contract_vectors.npz"))
    names = sorted({k.split("/")[0] for k in V.files})
    assert len(names) == 11
    differ = {0: 0, 1: 0}
    for name in names:
        for mode in (0, 1, 2):
            if name.startswith("fps"):
                got = orc.furthest_point_sampling(V[name + "/xyz"], int(V[name + "/npoint"]), contract=mode)
            elif name.startswith("bq"):
                got = orc.ball_query(V[name + "/new_xyz"], V[name + "/xyz"], float(V[name + "/radius"]), int(V[name + "/nsample"]),
                                     contract=mode)
            else:
                got = orc.three_nn(V[name + "/unknown"], V[name + "/known"], contract=mode)[1]
            assert (got == V[f"{name}/idx_mode{mode}"]).all(), (name, mode)
        differ[0] += int((V[name + "/idx_mode0"] != V[name + "/idx_mode1"]).sum())
        differ[1] += int((V[name + "/idx_mode1"] != V[name + "/idx_mode2"]).sum())
    assert differ[0] > 100 and differ[1] > 100
