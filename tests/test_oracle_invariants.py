"""Algebraic invariants of the C oracle for the nine _ext ops.  The reference pins
no vectors for these ops ("parity unpinned"), so these properties are what anchors the restatement."""
import numpy as np
import pytest

from oracle import oracle as orc


def _scene(rng, B, N):
    return rng.uniform(0.5, 4.0, size=(B, N, 3)).astype(np.float32)


def test_opt_n_threads():
    # include/cuda_utils.h:20-24
    assert [orc.opt_n_threads(n) for n in (1, 2, 3, 511, 512, 513, 1024, 40000)] == [1, 2, 2, 256, 512, 512, 512, 512]


@pytest.mark.parametrize("contract", [0, 1, 2])
def test_fps_basic(contract):
    rng = np.random.default_rng(1)
    xyz = _scene(rng, 3, 700)
    idx = orc.furthest_point_sampling(xyz, 128, contract)
    assert idx.dtype == np.int32 and idx.shape == (3, 128)
    assert (idx[:, 0] == 0).all()
    for b in range(3):
        assert len(set(idx[b].tolist())) == 128
    # brute-force greedy FPS in float64 agrees on tie-free data
    for b in range(3):
        d = np.full(700, 1e10)
        cur = 0
        for j in range(1, 128):
            d = np.minimum(d, ((xyz[b].astype(np.float64) - xyz[b, cur]) ** 2).sum(1))
            cur = int(d.argmax())
            assert cur == idx[b, j]


def test_fps_of_fps_prefix_is_arange():
    """backbone_module.py:108,114,119 relies on this."""
    rng = np.random.default_rng(2)
    xyz = _scene(rng, 2, 2000)
    idx = orc.furthest_point_sampling(xyz, 256)
    sub = np.stack([xyz[b, idx[b]] for b in range(2)])
    idx2 = orc.furthest_point_sampling(sub, 128)
    assert (idx2 == np.arange(128)[None]).all()


def test_fps_skip_rule_and_ties():
    """sampling_gpu.cu:106 skip rule and the :64-70 tie order (min over (bitrev(k mod P), k))."""
    rng = np.random.default_rng(3)
    N = 1500
    P = orc.opt_n_threads(N)
    bits = P.bit_length() - 1
    # heavily tied data: coordinates on a coarse integer grid
    xyz = rng.integers(1, 4, size=(4, N, 3)).astype(np.float32)
    xyz[:, 5] = 0.01  # |p|^2 = 3e-4 <= 1e-3 -> never selected
    idx = orc.furthest_point_sampling(xyz, 40)
    assert not (idx == 5).any()

    def bitrev(v):
        return int(format(v, f"0{bits}b")[::-1], 2)

    for b in range(4):
        temp = np.full(N, 1e10, np.float32)
        skip = (xyz[b].astype(np.float64) ** 2).sum(1) <= 1e-3
        old = 0
        for j in range(1, 40):
            d = ((xyz[b] - xyz[b, old]) ** 2).sum(1).astype(np.float32)
            temp = np.where(skip, temp, np.minimum(temp, d))
            cand = np.where(skip, -1.0, temp)
            best = cand.max()
            ks = np.nonzero(cand == best)[0]
            old = min(ks, key=lambda k: (bitrev(int(k) % P), int(k)))
            assert old == idx[b, j]


def test_ball_query_properties():
    rng = np.random.default_rng(4)
    xyz = _scene(rng, 2, 900)
    new_xyz = xyz[:, :50].copy()
    new_xyz[:, 7] = 100.0  # empty ball -> row stays zero
    r, ns = 0.6, 16
    idx = orc.ball_query(new_xyz, xyz, r, ns)
    r2 = np.float32(r) * np.float32(r)
    for b in range(2):
        d2 = ((new_xyz[b][:, None, :].astype(np.float64) - xyz[b][None]) ** 2).sum(-1)
        for j in range(50):
            hits = np.nonzero(d2[j] < r2 - 1e-6)[0]
            row = idx[b, j]
            if len(hits) == 0:
                assert (row == 0).all()
                continue
            cnt = min(len(hits), ns)
            assert (row[:cnt] == hits[:cnt]).all()
            assert (row[cnt:] == hits[0]).all()


def test_three_nn_matches_bruteforce():
    rng = np.random.default_rng(5)
    unknown, known = _scene(rng, 2, 300), _scene(rng, 2, 77)
    dist2, idx = orc.three_nn(unknown, known)
    d = ((unknown[:, :, None, :].astype(np.float64) - known[:, None]) ** 2).sum(-1)
    order = np.argsort(d, axis=2, kind="stable")[:, :, :3]
    assert (idx == order).all()
    np.testing.assert_allclose(dist2, np.take_along_axis(d, order, 2), rtol=1e-5)
    assert (np.diff(dist2, axis=2) >= 0).all()
    # fewer than 3 known points: (float)1e40 = inf, index 0 (interpolate_gpu.cu:32-33)
    dist2, idx = orc.three_nn(unknown, known[:, :2])
    assert np.isinf(dist2[..., 2]).all() and (idx[..., 2] == 0).all()


def test_gather_group_adjoint():
    rng = np.random.default_rng(6)
    B, C, N, M, S = 2, 5, 64, 16, 4
    x = rng.normal(size=(B, C, N)).astype(np.float32)
    gi = rng.integers(0, N, size=(B, M)).astype(np.int32)
    qi = rng.integers(0, N, size=(B, M, S)).astype(np.int32)
    y1 = rng.normal(size=(B, C, M)).astype(np.float32)
    y2 = rng.normal(size=(B, C, M, S)).astype(np.float32)
    assert np.isclose((orc.gather_points(x, gi) * y1).sum(), (x * orc.gather_points_grad(y1, gi, N)).sum(), rtol=1e-4)
    assert np.isclose((orc.group_points(x, qi) * y2).sum(), (x * orc.group_points_grad(y2, qi, N)).sum(), rtol=1e-4)
    assert (orc.group_points(x, qi)[0, 3, 2, 1] == x[0, 3, qi[0, 2, 1]])


def test_three_interpolate_grad_adjoint_and_asshipped():
    rng = np.random.default_rng(7)
    B, C, m, n = 2, 6, 20, 50
    feats = rng.normal(size=(B, C, m)).astype(np.float32)
    idx = rng.integers(0, m, size=(B, n, 3)).astype(np.int32)
    w = rng.random((B, n, 3)).astype(np.float32)
    g = rng.normal(size=(B, C, n)).astype(np.float32)
    out = orc.three_interpolate(feats, idx, w)
    ref = sum(np.take_along_axis(feats, np.broadcast_to(idx[:, None, :, t], (B, C, n)), 2) * w[:, None, :, t]
              for t in range(3))
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-6)
    assert np.isclose((out * g).sum(), (feats * orc.three_interpolate_grad(g, idx, w, m)).sum(), rtol=1e-4)
    # as shipped (interpolate.cpp:95): a forward blend of grad_out with a wrong batch stride — not the adjoint
    bug = orc.three_interpolate_grad_asshipped(g, idx, w, m)
    flat_i, flat_w = idx.reshape(-1, 3), w.reshape(-1, 3)
    for b in range(B):
        ii, ww = flat_i[b * m:(b + 1) * m], flat_w[b * m:(b + 1) * m]
        exp = sum(g[b][:, ii[:, t]] * ww[None, :, t] for t in range(3))
        np.testing.assert_allclose(bug[b], exp, rtol=1e-5, atol=1e-6)
