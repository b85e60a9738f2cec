"""GPU parity: the HIP kernels (through the C ABI / Python mirror) against the CPU oracle on the same
inputs.  Index outputs bit-exact; float outputs within the stated tolerance (north_star: 1e-4 rel)."""
import importlib

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pu():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return importlib.import_module("3dvlp_amd.pointnet2_utils")


@pytest.fixture(scope="module")
def ext():
    return importlib.import_module("3dvlp_amd._lib")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def scene(rng, B, N):
    return rng.uniform(0.5, 4.0, size=(B, N, 3)).astype(np.float32)


def test_fp_contract_matches_oracle_default(ext):
    assert ext.load().vlp3d_fp_contract() == orc.DEFAULT_CONTRACT
    assert ext.fp_contract() == orc.DEFAULT_CONTRACT


@pytest.fixture
def contract_mode(ext, request):
    """Runs the test body with the geometry ops in fp contract mode request.param, restoring the default afterwards."""
    prev = ext.set_fp_contract(request.param)
    yield request.param
    ext.set_fp_contract(prev)


def _lattice(rng, n_side, spacing=0.1, origin=0.35):
    g = np.stack(np.meshgrid(*[np.arange(n_side)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    return (origin + g * spacing)[rng.permutation(n_side ** 3)].astype(np.float32)


@pytest.mark.parametrize("contract_mode", [0, 1, 2], indirect=True)
def test_contract_vectors_file_all_modes(ext, contract_mode):
    """tests/golden/contract_vectors.npz (oracle indices under the three fp32 evaluation orders; the point sets are chosen
    so that the orders disagree): the HIP library in mode k reproduces the file's mode-k indices, entry for entry."""
    import os
    V = np.load(os.path.join(os.path.dirname(__file__), "golden", "contract_vectors.npz"))
    assert ext.fp_contract() == contract_mode
    for name in sorted({k.split("/")[0] for k in V.files}):
        want = V[f"{name}/idx_mode{contract_mode}"]
        if name.startswith("fps"):
            t = dev(V[name + "/xyz"])
            got = ext.furthest_point_sampling(t, int(V[name + "/npoint"])).cpu().numpy()
        elif name.startswith("bq"):
            a = (dev(V[name + "/new_xyz"]), dev(V[name + "/xyz"]), float(V[name + "/radius"]), int(V[name + "/nsample"]))
            got = ext.ball_query(*a, algorithm="scan").cpu().numpy()
        else:
            got = ext.three_nn(dev(V[name + "/unknown"]), dev(V[name + "/known"]))[1].cpu().numpy()
        assert (got == want).all(), (name, contract_mode, int((got != want).sum()))


@pytest.mark.parametrize("contract_mode", [0, 1, 2], indirect=True)
def test_geometry_ops_all_contract_modes_vs_oracle(ext, contract_mode):
    """Every kernel form of the index-producing ops (dense / four-wave / pruned / prefix-proof FPS, scan / grid ball query,
    three_nn) in fp contract mode k against the oracle with contract=k, on lattices whose exact and near ties the three
    evaluation orders resolve differently, plus a bench-like scene."""
    synth = importlib.import_module("3dvlp_amd.synth")
    k = contract_mode
    rng = np.random.default_rng(77 + k)
    lat_small = np.stack([_lattice(rng, 10), _lattice(rng, 10, origin=-0.45)])            # 1000 points: four-wave kernel
    lat_mid = np.stack([_lattice(rng, 16)])                                                # 4096 points: dense kernel
    lat_big = np.stack([_lattice(rng, 22, 0.1, -1.05)])                                    # 10 648 points: pruned kernel
    scene_ = np.stack([synth.make_scene(3000 + i, 20000)["xyz"] for i in range(2)])
    for xyz, m, algo in ((lat_small, 400, None), (lat_mid, 700, "dense"), (lat_big, 500, "pruned"), (lat_big, 500, "dense"),
                         (scene_, 1024, "pruned")):
        got = ext.furthest_point_sampling(dev(xyz), m, algo).cpu().numpy()
        assert (got == orc.furthest_point_sampling(xyz, m, contract=k)).all(), ("fps", xyz.shape, algo, k)
    # FPS of an FPS-ordered level: the proof kernels evaluate the same expression
    inds = orc.furthest_point_sampling(scene_, 2048, contract=k)
    lvl = np.take_along_axis(scene_, inds[..., None].astype(np.int64), 1)
    got, flag = ext.furthest_point_sampling(dev(lvl), 1024, prefix_hint=True, return_flag=True)
    assert (got.cpu().numpy() == orc.furthest_point_sampling(lvl, 1024, contract=k)).all()
    for new_xyz, xyz, r, ns in ((lat_mid[:, :200].copy(), lat_mid, 0.3, 16), (lat_big[:, :300].copy(), lat_big, 0.2, 32),
                                (lvl[:, :512].copy(), scene_, 0.2, 64)):
        want = orc.ball_query(new_xyz, xyz, r, ns, contract=k)
        for algo in ("scan", "grid"):
            got = ext.ball_query(dev(new_xyz), dev(xyz), r, ns, algorithm=algo).cpu().numpy()
            assert (got == want).all(), ("ball_query", algo, xyz.shape, k)
    known = lat_small[:1]
    unknown = (known[:, :500] + np.float32(0.05)).astype(np.float32)
    d2, idx = ext.three_nn(dev(unknown), dev(known))
    wd2, widx = orc.three_nn(unknown, known, contract=k)
    assert (idx.cpu().numpy() == widx).all() and (d2.cpu().numpy() == wd2).all()


@pytest.mark.parametrize("B,N,m", [(2, 1, 1), (2, 3, 3), (3, 64, 64), (2, 700, 128), (2, 1024, 512), (2, 2048, 1024),
                                   (2, 4096, 512), (1, 9000, 300), (2, 20000, 64), (1, 40000, 256),
                                   (1, 45000, 128)])
def test_fps_random(pu, B, N, m):
    rng = np.random.default_rng(N * 7 + m)
    xyz = scene(rng, B, N)
    got = pu.furthest_point_sample(dev(xyz), m).cpu().numpy()
    assert got.dtype == np.int32
    assert (got == orc.furthest_point_sampling(xyz, m)).all()


@pytest.mark.parametrize("N", [300, 1500, 5000, 36000])
def test_fps_ties_and_skip_rule(pu, N):
    """Coordinates on a coarse grid -> massive exact ties; plus points inside the |p|^2<=1e-3 skip ball."""
    rng = np.random.default_rng(N)
    xyz = rng.integers(1, 4, size=(3, N, 3)).astype(np.float32)
    xyz[:, 5] = 0.01
    xyz[1, 0] = 0.0  # even the start point may be a "skipped" one (it is still read as `old`)
    xyz[2, -1] = np.float32(np.sqrt(1e-3 / 3))  # |p|^2 right at the threshold
    m = min(N, 200)
    got = pu.furthest_point_sample(dev(xyz), m).cpu().numpy()
    assert (got == orc.furthest_point_sampling(xyz, m)).all()


@pytest.mark.parametrize("N", [1, 2, 63, 100, 255, 256, 257, 511, 512, 513, 600, 1023, 1024, 1025, 2047, 2048, 2049])
def test_fps_small_sets_ties_and_skip_rule(pu, N):
    """The four-wave kernel for N <= 2048 (csrc/fps.hip fps_small_kernel: 256 threads = half the reference's block at
    512 <= N <= 2048, lanes walk their even slots first; a whole block below) on massive exact ties, every point sampled
    (m = N), a skipped start point and a point exactly at the skip threshold — around every size where the slot count or the
    reference's block size changes."""
    rng = np.random.default_rng(1000 + N)
    xyz = rng.integers(1, 4, size=(3, N, 3)).astype(np.float32)  # 27 distinct positions: ties at every step
    if N > 5:
        xyz[:, 5] = 0.01
    xyz[1, 0] = 0.0
    xyz[2, -1] = np.float32(np.sqrt(1e-3 / 3))
    for m in sorted({1, min(N, 7), max(1, N // 2), N}):
        got = pu.furthest_point_sample(dev(xyz), m).cpu().numpy()
        assert (got == orc.furthest_point_sampling(xyz, m)).all(), (N, m)
    smooth = rng.normal(0, 1, size=(2, N, 3)).astype(np.float32) + 3
    got = pu.furthest_point_sample(dev(smooth), N).cpu().numpy()
    assert (got == orc.furthest_point_sampling(smooth, N)).all()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fps_small_sets_fuzz_vs_oracle(ext, seed):
    """Random N in 1..2600 (the four-wave kernel up to 2048, the dense kernel above), random m in 1..N, uniform / lattice
    (exact ties) / duplicated / planar / skip-ball point sets: same indices as the C oracle, 40 cases per seed (400 more were
    run once while the kernel was written: no mismatch)."""
    rng = np.random.default_rng(seed)
    for case in range(40):
        B = int(rng.integers(1, 4))
        N = int(rng.integers(1, 2600))
        m = int(rng.integers(1, N + 1))
        kind = rng.choice(["uniform", "lattice", "dups", "plane", "skip"])
        if kind == "uniform":
            p = rng.uniform(-3, 3, (B, N, 3))
        elif kind == "lattice":
            p = rng.integers(0, 4, (B, N, 3)).astype(np.float64) * 0.5 + 0.5
        elif kind == "dups":
            base = rng.uniform(-2, 2, (B, N // 8 + 1, 3))
            p = base[:, rng.integers(0, N // 8 + 1, N)]
        elif kind == "plane":
            p = rng.uniform(-3, 3, (B, N, 3))
            p[..., 2] = 1.0
        else:
            p = rng.uniform(-1, 1, (B, N, 3))
            p[:, rng.integers(0, N, N // 10 + 1)] *= 0.01
        xyz = p.astype(np.float32)
        got = ext.furthest_point_sampling(dev(xyz), m, "dense").cpu().numpy()
        assert (got == orc.furthest_point_sampling(xyz, m)).all(), (case, kind, B, N, m)


@pytest.mark.parametrize("B,N,m", [(2, 9000, 300), (2, 20000, 512), (1, 40000, 700), (1, 65536, 64), (3, 5000, 5000),
                                   (2, 80000, 300), (1, 131072, 96), (1, 65537, 128)])
def test_fps_pruned_equals_dense_and_oracle(ext, B, N, m):
    """Distance-bound pruning must not change a single index (random surfaces-like data)."""
    synth = importlib.import_module("3dvlp_amd.synth")
    xyz = np.stack([synth.make_scene(2000 + i, N)["xyz"] for i in range(B)])
    t = dev(xyz)
    pruned = ext.furthest_point_sampling(t, m, "pruned").cpu().numpy()
    dense = ext.furthest_point_sampling(t, m, "dense").cpu().numpy()
    assert (pruned == dense).all()
    assert (pruned == orc.furthest_point_sampling(xyz, m)).all()


@pytest.mark.parametrize("N", [1500, 9000, 36000, 70000])
def test_fps_pruned_ties_and_skip_rule(ext, N):
    rng = np.random.default_rng(N + 1)
    xyz = rng.integers(1, 4, size=(3, N, 3)).astype(np.float32)  # massive exact ties
    xyz[:, 5] = 0.01
    xyz[1, 0] = 0.0
    xyz[2, -1] = np.float32(np.sqrt(1e-3 / 3))
    m = min(N, 300)
    got = ext.furthest_point_sampling(dev(xyz), m, "pruned").cpu().numpy()
    assert (got == orc.furthest_point_sampling(xyz, m)).all()
    skipped = np.full((2, 200, 3), 0.001, np.float32)
    assert (ext.furthest_point_sampling(dev(skipped), 10, "pruned").cpu().numpy() == 0).all()


@pytest.mark.parametrize("kind,N,m", [("lattice", 12345, 700), ("dups", 16384, 512), ("plane", 20000, 256),
                                      ("line", 9000, 300), ("skip", 40000, 400)])
def test_fps_pruned_degenerate_sets_equal_dense_and_oracle(ext, kind, N, m):
    """Point sets built to stress the lazy tie resolution and the bounding-box test of the pruned kernel (duplicates,
    lattices with exact distance ties, flat / collinear boxes, many points inside the skip ball); tools/fuzz_fps.py
    runs the same generators at scale."""
    rng = np.random.default_rng(N + m)
    B = 2
    if kind == "lattice":
        p = rng.integers(0, 12, (B, N, 3)).astype(np.float64) * 0.25 + 0.5
    elif kind == "dups":
        base = rng.uniform(-2, 2, (B, N // 8 + 1, 3))
        p = base[:, rng.integers(0, N // 8 + 1, N)]
    elif kind == "plane":
        p = rng.uniform(-3, 3, (B, N, 3))
        p[..., 2] = 1.0
    elif kind == "line":
        t = rng.uniform(-5, 5, (B, N, 1))
        p = np.concatenate([t, 0.5 * t + 1, np.full_like(t, 0.3)], -1)
    else:
        p = rng.uniform(-1, 1, (B, N, 3))
        p[:, rng.integers(0, N, N // 10)] *= 0.01
    xyz = p.astype(np.float32)
    pruned = ext.furthest_point_sampling(dev(xyz), m, "pruned").cpu().numpy()
    dense = ext.furthest_point_sampling(dev(xyz), m, "dense").cpu().numpy()
    assert (pruned == dense).all()
    assert (pruned == orc.furthest_point_sampling(xyz, m)).all()


@pytest.mark.parametrize("B,N,n1,m", [(2, 6000, 2048, 1024), (3, 3000, 1024, 512), (2, 1500, 512, 256), (1, 900, 300, 300),
                                      (2, 40000, 2048, 1),
                                      # m = 2048: the largest sample count of the eight-lane proof kernels; 2049 / 2500: the
                                      # one-lane kernels above it; 33 / 31: one workgroup of 32 points, partly filled
                                      (1, 9000, 4096, 2048), (1, 9000, 4096, 2049), (2, 8000, 3000, 2500), (2, 700, 33, 33),
                                      (2, 700, 64, 31)])
def test_fps_prefix_hint_proves_arange_on_fps_ordered_input(ext, B, N, n1, m):
    """sa2..sa4 sample from the previous level's samples, stored in sampling order: the parallel proof must succeed
    (flag 0) and the result is 0..m-1 == the sequential kernel == the oracle."""
    synth = importlib.import_module("3dvlp_amd.synth")
    xyz = np.stack([synth.make_scene(3100 + i, N)["xyz"] for i in range(B)])
    t = dev(xyz)
    first = ext.furthest_point_sampling(t, n1).long()
    level = torch.gather(t, 1, first[..., None].expand(-1, -1, 3)).contiguous()
    got, flag = ext.furthest_point_sampling(level, m, prefix_hint=True, return_flag=True)
    assert int(flag.item()) == 0
    assert (got.cpu().numpy() == np.arange(m)[None]).all()
    assert torch.equal(got, ext.furthest_point_sampling(level, m, "dense"))
    assert (got.cpu().numpy() == orc.furthest_point_sampling(level.cpu().numpy(), m)).all()


@pytest.mark.parametrize("kind", ["random", "swapped", "tie", "skip", "dup"])
def test_fps_prefix_hint_falls_back_exactly(ext, kind):
    """A wrong hint, an exact tie at some step, a point inside the skip ball or a duplicate: the proof fails (flag 1)
    and the sequential kernel runs — same indices as without the hint and as the oracle."""
    rng = np.random.default_rng(17)
    B, N, m = 2, 1024, 256
    base = dev(scene(rng, B, 5000))
    first = ext.furthest_point_sampling(base, N).long()
    level = torch.gather(base, 1, first[..., None].expand(-1, -1, 3)).contiguous().cpu().numpy()
    if kind == "random":
        level = scene(rng, B, N)
    elif kind == "swapped":
        level[1, [40, 41]] = level[1, [41, 40]]
    elif kind == "tie":      # points 0 and 1 then two candidates at exactly the same distance to both
        level[0, 0] = (1, 1, 1)
        level[0, 1] = (3, 1, 1)
        level[0, 2] = (2, 2.5, 1)
        level[0, 3] = (2, -0.5, 1)
        level[0, 4:] = level[0, 4:] * 0.01 + np.float32(1.5)
    elif kind == "skip":
        level[1, 700] = (0.01, 0.01, 0.01)
    else:
        level[0, 300] = level[0, 7]   # a duplicate of sample 7 ties with it at step 7
    t = dev(level)
    got, flag = ext.furthest_point_sampling(t, m, prefix_hint=True, return_flag=True)
    want = orc.furthest_point_sampling(level, m)
    assert (got.cpu().numpy() == want).all()
    assert torch.equal(got, ext.furthest_point_sampling(t, m, "dense"))
    assert int(flag.item()) == 1


def test_fps_prefix_hint_in_graph_replays_with_new_points(ext):
    """The flag is rewritten by every replay: ordered input -> 0..m-1, then unordered input -> the sequential result."""
    rng = np.random.default_rng(5)
    B, N, m = 2, 2048, 1024
    base = dev(scene(rng, B, 9000))
    first = ext.furthest_point_sampling(base, N).long()
    ordered = torch.gather(base, 1, first[..., None].expand(-1, -1, 3)).contiguous()
    unordered = dev(scene(rng, B, N))
    buf = ordered.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ext.furthest_point_sampling(buf, m, prefix_hint=True)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = ext.furthest_point_sampling(buf, m, prefix_hint=True)
    for src in (ordered, unordered, ordered, unordered):
        buf.copy_(src)
        g.replay()
        assert torch.equal(out, ext.furthest_point_sampling(src, m, "dense"))
    assert not torch.equal(ext.furthest_point_sampling(unordered, m, "dense")[0].cpu(), torch.arange(m, dtype=torch.int32))


def test_fps_all_points_skipped(pu):
    xyz = np.full((2, 100, 3), 0.001, np.float32)
    got = pu.furthest_point_sample(dev(xyz), 10).cpu().numpy()
    assert (got == 0).all() and (orc.furthest_point_sampling(xyz, 10) == 0).all()


def test_fps_full_size_sa1(pu):
    """BASELINE config 2 shape: 8 scenes x 40 000 points -> 2048."""
    bench = importlib.import_module("3dvlp_amd.synth")
    xyz = np.stack([bench.make_scene(1000 + i, 40000)["xyz"] for i in range(8)])
    got = pu.furthest_point_sample(dev(xyz), 2048).cpu().numpy()
    ref = orc.furthest_point_sampling(xyz, 2048)
    assert (got == ref).all()
    # the reference relies on this (backbone_module.py:108): FPS of the FPS-ordered set is arange
    sub = np.stack([xyz[b, got[b]] for b in range(8)])
    got2 = pu.furthest_point_sample(dev(sub), 1024).cpu().numpy()
    assert (got2 == orc.furthest_point_sampling(sub, 1024)).all()


@pytest.mark.parametrize("B,N,M,r,ns", [(2, 900, 50, 0.6, 16), (1, 70, 33, 0.9, 64), (3, 5000, 300, 0.3, 32),
                                        (2, 64, 1, 10.0, 8), (2, 1, 5, 1.0, 4), (8, 2048, 1024, 0.4, 32)])
def test_ball_query(ext, B, N, M, r, ns):
    rng = np.random.default_rng(N + M)
    xyz = scene(rng, B, N)
    new_xyz = scene(rng, B, M)
    new_xyz[:, : min(M, N)] = xyz[:, : min(M, N)]
    if M > 7:
        new_xyz[:, 7] = 100.0  # empty ball
    got = ext.ball_query(dev(new_xyz), dev(xyz), r, ns).cpu().numpy()
    assert (got == orc.ball_query(new_xyz, xyz, r, ns)).all()


def test_ball_query_full_size_sa1(pu, ext):
    synth = importlib.import_module("3dvlp_amd.synth")
    xyz = np.stack([synth.make_scene(1000 + i, 40000)["xyz"] for i in range(8)])
    inds = orc.furthest_point_sampling(xyz[:, :], 2048)
    new_xyz = np.stack([xyz[b, inds[b]] for b in range(8)])
    got = pu.ball_query(0.2, 64, dev(xyz), dev(new_xyz)).cpu().numpy()
    ref = orc.ball_query(new_xyz, xyz, 0.2, 64)
    assert (got == ref).all()
    # size-independent properties: ascending up to the hit count, padded with the first hit
    assert (np.diff(got, axis=2) >= 0).sum() > 0 and (got[:, :, 0] <= got[:, :, 1]).all()


@pytest.mark.parametrize("B,n,m", [(2, 300, 77), (8, 1024, 512), (8, 512, 256), (2, 50, 2), (1, 3000, 2500)])
def test_three_nn(pu, ext, B, n, m):
    rng = np.random.default_rng(n + m)
    unknown, known = scene(rng, B, n), scene(rng, B, m)
    known[:, : min(3, m)] = unknown[:, : min(3, m)]  # zero distances and near ties
    d2, idx = ext.three_nn(dev(unknown), dev(known))
    rd2, ridx = orc.three_nn(unknown, known)
    assert (idx.cpu().numpy() == ridx).all()
    assert (d2.cpu().numpy() == rd2).all()  # same fp32 expression -> bit-identical distances
    dist, _ = pu.three_nn(dev(unknown), dev(known))
    np.testing.assert_allclose(dist.cpu().numpy(), np.sqrt(rd2), rtol=1e-6)


@pytest.mark.parametrize("n,m", [(257, 1), (300, 2), (64, 3), (1000, 4), (513, 5), (2048, 7), (700, 1024), (1024, 1025), (100, 2050)])
def test_three_nn_exact_ties_and_tiny_known_sets(ext, n, m):
    """Coordinates on a 3 x 3 x 3 lattice: nearly every distance is tied many times over, so the result is decided by the
    reference's strict '<' chain in index order (lowest index wins) — the order the four-lanes-per-point kernel has to
    reproduce when it merges its lanes' triples (csrc/interpolate.hip); known sets of 1 and 2 points leave +inf / index 0 in
    the unfilled slots; known counts around the 1024-point LDS tile."""
    rng = np.random.default_rng(7 * n + m)
    unknown = rng.integers(0, 3, size=(3, n, 3)).astype(np.float32)
    known = rng.integers(0, 3, size=(3, m, 3)).astype(np.float32)
    d2, idx = ext.three_nn(dev(unknown), dev(known))
    rd2, ridx = orc.three_nn(unknown, known)
    assert (idx.cpu().numpy() == ridx).all()
    got = d2.cpu().numpy()
    assert ((got == rd2) | (np.isinf(got) & np.isinf(rd2))).all()


@pytest.mark.parametrize("seed", [11, 12])
def test_three_nn_fuzz_vs_oracle(ext, seed):
    """Random unknown / known counts (1..2500), uniform clouds, lattices with exact ties and duplicated points: indices and
    squared distances of the four-lanes-per-point kernel equal the C oracle's, 30 cases per seed."""
    rng = np.random.default_rng(seed)
    for case in range(30):
        B = int(rng.integers(1, 4))
        n, m = int(rng.integers(1, 2500)), int(rng.integers(1, 2500))
        kind = rng.choice(["uniform", "lattice", "dups"])
        if kind == "uniform":
            u, k = rng.uniform(-2, 2, (B, n, 3)), rng.uniform(-2, 2, (B, m, 3))
        elif kind == "lattice":
            u, k = rng.integers(0, 4, (B, n, 3)) * 0.5, rng.integers(0, 4, (B, m, 3)) * 0.5
        else:
            base = rng.uniform(-2, 2, (B, 40, 3))
            u, k = base[:, rng.integers(0, 40, n)], base[:, rng.integers(0, 40, m)]
        u, k = u.astype(np.float32), k.astype(np.float32)
        d2, idx = ext.three_nn(dev(u), dev(k))
        rd2, ridx = orc.three_nn(u, k)
        assert (idx.cpu().numpy() == ridx).all(), (case, kind, B, n, m)
        got = d2.cpu().numpy()
        assert ((got == rd2) | (np.isinf(got) & np.isinf(rd2))).all(), (case, kind, B, n, m)


@pytest.mark.parametrize("B,N,M,S", [(8, 2048, 1024, 32), (3, 500, 77, 5), (2, 8192, 64, 16), (2, 9000, 64, 16), (1, 1, 1, 1)])
@pytest.mark.parametrize("compact", [False, True])
def test_sa_inverse_lists_exactly_the_rows_of_every_point(ext, B, N, M, S, compact):
    """vlp3d_sa_inverse (one workgroup per cloud for N <= 8192, the five-launch form above that): inv_start is the exclusive
    scan of the per-point row counts and the slice of a point holds exactly the rows that gather it (any order) — over all
    padded rows, and over the distinct rows of the compact map (whose row u of ball j reads point crow[u].x)."""
    rng = np.random.default_rng(B * 1000 + N)
    idx = rng.integers(0, N, (B, M, S)).astype(np.int32)
    idx[:, :, S // 2:] = idx[:, :, :1]  # ball-query padding: the tail repeats the first neighbour
    d = dev(idx)
    cmap = ext.sa_compact(d, N) if compact else None
    start, rows = ext.sa_inverse(d, N, cmap)
    start, rows = start.cpu().numpy().astype(np.int64), rows.cpu().numpy()
    if compact:
        rowptr, crow = cmap[0].cpu().numpy(), cmap[1].cpu().numpy()
        total = int(rowptr[-1])
        point = crow[:total, 0].astype(np.int64)
    else:
        total = B * M * S
        point = (np.arange(B)[:, None, None] * N + idx).reshape(-1).astype(np.int64)
    counts = np.bincount(point, minlength=B * N)
    assert start[0] == 0 and start[-1] == total and (np.diff(start) == counts).all()
    got = rows[:total].astype(np.int64)
    assert (np.sort(got) == np.arange(total)).all()                      # every row listed exactly once
    assert (point[got] == np.repeat(np.arange(B * N), counts)).all()     # ... in the slice of the point it gathers


def test_gather_and_group_forward_backward(pu):
    rng = np.random.default_rng(11)
    B, C, N, M, S = 3, 37, 500, 60, 9
    x = rng.normal(size=(B, C, N)).astype(np.float32)
    gi = rng.integers(0, N, size=(B, M)).astype(np.int32)
    qi = rng.integers(0, N, size=(B, M, S)).astype(np.int32)
    xt = dev(x).requires_grad_(True)
    out = pu.gather_operation(xt, dev(gi))
    assert (out.detach().cpu().numpy() == orc.gather_points(x, gi)).all()
    g = rng.normal(size=out.shape).astype(np.float32)
    out.backward(dev(g))
    np.testing.assert_allclose(xt.grad.cpu().numpy(), orc.gather_points_grad(g, gi, N), rtol=1e-5, atol=1e-5)

    xt = dev(x).requires_grad_(True)
    out = pu.grouping_operation(xt, dev(qi))
    assert (out.detach().cpu().numpy() == orc.group_points(x, qi)).all()
    g = rng.normal(size=out.shape).astype(np.float32)
    out.backward(dev(g))
    np.testing.assert_allclose(xt.grad.cpu().numpy(), orc.group_points_grad(g, qi, N), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,C,m,n", [(2, 33, 64, 300), (8, 256, 512, 1024), (1, 6, 2500, 700)])
def test_three_interpolate_forward_backward(pu, B, C, m, n):
    """(8,256,512,1024) is the FP2 shape of the path; m = 2500 takes the global-atomic fallback of the adjoint
    (the LDS-privatised kernel covers m <= 2048)."""
    rng = np.random.default_rng(12)
    feats = rng.normal(size=(B, C, m)).astype(np.float32)
    idx = rng.integers(0, m, size=(B, n, 3)).astype(np.int32)
    w = rng.random((B, n, 3)).astype(np.float32)
    ft = dev(feats).requires_grad_(True)
    out = pu.three_interpolate(ft, dev(idx), dev(w))
    assert (out.detach().cpu().numpy() == orc.three_interpolate(feats, idx, w)).all()  # same fma order
    g = rng.normal(size=out.shape).astype(np.float32)
    out.backward(dev(g))
    # true adjoint (documented deviation from the reference's interpolate.cpp:95 bug)
    np.testing.assert_allclose(ft.grad.cpu().numpy(), orc.three_interpolate_grad(g, idx, w, m), rtol=1e-5, atol=1e-5)


def test_three_interpolate_grad_as_shipped_switch(pu, monkeypatch):
    """THREE_INTERPOLATE_GRAD_AS_SHIPPED reproduces what the reference executes (interpolate.cpp:95: the forward wrapper on
    grad_out) — bit-equal to the oracle's restatement of that bug; the default stays the true adjoint."""
    rng = np.random.default_rng(13)
    B, C, m, n = 4, 64, 256, 512
    feats = rng.normal(size=(B, C, m)).astype(np.float32)
    idx = rng.integers(0, m, size=(B, n, 3)).astype(np.int32)
    w = rng.random((B, n, 3)).astype(np.float32)
    g = rng.normal(size=(B, C, n)).astype(np.float32)
    monkeypatch.setattr(pu, "THREE_INTERPOLATE_GRAD_AS_SHIPPED", True)
    ft = dev(feats).requires_grad_(True)
    pu.three_interpolate(ft, dev(idx), dev(w)).backward(dev(g))
    assert (ft.grad.cpu().numpy() == orc.three_interpolate_grad_asshipped(g, idx, w, m)).all()
    monkeypatch.setattr(pu, "THREE_INTERPOLATE_GRAD_AS_SHIPPED", False)
    ft = dev(feats).requires_grad_(True)
    pu.three_interpolate(ft, dev(idx), dev(w)).backward(dev(g))
    np.testing.assert_allclose(ft.grad.cpu().numpy(), orc.three_interpolate_grad(g, idx, w, m), rtol=1e-5, atol=1e-5)


def test_nn_distance_golden_and_hot_path_shapes(golden):
    nnd = importlib.import_module("3dvlp_amd.nn_distance")
    g = golden("nn_distance")
    for seed in range(4):
        pc1, pc2 = dev(g[f"{seed}/pc1"]), dev(g[f"{seed}/pc2"])
        for tag, kw in (("l2", {}), ("l1", {"l1": True}), ("huber", {"l1smooth": True, "delta": 0.5})):
            d1, i1, d2, i2 = nnd.nn_distance(pc1, pc2, **kw)
            np.testing.assert_allclose(d1.cpu().numpy(), g[f"{seed}/{tag}/dist1"], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(d2.cpu().numpy(), g[f"{seed}/{tag}/dist2"], rtol=1e-6, atol=1e-7)
            assert i1.dtype == torch.int64 and (i1.cpu().numpy() == g[f"{seed}/{tag}/idx1"]).all()
            assert (i2.cpu().numpy() == g[f"{seed}/{tag}/idx2"]).all()
    # loss_detection.py:66 and :92 call shapes
    rng = np.random.default_rng(5)
    for (B, N, M, kw) in ((8192, 1, 3, {"l1": True}), (8, 256, 256, {})):
        a, b = rng.normal(size=(B, N, 3)).astype(np.float32), rng.normal(size=(B, M, 3)).astype(np.float32)
        d1, i1, d2, i2 = nnd.nn_distance(dev(a), dev(b), **kw)
        r1, ri1, r2, ri2 = orc.nn_distance(a, b, **kw)
        np.testing.assert_allclose(d1.cpu().numpy(), r1, rtol=1e-6, atol=1e-7)
        assert (i1.cpu().numpy() == ri1).all() and (i2.cpu().numpy() == ri2).all()


def test_nn_distance_gradients_match_dense_torch():
    nnd = importlib.import_module("3dvlp_amd.nn_distance")
    torch.manual_seed(0)
    a = torch.randn(4, 50, 3, device="cuda", requires_grad=True)
    b = torch.randn(4, 20, 3, device="cuda", requires_grad=True)
    for kw in ({}, {"l1": True}, {"l1smooth": True, "delta": 0.3}):
        d1, _, d2, _ = nnd.nn_distance(a, b, **kw)
        ga, gb = torch.autograd.grad(d1.sum() + 2 * d2.sum(), [a, b])
        diff = a.unsqueeze(2) - b.unsqueeze(1)
        dist = nnd._pair_dist(diff, 2 if kw.get("l1smooth") else (1 if kw.get("l1") else 0), kw.get("delta", 1.0))
        ra, rb = torch.autograd.grad(dist.min(2)[0].sum() + 2 * dist.min(1)[0].sum(), [a, b])
        torch.testing.assert_close(ga, ra, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(gb, rb, rtol=1e-5, atol=1e-6)


# ---- the other BASELINE configurations' shapes (cfg3 B=32/GPU and cfg5 N=80 000 were untested) ----
def test_fps_ball_query_cfg5_80k_points(pu, ext):
    """cfg5 (ScanQA + grounding): 80 000-point scenes through SA1's FPS (40k->2048 rule: npoint 2048) and ball query."""
    synth = importlib.import_module("3dvlp_amd.synth")
    xyz = np.stack([synth.make_scene(3000 + i, 80000)["xyz"] for i in range(2)])
    t = dev(xyz)
    ref = orc.furthest_point_sampling(xyz, 2048)
    got = pu.furthest_point_sample(t, 2048)
    assert (got.cpu().numpy() == ref).all()
    assert (ext.furthest_point_sampling(t, 2048, "dense").cpu().numpy() == ref).all()
    new_xyz = np.stack([xyz[b, ref[b]] for b in range(2)])
    idx = pu.ball_query(0.2, 64, t, dev(new_xyz)).cpu().numpy()
    assert (idx == orc.ball_query(new_xyz, xyz, 0.2, 64)).all()


def test_fps_ball_query_cfg3_batch32(pu):
    """cfg3 (pre-training, 32 scenes per GPU): every scene of a 32-scene batch equals the oracle (grid / slab sizing
    by B)."""
    synth = importlib.import_module("3dvlp_amd.synth")
    xyz = np.stack([synth.make_scene(4000 + i, 40000)["xyz"] for i in range(32)])
    t = dev(xyz)
    ref = orc.furthest_point_sampling(xyz, 2048)
    assert (pu.furthest_point_sample(t, 2048).cpu().numpy() == ref).all()
    new_xyz = np.stack([xyz[b, ref[b]] for b in range(32)])
    idx = pu.ball_query(0.2, 64, t, dev(new_xyz)).cpu().numpy()
    assert (idx == orc.ball_query(new_xyz, xyz, 0.2, 64)).all()
    # SA2 shape on the sampled set
    ref2 = orc.furthest_point_sampling(new_xyz, 1024)
    assert (pu.furthest_point_sample(dev(new_xyz), 1024).cpu().numpy() == ref2).all()


def test_sa1_geometry_on_scene_with_skip_points(pu, ext):
    """SURVEY §8d's variant scene: 16 points inside the FPS skip ball |p|^2 <= 1e-3 (sampling_gpu.cu:106).  They are
    never selected (unless first), never update the running minimum, and ball query / three_nn treat them as ordinary
    points."""
    synth = importlib.import_module("3dvlp_amd.synth")
    scenes = [synth.make_scene(1000 + i, 40000, skip_points=16) for i in range(2)]
    xyz = np.stack([s["xyz"] for s in scenes])
    n_skip = ((xyz.astype(np.float64) ** 2).sum(-1) <= 1e-3).sum(1)
    assert (n_skip == 16).all()
    t = dev(xyz)
    ref = orc.furthest_point_sampling(xyz, 2048)
    for alg in ("pruned", "dense"):
        assert (ext.furthest_point_sampling(t, 2048, alg).cpu().numpy() == ref).all(), alg
    skipped = (xyz.astype(np.float64) ** 2).sum(-1) <= 1e-3
    for b in range(2):
        assert not skipped[b, ref[b, 1:]].any()
    new_xyz = np.stack([xyz[b, ref[b]] for b in range(2)])
    assert (pu.ball_query(0.2, 64, t, dev(new_xyz)).cpu().numpy() == orc.ball_query(new_xyz, xyz, 0.2, 64)).all()
    d2, i3 = ext.three_nn(dev(new_xyz[:, :1024]), dev(new_xyz[:, :512].copy()))
    r2, ri = orc.three_nn(new_xyz[:, :1024], new_xyz[:, :512].copy())
    assert (i3.cpu().numpy() == ri).all() and (d2.cpu().numpy() == r2).all()


@pytest.mark.parametrize("kind,B,N,M,r,ns", [("scene", 2, 40000, 2048, 0.2, 64), ("scene", 1, 9000, 512, 0.4, 32),
                                            ("cluster", 2, 20000, 256, 0.3, 16), ("dense", 1, 12000, 64, 0.5, 64),
                                            ("outside", 2, 10000, 128, 0.25, 8), ("empty", 1, 8192, 64, 0.01, 16)])
def test_ball_query_grid_equals_scan_and_oracle(ext, kind, B, N, M, r, ns):
    """The grid kernel returns the all-pairs kernel's rows bit for bit: ascending first-nsample order, first-hit padding,
    all-zero rows for empty balls — on scenes, on clusters denser than the hit list (fallback path), with more than
    nsample hits per ball (rank selection), with centres outside the points' bounding box and with empty balls."""
    synth = importlib.import_module("3dvlp_amd.synth")
    rng = np.random.default_rng(N + M)
    if kind == "scene":
        xyz = np.stack([synth.make_scene(5000 + i, N)["xyz"] for i in range(B)])
        new_xyz = xyz[:, rng.permutation(N)[:M]].copy()
    elif kind == "cluster":   # most points inside a few tiny blobs: thousands of hits per ball
        cen = rng.uniform(0, 4, (B, 4, 3))
        xyz = (cen[:, rng.integers(0, 4, N)] + rng.normal(0, 0.05, (B, N, 3))).astype(np.float32)
        new_xyz = xyz[:, :M].copy()
    elif kind == "dense":
        xyz = rng.uniform(0, 1.5, (B, N, 3)).astype(np.float32)
        new_xyz = xyz[:, :M].copy()
    elif kind == "outside":
        xyz = rng.uniform(0, 3, (B, N, 3)).astype(np.float32)
        new_xyz = rng.uniform(-1, 4, (B, M, 3)).astype(np.float32)
    else:
        xyz = rng.uniform(0, 5, (B, N, 3)).astype(np.float32)
        new_xyz = (xyz[:, :M] + 0.5).astype(np.float32)
    a = ext.ball_query(dev(new_xyz), dev(xyz), r, ns, "grid").cpu().numpy()
    b = ext.ball_query(dev(new_xyz), dev(xyz), r, ns, "scan").cpu().numpy()
    assert (a == b).all()
    assert (a == orc.ball_query(new_xyz, xyz, r, ns)).all()
    # the one-launch form on the pruned FPS's spatial sort of the same cloud (csrc/ball_query_sorted.hip)
    t = dev(xyz)
    _, ws = ext.furthest_point_sampling(t, 16, "pruned", return_workspace=True)
    c = ext.ball_query_sorted(dev(new_xyz), t, r, ns, ws).cpu().numpy()
    assert (c == b).all(), (kind, int((c != b).sum()))


@pytest.mark.parametrize("contract_mode", [0, 1, 2], indirect=True)
def test_ball_query_sorted_all_contract_modes_and_odd_shapes(ext, contract_mode):
    """vlp3d_ball_query_sorted == the oracle in every fp contract mode on a lattice whose radius equals lattice distances
    (hits decided by the last bit), with M not a multiple of the four centres a workgroup owns, nsample above the usual 64,
    flat clouds (zero extent on an axis) and a 70 000-point cloud (two lane-slots in the FPS, same workspace layout)."""
    synth = importlib.import_module("3dvlp_amd.synth")
    k = contract_mode
    rng = np.random.default_rng(9 + k)
    lat = np.stack([_lattice(rng, 22, 0.1, -1.05), _lattice(rng, 22, 0.1, 0.35)])
    flat = rng.uniform(0, 3, (2, 9000, 3)).astype(np.float32)
    flat[..., 2] = 1.25
    big = np.stack([synth.make_scene(7000, 70000)["xyz"]])
    for xyz, M, r, ns in ((lat, 301, 0.3, 16), (lat, 64, 0.2, 100), (flat, 130, 0.15, 32), (big, 1000, 0.2, 64)):
        new_xyz = xyz[:, rng.permutation(xyz.shape[1])[:M]].copy()
        t = dev(xyz)
        _, ws = ext.furthest_point_sampling(t, 8, "pruned", return_workspace=True)
        got = ext.ball_query_sorted(dev(new_xyz), t, r, ns, ws).cpu().numpy()
        assert (got == orc.ball_query(new_xyz, xyz, r, ns, contract=k)).all(), (xyz.shape, M, r, ns, k)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(8, 128, 1024, 256), (8, 32, 40000, 2048)])
def test_zeroing_inside_captured_graph_is_ordered(ext, shape):
    """Accumulate-into-zeroed-buffer entries stay correct when the captured graph is replayed and the output block
    held other data in between (hipMemsetAsync nodes did not guarantee that on ROCm 7.2: common.h vlp3d_zero_words)."""
    B, C, N, M = shape
    torch.manual_seed(3)
    go = torch.randn(B, C, M, device="cuda")
    idx = torch.randint(0, N, (B, M), device="cuda", dtype=torch.int32)
    ref = ext.gather_points_grad(go, idx, N).clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = ext.gather_points_grad(go, idx, N)
        res = out.clone()
        del out
        a = torch.empty(B, C, N, device="cuda")
        a.fill_(1.0)  # same size: takes the block `out` had
        del a
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        assert (res - ref).abs().max().item() < 1e-5  # atomic accumulation order varies; a missed clear is off by 1.0


@pytest.mark.gpu
def test_grid_ball_query_replayed_from_a_graph(ext):
    synth = importlib.import_module("3dvlp_amd.synth")
    B, N, M = 4, 20000, 1024
    xyz = torch.from_numpy(synth.make_batch(0, B, N, 2)["point_clouds"][..., :3].copy()).cuda().contiguous()
    new_xyz = xyz[:, ::N // M][:, :M].contiguous()
    ref = ext.ball_query(new_xyz, xyz, 0.2, 64, algorithm="scan")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = ext.ball_query(new_xyz, xyz, 0.2, 64, algorithm="grid")
        junk = torch.zeros(4 * 1024 * 1024, device="cuda") + 1.0  # lands in the freed workspace block
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
