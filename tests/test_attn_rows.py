"""The match decoder's attention cores on bf16 ROWS (include/vlp3d.h: vlp3d_sdpa_fwd_io / _bwd_io, vlp3d_rows_chain_io,
vlp3d_linear_fwd_rows16, vlp3d_linear_wgrad_job.x_bf16) — SURVEY.md §8(d)'s bytes for a13-a15 (attention.py:63-75 between the
projections, mmattention.py:68-86): q, k, v, out cross memory once, as bf16.

The bf16-MFMA kernels round q / k / v (and a chain its input tile) to bf16 on the way in; operands that are stored as bf16 must
therefore give the SAME BITS as fp32 operands holding the same values.  That is what these tests pin, kernel by kernel; the
composed test bounds what the one real difference (delta = rowsum(dout * out) reads the stored, once-more-rounded out) costs."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu


def _mods():
    return (importlib.import_module("3dvlp_amd._lib"), importlib.import_module("3dvlp_amd.fused_attention"),
            importlib.import_module("3dvlp_amd.row_chain"), importlib.import_module("3dvlp_amd.mfma_linear"),
            importlib.import_module("3dvlp_amd.add_norm"))


def _r16(t):
    return t.bfloat16().float()


@pytest.mark.parametrize("B", [16, 20])   # 16: a batch element's workgroups share an XCD (xcd_block); 20: the plain mapping
@pytest.mark.parametrize("nk,merged", [(49, False), (256, True), (40, False), (96, True), (49, "kv16")])
def test_cores_on_bf16_rows_give_the_bits_of_the_fp32_rows(nk, merged, B):
    """forward: out rows == bf16(out of the fp32-row core), lse equal; backward with the SAME (rounded) out on both sides:
    dq / dk / dv bit-equal.  cross (q bf16, k|v fp32 merged; "kv16": k|v bf16 rows too) and self (q|k|v bf16 merged)."""
    ext = _mods()[0]
    torch.manual_seed(nk)
    H, nq, HD = 4, 256 if merged is True else 200, 128
    if merged is True:
        nq = nk
        a = _r16(torch.randn(B, nq, 3 * HD, device="cuda"))
        q, k, v = a[..., :HD], a[..., HD:2 * HD], a[..., 2 * HD:]
        a16 = a.bfloat16()
        q16, k16, v16 = a16[..., :HD], a16[..., HD:2 * HD], a16[..., 2 * HD:]
    else:
        q = _r16(torch.randn(B, nq, HD, device="cuda"))
        kv = torch.randn(B, nk, 2 * HD, device="cuda")  # fp32 k | v: rounded by the kernel, as in the fp32-row core
        if merged == "kv16":
            kv = _r16(kv)
        k, v = kv[..., :HD], kv[..., HD:]
        q16, k16, v16 = q.bfloat16(), k, v
        if merged == "kv16":
            kv16 = kv.bfloat16()
            k16, v16 = kv16[..., :HD], kv16[..., HD:]
    out, lse = ext.sdpa_fwd(q, k, v, H, None, 0, None, True)
    out16, lse16 = ext.sdpa_fwd_rows(q16, k16, v16, H, None, True)
    assert out16.dtype == torch.bfloat16
    assert torch.equal(out16, out.bfloat16())
    assert torch.equal(lse16, lse)
    dout = torch.randn(B, nq, HD, device="cuda")
    dq, dk, dv, _ = ext.sdpa_bwd(q, k, v, H, None, 0, None, out16.float(), lse, dout, False, True)
    dq16, dk16, dv16 = ext.sdpa_bwd_rows(q16, k16, v16, H, None, out16, lse16, dout)
    assert torch.equal(dq16, dq) and torch.equal(dk16, dk) and torch.equal(dv16, dv)
    # and against the unrounded out: only delta moves (relative 2^-9 per element of out)
    dq0, dk0, dv0, _ = ext.sdpa_bwd(q, k, v, H, None, 0, None, out, lse, dout, False, True)
    for g16, g0 in ((dq16, dq0), (dk16, dk0), (dv16, dv0)):
        assert float((g16 - g0).norm()) < 4e-3 * float(g0.norm())


def test_cores_on_bf16_rows_refuse_what_the_lds_kernels_do_not_cover():
    ext = _mods()[0]
    q = torch.randn(2, 64, 128, device="cuda").bfloat16()   # 2 x 4 (batch, head) pairs: below the LDS kernels' threshold in backward
    kv = torch.randn(2, 49, 256, device="cuda")
    out, lse = ext.sdpa_fwd_rows(q, kv[..., :128], kv[..., 128:], 4, None, True)
    with pytest.raises(ext.Vlp3dError):
        ext.sdpa_bwd_rows(q, kv[..., :128], kv[..., 128:], 4, None, out, lse, torch.randn(2, 64, 128, device="cuda"))
    with pytest.raises(RuntimeError):   # fp32 q with bf16 k / v: not a built combination
        ext.sdpa_fwd_rows(q.float(), kv[..., :128].bfloat16(), kv[..., 128:].bfloat16(), 4, None, False)


def test_query_projection_stored_as_bf16_rows():
    ext, _, _, ml, _ = _mods()
    torch.manual_seed(3)
    x = torch.randn(4096, 128, device="cuda", requires_grad=True)
    lin = torch.nn.Linear(128, 128).cuda()
    with ml.bf16_mma(True):
        y, xr = ml.linear(x, lin.weight, lin.bias, with_residual=True)
        assert ml.rows16_supported(x, lin.weight)
        ys, xr2, rows = ml.linear_rows16(x, lin.weight, lin.bias)
    assert rows.dtype == torch.bfloat16 and ys.dtype == torch.float32 and not rows.requires_grad
    d = (rows.float() - y.detach()).abs()
    assert float(d.max()) <= float(y.detach().abs().max()) * 2.0 ** -8   # the fp32 result rounded once
    g, gr = torch.randn_like(y), torch.randn_like(x)
    gx, gw, gb = torch.autograd.grad([y, xr], [x, lin.weight, lin.bias], [g, gr])
    gx2, gw2, gb2 = torch.autograd.grad([ys, xr2], [x, lin.weight, lin.bias], [g, gr])
    assert torch.equal(gx, gx2) and torch.equal(gw, gw2) and torch.equal(gb, gb2)


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("deferred", [False, True])
def test_chain_on_bf16_input_rows_and_bf16_last_stage(deferred, compact):
    """row_chain.run(shell, x_rows=bf16 rows, last_rows=True) == row_chain.run(the same values as fp32): every stored tile, the
    last stage's rows (= its fp32 tile rounded), the input gradient and every parameter gradient bit for bit — also with the
    stage-0 weight gradient taken by the batched kernel from the bf16 rows (deferred queue).  compact: the FFN stage keeps only
    its bf16 output (no pre-activation, compact_acts) — the same bits again: h > 0 <=> z > 0 where the dropout mask kept the
    element, and the products read h rounded to bf16 either way."""
    ext, _, rc, ml, an = _mods()
    torch.manual_seed(11)
    R, p = 64 * 21, 0.1
    mk = lambda n, k: torch.nn.Linear(k, n).cuda()
    fo, l1, l2, nx = mk(128, 128), mk(256, 128), mk(128, 256), mk(384, 128)
    n1, n2 = torch.nn.LayerNorm(128).cuda(), torch.nn.LayerNorm(128).cuda()
    params = [q for m in (fo, l1, l2, nx, n1, n2) for q in m.parameters()]
    a0, x0 = _r16(torch.randn(R, 128, device="cuda")), torch.randn(R, 128, device="cuda")
    g0, g2, g3 = torch.randn(R, 128, device="cuda"), torch.randn(R, 128, device="cuda"), torch.randn(R, 384, device="cuda")

    def once(rows):
        an._CALLS[0] = 700
        a, x = a0.clone().requires_grad_(), x0.clone().requires_grad_()
        for q in params:
            q.grad = None
        st = [rc.linear_add_norm(fo.weight, fo.bias, n1, x, p), rc.linear(l1.weight, l1.bias, "relu", p),
              rc.linear_add_norm(l2.weight, l2.bias, n2, ("tile", 1), p), rc.linear(nx.weight, nx.bias)]
        with ml.bf16_mma(True):
            assert rc.supported(a, st)
            if rows:
                shell = torch.full_like(a0, float("nan")).requires_grad_()   # never read: NaNs would surface anywhere
                t = rc.run(shell, st, True, x_rows=a0.bfloat16(), last_rows=True, compact_acts=compact)
                assert t[1].dtype == (torch.bfloat16 if compact else torch.float32)
                a = shell
            else:
                t = rc.run(a, st, True)
            loss_in = [t[0], t[2], t[3]]
            if deferred:
                with ext.deferred_slab_reduce():
                    torch.autograd.backward(loss_in, [g0, g2, g3])
            else:
                torch.autograd.backward(loss_in, [g0, g2, g3])
        torch.cuda.synchronize()
        return [u.detach() for u in t], [a.grad, x.grad] + [q.grad for q in params]

    (tf, gf), (tr, gr) = once(False), once(True)
    assert len(tr) == len(tf) + 1 and tr[-1].dtype == torch.bfloat16
    for u, c in zip(tf[:3], tr[:3]):
        assert torch.equal(u, c) if c.dtype == u.dtype else torch.equal(u.bfloat16(), c)
    assert torch.equal(tr[-1], tf[3].bfloat16())
    for u, c in zip(gf, gr):
        assert torch.equal(u, c)


def test_compact_chain_backward_on_the_unfused_entry_points():
    """The same chain with the one-launch backward switched off (row_chain.FUSED_BACKWARD = False: vlp3d_act_dropout /
    vlp3d_linear_dgrad per stage): the compact form hands the FFN stage's bf16 output to vlp3d_act_dropout in place of the
    pre-activation it no longer stores — the same gradients bit for bit."""
    ext, _, rc, ml, an = _mods()
    torch.manual_seed(13)
    R, p = 64 * 9, 0.1
    mk = lambda n, k: torch.nn.Linear(k, n).cuda()
    fo, l1, l2 = mk(128, 128), mk(256, 128), mk(128, 256)
    n1, n2 = torch.nn.LayerNorm(128).cuda(), torch.nn.LayerNorm(128).cuda()
    params = [q for m in (fo, l1, l2, n1, n2) for q in m.parameters()]
    a0, x0, g2 = torch.randn(R, 128, device="cuda"), torch.randn(R, 128, device="cuda"), torch.randn(R, 128, device="cuda")

    def once(compact):
        an._CALLS[0] = 900
        a, x = a0.clone().requires_grad_(), x0.clone().requires_grad_()
        for q in params:
            q.grad = None
        st = [rc.linear_add_norm(fo.weight, fo.bias, n1, x, p), rc.linear(l1.weight, l1.bias, "relu", p),
              rc.linear_add_norm(l2.weight, l2.bias, n2, ("tile", 1), p)]
        with ml.bf16_mma(True):
            t = rc.run(a, st, True, compact_acts=compact)
            t[2].backward(g2)
        torch.cuda.synchronize()
        return [a.grad, x.grad] + [q.grad for q in params]

    old = rc.FUSED_BACKWARD
    rc.FUSED_BACKWARD = False
    try:
        plain, compact = once(False), once(True)
    finally:
        rc.FUSED_BACKWARD = old
    for u, c in zip(plain, compact):
        assert torch.equal(u, c)


def test_match_module_decoder_on_bf16_rows_equals_the_fp32_rows():
    """MatchModule.forward (chained decoder stack, dropout off) with transformer.ATTN_BF16_ROWS on and off: the forward is the
    same up to the query projection's launch shape, the gradients differ through delta only — bounded by a fraction of what
    bf16 operands cost in the first place (the distance between the fp32-row form and the exact-fp32 MFMA form)."""
    ext, fa, rc, ml, _ = _mods()
    gr = importlib.import_module("3dvlp_amd.grounding")
    tr = importlib.import_module("3dvlp_amd.transformer")
    torch.manual_seed(5)
    B, L, K, C, T = 4, 4, 256, 128, 20
    mm = gr.MatchModule(num_proposals=K, hidden_size=C).cuda().train()
    for mod in mm.modules():
        if hasattr(mod, "fused_norm"):
            mod.fused_norm = True
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    feats = torch.randn(B, K, C, device="cuda")
    lang = torch.randn(B * L, T + 1, C, device="cuda")
    gconf, gfeat = torch.randn(B * L, K, device="cuda"), torch.randn(B * L, K, C, device="cuda")

    def run(bf, rows):
        old = tr.ATTN_BF16_ROWS
        tr.ATTN_BF16_ROWS = rows
        calls = []
        orig = ext.sdpa_fwd_rows
        ext.sdpa_fwd_rows = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            for mod in mm.modules():
                if hasattr(mod, "bf16_mma"):
                    mod.bf16_mma = bf
            mm.zero_grad()
            x = feats.clone().requires_grad_()
            dd = {"bbox_feature": x, "input_ids": torch.zeros(B, L, T + 1), "istrain": [0], "lang_fea": lang}
            with ml.bf16_mma(bf):
                dd = mm(dd)
                ((dd["cluster_ref"] * gconf).sum() + (dd["cross_box_feature"] * gfeat).sum()).backward()
            out = {"cluster_ref": dd["cluster_ref"].detach().double(), "cross_box_feature": dd["cross_box_feature"].detach().double(),
                   "d bbox_feature": x.grad.double()}
            out.update({n: p_.grad.double() for n, p_ in mm.named_parameters() if p_.grad is not None})
            return out, len(calls)
        finally:
            tr.ATTN_BF16_ROWS = old
            ext.sdpa_fwd_rows = orig

    (exact, n0), (f32, n1), (rows, n2) = run(False, False), run(True, False), run(True, True)
    assert (n0, n1, n2) == (0, 0, 3)   # cross (layer 0), self + cross (layer 1)
    scale = max(float(v.norm()) for n, v in exact.items() if "." in n)
    for n in exact:
        if n.endswith("fc_k.bias"):
            assert float(rows[n].norm()) < 1e-3 * scale
            continue
        bf_cost = float((f32[n] - exact[n]).norm())
        assert float((rows[n] - f32[n]).norm()) < 0.5 * bf_cost + 1e-6 * scale, (n, float((rows[n] - f32[n]).norm()), bf_cost)
