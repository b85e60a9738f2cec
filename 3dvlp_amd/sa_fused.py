"""Autograd wrapper of the fused grouped-MLP kernels (csrc/sa_mlp.hip) for PointnetSAModuleVotes.

forward : gather -> (GEMM, BN batch statistics) x 3 -> max over nsample through BN+ReLU, all on the matrix
          cores, nothing of shape (B,C,npoint,nsample) except the stored pre-activations Y_l.
backward: BatchNorm / ReLU / max-pool backward folded into the loaders and epilogues of the same kernels.
Same math as pointnet2_modules.py:233-267 of the reference (training-mode BatchNorm2d statistics over
B*npoint*nsample rows == batch_norm over the rows of a (rows, C) matrix).
"""
import os

import torch
from torch.autograd import Function

from . import _lib as _ext

_ext.load()

# most workgroups (= partial dW slabs) of the weight-gradient kernel.  Round 3: 512 / no slab cap -> 2048 / 32 MB: layers with a
# small dW (SA1: 16-36 KB) get up to 2048 workgroups (their tiles are staging-latency bound: 4.5 us per 32-row tile at two
# workgroups per CU), layers with a large dW (131-147 KB) fewer than before; 5.34 -> 5.30 ms per step
WGRAD_BLOCKS = int(os.environ.get("VLP3D_WGRAD_BLOCKS", 2048))
WGRAD_TILES = int(os.environ.get("VLP3D_WGRAD_TILES", 8))    # 32-row tiles a workgroup accumulates before writing its slab (round 4: 4 -> 8)


WGRAD_SLAB_MB = float(os.environ.get("VLP3D_WGRAD_SLAB_MB", 16))  # cap on the partial-dW slabs of one launch
# last layer's backward without its pre-activation (csrc/sa_last.hip, bf16 storage): "1" on, "0" the round-3 pooled loaders
SA_LAST = os.environ.get("VLP3D_SA_LAST", "1") != "0"
# padded rows (B * npoint * nsample) from which the new kernels are used (SA1: 1 048 576, SA2: 262 144, SA3: 65 536, SA4 / vote
# aggregation: 32 768): measured in the step, see backward()
SA_LAST_WGRAD_MIN_ROWS = int(os.environ.get("VLP3D_SA_LAST_WGRAD_MIN_ROWS", 500000))
SA_LAST_DGRAD_MIN_ROWS = int(os.environ.get("VLP3D_SA_LAST_DGRAD_MIN_ROWS", 65536))
SA_LAST_WBLOCKS = int(os.environ.get("VLP3D_SA_LAST_WBLOCKS", 512))  # most workgroups (= slabs) of the last layer's wgrad


def _wgrad_blocks(R, cout, K):
    """Workgroups of the weight-gradient kernel: enough resident waves to cover the memory latency of the
    staging loads (light tiles: several workgroups per CU), few enough that writing and re-reading the
    per-workgroup dW slabs (cout*K floats each) stays small against the operands themselves."""
    ntiles = R // 32
    by_slab = int(WGRAD_SLAB_MB * 2 ** 20 / (cout * K * 4))
    return max(64, min(WGRAD_BLOCKS, ntiles // WGRAD_TILES, by_slab))


def _round_up(x, m):
    return (x + m - 1) // m * m


def supported(C, mlp_out, S, R, M=None):
    """Shapes the fused kernels cover; M*S % 32 == 0 keeps every 32-row tile inside one scene."""
    return C % 4 == 0 and all(c in (64, 128, 256) for c in mlp_out) and len(mlp_out) == 3 and S <= 255 and \
        R % 32 == 0 and R < 2 ** 31 and (M is None or (M * S) % 32 == 0)


class FusedSAMLP(Function):
    """(xyz, new_xyz, idx, feat_pm, W1,g1,b1, W2,g2,b2, W3,g3,b3) -> pooled (B*M, C3) fp32."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, idx, feat_pm, radius, bns, training, use_bf16, cmap, inv, feat_c, feat_rows, want_rows,
                *params):
        W, gam, bet = params[0::3], params[1::3], params[2::3]
        # inv = (inv_start, inv_rows) of _lib.sa_inverse: backward sums the gather layer's input gradient per point through
        # this map (no atomics, csrc/sa_gather_sum.hip) — bf16 rows, feature gradients only
        ctx.inv = inv if (inv is not None and use_bf16) else None
        # cmap = (rowptr, crow) of _lib.sa_compact: the stack then runs on the DISTINCT rows of every ball (bf16 only)
        cm = (cmap[1], cmap[0], idx.shape[0] * idx.shape[1]) if (cmap is not None and use_bf16) else (None, None, 0)
        B, N, _ = xyz.shape
        _, M, S = idx.shape
        # feat_pm as BF16 rows (the loader's bf16 copy of the cloud's channels, zero padded to a multiple of 8 columns; feat_c =
        # the real channel count): read as they are by the gather layer and its weight gradient (include/vlp3d.h: bf16_io bit 1)
        # feat_rows: the SAME values as bf16 rows beside an fp32 feat_pm that carries the gradient (the previous level's pooled
        # output, vlp3d_sa_pool_rows): the kernels read the rows, autograd sees feat_pm.  want_rows: also return this level's
        # pooled output as bf16 rows for the next level.
        src = feat_rows if feat_rows is not None else feat_pm
        feat_bf = src.dtype == torch.bfloat16
        C = int(feat_c) if feat_bf else feat_pm.shape[2]
        if feat_bf and not (use_bf16 and src.shape[2] == _round_up(C, 8) and src.is_contiguous()):
            raise RuntimeError("FusedSAMLP: bf16 feature rows need the bf16 configuration and (B, N, round_up(C, 8)) contiguous rows")
        fbit = 2 if feat_bf else 0
        R = B * M * S
        dt = torch.bfloat16 if use_bf16 else torch.float32
        bf = int(use_bf16)
        dev = xyz.device
        cout = [w.shape[0] for w in W]
        K1 = _round_up(C + 4, 16 if use_bf16 else 8)
        kpad = _round_up(C + 3, 32)
        # every weight layout of the stack (forward operands + the transposes backward needs) from one launch
        sizes = [cout[0] * K1, cout[1] * cout[0], cout[2] * cout[1], kpad * cout[0], cout[0] * cout[1], cout[1] * cout[2]]
        wbuf = torch.empty((sum(sizes),), dtype=dt, device=dev)
        _ext.call("vlp3d_sa_prep_weights", W[0].contiguous(), W[1].contiguous(), W[2].contiguous(), C, cout[0], cout[1],
                  cout[2], K1, kpad, wbuf, bf)
        parts = torch.split(wbuf, sizes)
        Wd = [parts[0].view(cout[0], K1), parts[1].view(cout[1], cout[0]), parts[2].view(cout[2], cout[1])]
        WTs = [parts[3].view(kpad, cout[0]), parts[4].view(cout[0], cout[1]), parts[5].view(cout[1], cout[2])]
        Ks = [K1, cout[0], cout[1]]
        nslab = int(_ext.load().vlp3d_sa_stat_slabs(R))  # one [sum | sumsq] slab per workgroup, no atomics
        Y, vecs = [], []
        for l in range(3):
            y = torch.empty((R, cout[l]), dtype=dt, device=dev)
            st = torch.empty((nslab, 2, cout[l]), dtype=torch.float64, device=dev)
            if l == 0:
                _ext.call("vlp3d_sa_fwd_gather", xyz, new_xyz, idx, src, B, N, M, S, C, float(radius), Wd[0], K1,
                          cout[0], y, st, bf | fbit, *cm)
            else:
                _ext.call("vlp3d_sa_fwd_layer", Y[l - 1], R, Ks[l], vecs[l - 1][0], vecs[l - 1][1], Wd[l], cout[l], y,
                          st, bf, *cm)
            bn = bns[l]
            vec = torch.empty((4, cout[l]), dtype=torch.float32, device=dev)
            track = training and bn.track_running_stats
            if track:
                if bn.num_batches_tracked is not None:  # None: the step driver increments all counters at once
                    bn.num_batches_tracked.add_(1)
                mom = 1.0 / float(bn.num_batches_tracked) if bn.momentum is None else bn.momentum
            else:
                mom = 0.0
            _ext.call("vlp3d_sa_bn_fold", st, nslab, gam[l], bet[l], bn.running_mean if (track or not training) else None,
                      bn.running_var if (track or not training) else None, cout[l], R, float(bn.eps), float(mom),
                      int(training), vec)
            Y.append(y)
            vecs.append(vec)
        out = torch.empty((B * M, cout[2]), dtype=torch.float32, device=dev)
        sel = torch.empty((B * M, cout[2]), dtype=torch.uint8, device=dev)
        out_rows = None
        if want_rows and bf and cout[2] % 8 == 0:
            out_rows = torch.empty((B * M, cout[2]), dtype=torch.bfloat16, device=dev)
            _ext.call("vlp3d_sa_pool_rows", Y[2], B * M, S, cout[2], vecs[2][0], vecs[2][1], out, out_rows, sel, bf, cm[1])
        else:
            _ext.call("vlp3d_sa_pool", Y[2], B * M, S, cout[2], vecs[2][0], vecs[2][1], out, sel, bf, cm[1])
        # (backward reads the gathered operand again for the first layer's weight gradient: the rows the forward read)
        ctx.save_for_backward(xyz, new_xyz, idx, src, out, sel, *Y, *vecs, *WTs, *gam, *bet, Wd[2])
        ctx.feat_shape = (B, N, C)
        ctx.cm = cm
        ctx.cfg = (B, N, M, S, C, R, float(radius), bf, dt, cout, Ks, training)
        ctx.fbit = fbit
        ctx.set_materialize_grads(False)   # no zero-filled cotangent for the bf16 rows output (a fill launch per backward)
        if want_rows:
            if out_rows is None:   # configuration without the bf16 copy: an empty tensor keeps the output arity fixed
                out_rows = torch.empty((0,), dtype=torch.bfloat16, device=dev)
            ctx.mark_non_differentiable(out_rows)
            return out, out_rows
        return out

    @staticmethod
    def backward(ctx, dP, _drows=None):
        if dP is None:   # (set_materialize_grads(False): the pooled output took no part in the loss)
            return (None,) * (13 + 9)
        B, N, M, S, C, R, radius, bf, dt, cout, Ks, training = ctx.cfg
        cm = ctx.cm
        sv = ctx.saved_tensors
        xyz, new_xyz, idx, feat_pm, out, sel = sv[:6]
        Y, vecs, WTs, gam, bet, W3d = sv[6:9], sv[9:12], sv[12:15], sv[15:18], sv[18:21], sv[21]
        dev = xyz.device
        dP = dP.contiguous().float()
        need = ctx.needs_input_grad  # xyz, new_xyz, idx, feat_pm, ...
        dparams = [None] * 9

        # BN-backward reductions: per-workgroup slabs for layers 1-2 (written by the MASK epilogue) and for the last layer
        # (pool_tstats: one slab per row group — nothing to clear)
        nslab = int(_ext.load().vlp3d_sa_stat_slabs(R))
        ns3 = int(_ext.load().vlp3d_sa_pool_tstats_slabs(B * M))
        # ONE zero-filled arena per backward for the scatter-add targets (one fill launch; none with the CSR adjoint and no
        # coordinate gradients): [d(features) | d(xyz) | d(new_xyz)]
        n_t3 = 0
        csr = ctx.inv is not None and need[3] and not need[0] and not need[1] and cout[0] in (64, 128)
        n_df = B * N * C if (need[3] and not csr) else 0
        n_dx = B * N * 3 if need[0] else 0
        n_dn = B * M * 3 if need[1] else 0
        arena = torch.zeros((n_df + n_dx + n_dn,), dtype=torch.float32, device=dev) if n_df + n_dx + n_dn else None
        t = [torch.empty((nslab, 2, cout[0]), dtype=torch.float64, device=dev),
             torch.empty((nslab, 2, cout[1]), dtype=torch.float64, device=dev),
             torch.empty((ns3, 2, cout[2]), dtype=torch.float64, device=dev)]
        tn = [nslab, nslab, ns3]

        # layer 3: the masked gradient lives only at the selected sample of each ball — it is synthesised inside the
        # loaders from (dP, out, sel), and its BN reductions come from the pooled tensors
        G = None
        gsel = torch.empty_like(out)
        _ext.call("vlp3d_sa_pool_tstats", dP, out, gam[2], bet[2], B * M, cout[2], t[2], gsel)
        pool = (gsel, sel, S)
        dfeat = dxyz = dnew = None
        for l in (2, 1, 0):
            c5 = torch.empty((5, cout[l]), dtype=torch.float32, device=dev)
            dg = torch.empty((cout[l],), dtype=torch.float32, device=dev)
            db = torch.empty((cout[l],), dtype=torch.float32, device=dev)
            _ext.call("vlp3d_sa_bn_bwd_consts", vecs[l], gam[l], t[l], tn[l], cout[l], R, int(training), c5, dg, db)
            dparams[3 * l + 1], dparams[3 * l + 2] = dg, db
            q = _ext.slab_queue()  # set by the step driver: all slab sums of the backward pass in one launch at its end
            dW = torch.empty((cout[l], Ks[l]) if (l > 0 or q is None) else (cout[0], C + 3), dtype=torch.float32, device=dev)
            nblk = _wgrad_blocks(R, cout[l], Ks[l])
            part = torch.empty((nblk, cout[l], Ks[l]), dtype=torch.float32, device=dev)
            last = (l == 2 and bf and SA_LAST and Ks[2] == cout[1] and int(_ext.load().vlp3d_sa_last_supported(cout[1], cout[2])))
            # the last layer WITHOUT its pre-activation Y3 (csrc/sa_last.hip): dW3 slabs / the masked gradient of layer 2 from Y2
            # and the balls' pooled rows alone.  Per kernel, where it is faster than the pooled loaders of csrc/sa_mlp.hip
            # (in-step, cfg2: weight gradient SA1 153 -> 43 us; input gradient SA1 99 -> 52, SA2 93 -> 61, SA3 46 -> 39; the small
            # modules are dominated by the per-workgroup constants — W3^T, Q = W3^T diag(beta) W3 — and keep the old kernels)
            last_w = last and R >= SA_LAST_WGRAD_MIN_ROWS
            last_d = last and R >= SA_LAST_DGRAD_MIN_ROWS
            if l > 0 and last_w:
                tiles = (R // 32) if cm[0] is None else max(1, R // 32 // 2)
                nb3 = max(16, min(SA_LAST_WBLOCKS, tiles // 8, int(WGRAD_SLAB_MB * 2 ** 20 / (cout[2] * cout[1] * 4))))
                part = torch.empty((nb3, cout[2], cout[1]), dtype=torch.float32, device=dev)
                _ext.call("vlp3d_sa_last_wgrad", Y[1], vecs[1], c5, W3d, gsel, sel, B * M, S, cout[1], cout[2], part, nb3, *cm)
                if q is not None:
                    q.add(part, nb3, dW, cout[2] * cout[1], cout[1], cout[1])
                else:
                    torch.sum(part, dim=0, out=dW)
                dparams[6] = dW.view(cout[2], cout[1], 1, 1)
            elif l > 0:
                _ext.call("vlp3d_sa_wgrad", G, Y[l], R, cout[l], c5, 0, Y[l - 1], Ks[l], vecs[l - 1][0],
                          vecs[l - 1][1], None, None, None, None, 0, 0, 0, 0, 1.0, dW, part, nblk,
                          *(pool if G is None else (None, None, 0)), bf, int(q is not None), *cm)
                if q is not None:  # after the launch: the queue may sum right away
                    q.add(part, _ext.wgrad_slabs(R, nblk), dW, cout[l] * Ks[l], Ks[l], Ks[l])
                dparams[3 * l] = dW.view(cout[l], Ks[l], 1, 1)
            if l > 0 and last_d:
                Gp = torch.empty((R, cout[1]), dtype=dt, device=dev)
                _ext.call("vlp3d_sa_last_dgrad", Y[1], vecs[1], c5, WTs[2], gsel, sel, B * M, S, cout[1], cout[2], Gp, t[1], tn[1],
                          *cm)
                G = Gp
            elif l > 0:
                Gp = torch.empty((R, cout[l - 1]), dtype=dt, device=dev)
                _ext.call("vlp3d_sa_bwd_layer", G, Y[l], R, cout[l], c5, WTs[l], cout[l - 1], Y[l - 1], vecs[l - 1], Gp,
                          t[l - 1], *(pool if G is None else (None, None, 0)), bf, *cm)
                G = Gp
            if l == 0:
                _ext.call("vlp3d_sa_wgrad", G, Y[0], R, cout[0], c5, 1, None, Ks[0], None, None, xyz, new_xyz, idx,
                          feat_pm, N, M, S, C, radius, dW, part, nblk, None, None, 0, bf | ctx.fbit, int(q is not None), *cm)
                if q is not None:  # the batched slab sum writes [xyz | features] columns directly
                    q.add(part, _ext.wgrad_slabs(R, nblk), dW, cout[0] * Ks[0], Ks[0], C + 3, ncol_out=C + 3, rot=3)
                    dparams[0] = dW.view(cout[0], C + 3, 1, 1)
                else:
                    dparams[0] = torch.cat([dW[:, C:C + 3], dW[:, :C]], dim=1).view(cout[0], C + 3, 1, 1)
                if csr:  # d(features) by a per-point gather of the layer-1 rows + one product: every element written once
                    dfeat = torch.empty((B, N, C), dtype=torch.float32, device=dev)
                    gsum = torch.empty((B * N, cout[0]), dtype=torch.float32, device=dev)
                    _ext.call("vlp3d_sa_bwd_gather_csr", G, Y[0], cout[0], c5, WTs[0], cm[0], ctx.inv[0], ctx.inv[1], B, N, C,
                              gsum, dfeat)
                elif need[0] or need[1] or need[3]:
                    kpad = WTs[0].shape[0]
                    o = n_t3
                    dfeat = arena[o:o + n_df].view(B, N, C) if need[3] else None
                    dxyz = arena[o + n_df:o + n_df + n_dx].view(B, N, 3) if need[0] else None
                    dnew = arena[o + n_df + n_dx:].view(B, M, 3) if need[1] else None
                    _ext.call("vlp3d_sa_bwd_gather", G, Y[0], cout[0], c5, WTs[0], kpad, idx, B, N, M, S, C, radius, dfeat,
                              dxyz, dnew, bf, *cm)
        return (dxyz, dnew, None, dfeat, None, None, None, None, None, None, None, None, None, *dparams)


def sa_mlp_pool(xyz, new_xyz, idx, feat_pm, radius, mlp_module, use_bf16, cmap=None, inv=None, feat_c=None, feat_rows=None,
                want_rows=False):
    """Run the 3-layer SharedMLP + max-pool of an SA layer fused.  Returns pooled (B, npoint, C3) fp32.
    cmap: (rowptr, crow) of _lib.sa_compact(idx, N) — evaluate the stack on the distinct rows only (bf16 configuration);
    inv: (inv_start, inv_rows) of _lib.sa_inverse(idx, N, cmap) — atomic-free backward of the gather (bf16 configuration).
    feat_pm may be BF16 rows (B, N, round_up(feat_c, 8)), columns beyond feat_c zero (bf16 configuration, no gradient to them);
    or fp32 rows with feat_rows = their bf16 copy (what the kernels then read; the gradient goes to feat_pm).  want_rows: also
    return this level's pooled output as bf16 rows — the next level's feat_rows."""
    bns = [layer.bn.bn for layer in mlp_module]
    params = []
    for layer in mlp_module:
        params += [layer.conv.weight, layer.bn.bn.weight, layer.bn.bn.bias]
    B, M = new_xyz.shape[:2]
    res = FusedSAMLP.apply(xyz, new_xyz, idx, feat_pm, radius, bns, bns[0].training, use_bf16, cmap, inv, feat_c, feat_rows,
                           want_rows, *params)
    if want_rows:   # -> (pooled (B, M, C3) fp32, the same as bf16 rows (B, M, C3) or None)
        out, rows = res
        return out.view(B, M, -1), (rows.view(B, M, -1) if rows.numel() else None)
    return res.view(B, M, -1)
