"""Autograd wrappers of the fused glue kernels (csrc/glue.hip): each stands for a dozen element-wise / index launches of
the reference's Python between the matrix-core kernels.  CUDA fp32 tensors only (callers keep the op-by-op torch form
for host-side tests)."""
import numpy as np
import torch
from torch.autograd import Function

from . import _lib as _ext

_ext.load()


class _RoiSplit(Function):
    """out (B,K,ld) = [heading_reg NH | heading_cls NH | box 6 | objectness 2 | sem NC | pad] -> the tensors
    StandardROIHeads.forward puts into data_dict (roi_heads.py:135-147) + the two arg-max masks."""

    @staticmethod
    def forward(ctx, out, NH, NC):
        out = out.contiguous()
        B, K, ld = out.shape
        R, dev = B * K, out.device
        e = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        hreg, hres, hcls, rois, obj, sem = e(B, K, NH), e(B, K, NH), e(B, K, NH), e(B, K, 6), e(B, K, 2), e(B, K, NC)
        omask = torch.empty((B, K), dtype=torch.int64, device=dev)
        sarg = torch.empty((B, K), dtype=torch.int64, device=dev)
        scale = float(np.pi / NH)
        _ext.call("vlp3d_roi_split", out, ld, R, NH, NC, scale, hreg, hres, hcls, rois, obj, sem, omask, sarg)
        ctx.save_for_backward(rois)
        ctx.cfg = (R, NH, NC, scale, ld, out.shape)
        ctx.set_materialize_grads(False)  # unused outputs arrive as None (the kernels take NULL), not as zero fills
        ctx.mark_non_differentiable(omask, sarg)
        return hreg, hres, hcls, rois, obj, sem, omask, sarg

    @staticmethod
    def backward(ctx, d_hreg, d_hres, d_hcls, d_rois, d_obj, d_sem, _m, _s):
        (rois,) = ctx.saved_tensors
        R, NH, NC, scale, ld, shape = ctx.cfg
        c = lambda g: None if g is None else g.contiguous()
        d_out = torch.empty(shape, dtype=torch.float32, device=rois.device)
        _ext.call("vlp3d_roi_split_bwd", c(d_hreg), c(d_hres), c(d_hcls), c(d_rois), c(d_obj), c(d_sem), rois, R, NH, NC,
                  scale, d_out, ld)
        return d_out, None, None


def roi_split(out, NH, NC):
    return _RoiSplit.apply(out, NH, NC)


class _VoteEpilogue(Function):
    """(seed_xyz (B,S,3), seed_pm (B,S,C), net (B*S, ld)) -> vote_xyz (B,S,3), L2-normalised vote_features (B,S,C)."""

    @staticmethod
    def forward(ctx, seed_xyz, seed_pm, net):
        seed_xyz, seed_pm = seed_xyz.contiguous(), seed_pm.contiguous()
        B, S, C = seed_pm.shape
        ld = net.stride(0)
        assert net.stride(1) == 1 and net.shape[0] == B * S
        vx = torch.empty((B, S, 3), dtype=torch.float32, device=net.device)
        vf = torch.empty((B, S, C), dtype=torch.float32, device=net.device)
        norm = torch.empty((B * S,), dtype=torch.float32, device=net.device)
        _ext.call("vlp3d_vote_epilogue", seed_xyz, seed_pm, net, ld, B * S, C, vx, vf, norm)
        ctx.save_for_backward(vf, norm)
        ctx.cfg = (B * S, C, ld, tuple(net.shape))
        ctx.set_materialize_grads(False)  # unused outputs arrive as None (the kernels take NULL), not as zero fills
        return vx, vf

    @staticmethod
    def backward(ctx, d_vx, d_vf):
        vf, norm = ctx.saved_tensors
        R, C, ld, nshape = ctx.cfg
        c = lambda g: None if g is None else g.contiguous()
        d_seed = torch.empty_like(vf)
        d_net_full = torch.empty((R, ld), dtype=torch.float32, device=vf.device)
        _ext.call("vlp3d_vote_epilogue_bwd", c(d_vx), c(d_vf), vf, norm, R, C, d_seed, d_net_full, ld)
        return c(d_vx), d_seed, d_net_full[:, :nshape[1]]


def vote_epilogue(seed_xyz, seed_pm, net):
    return _VoteEpilogue.apply(seed_xyz, seed_pm, net)


class _L2NormRows(Function):
    @staticmethod
    def forward(ctx, x, eps):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]).contiguous().float()
        y = torch.empty_like(x2)
        norm = torch.empty((x2.shape[0],), dtype=torch.float32, device=x.device)
        _ext.call("vlp3d_l2norm_rows", x2, x2.shape[0], x2.shape[1], float(eps), y, norm)
        ctx.save_for_backward(y, norm)
        ctx.eps, ctx.shape = float(eps), shape
        return y.view(shape)

    @staticmethod
    def backward(ctx, g):
        y, norm = ctx.saved_tensors
        g2 = g.reshape(y.shape).contiguous()
        dx = torch.empty_like(y)
        _ext.call("vlp3d_l2norm_rows_bwd", g2, y, norm, y.shape[0], y.shape[1], ctx.eps, dx)
        return dx.view(ctx.shape), None


def l2norm_rows(x, eps=1e-12):
    """F.normalize(x, dim=-1) on the last dimension."""
    return _L2NormRows.apply(x, eps)


@torch.no_grad()
def relation_inputs(pc, seed_inds, vote_inds, corners, col0=6):
    """-> obj_feat (B,K,128), manual_bbox_feat (B,K,27), centre (B,K,3); no gradient (all inputs are detached data).
    pc: point-major rows holding the 128 multiview channels in columns col0..col0+127 (raw cloud: 6; k/feat_pm: 3), fp32 or
    bf16 (k/feat_bf: the loader's bf16 copy, any row stride)."""
    B, N, Cpc = pc.shape
    K = corners.shape[1]
    dev = pc.device
    obj_feat = torch.empty((B, K, 128), dtype=torch.float32, device=dev)
    bbox = torch.empty((B, K, 27), dtype=torch.float32, device=dev)
    centre = torch.empty((B, K, 3), dtype=torch.float32, device=dev)
    entry = "vlp3d_relation_inputs_bf16" if pc.dtype == torch.bfloat16 else "vlp3d_relation_inputs"
    _ext.call(entry, pc.contiguous(), Cpc, int(col0), N, seed_inds.contiguous().int(), seed_inds.shape[1],
              vote_inds.contiguous().int(), corners.contiguous().float(), B, K, obj_feat, bbox, centre)
    return obj_feat, bbox, centre


class _CopyPaste(Function):
    @staticmethod
    def forward(ctx, features, obj_mask, coin):
        B, K, D = features.shape
        f2 = features.contiguous().view(B * K, D)
        src = torch.empty((B * K,), dtype=torch.int32, device=features.device)
        _ext.call("vlp3d_copy_paste_map", obj_mask.contiguous(), B, K, coin, src)
        out = torch.empty_like(f2)
        _ext.call("vlp3d_gather_rows", f2, src, B * K, D, out)
        ctx.save_for_backward(src)
        ctx.shape = (B, K, D)
        return out.view(B, K, D)

    @staticmethod
    def backward(ctx, g):
        (src,) = ctx.saved_tensors
        B, K, D = ctx.shape
        dx = torch.zeros((B * K, D), dtype=torch.float32, device=g.device)
        _ext.call("vlp3d_scatter_rows_add", g.contiguous().view(B * K, D), src, B * K, D, dx)
        return dx.view(B, K, D), None, None


def copy_paste(features, obj_mask, coin):
    """match_module.py:97-121 gated by `coin < 0.5` (device scalar): features (B,K,D) fp32, obj_mask (B,K) int64."""
    return _CopyPaste.apply(features, obj_mask, coin.reshape(1).float())


class _SmallK(Function):
    """base + x[:, :K] @ W.T + b for K <= 32 (csrc/glue.hip: smallk); x carries no gradient (input data)."""

    @staticmethod
    def forward(ctx, x, weight, bias, base):
        N, K = weight.shape
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        R = x2.shape[0]
        out = torch.empty((R, N), dtype=torch.float32, device=x.device)
        b2 = None if base is None else base.reshape(R, N).contiguous()
        _ext.call("vlp3d_smallk_fwd", x2, x2.shape[1], weight.contiguous(), bias, b2, R, K, N, out)
        ctx.save_for_backward(x2)
        ctx.cfg = (R, K, N, bias is not None, base is not None, None if base is None else base.shape)
        return out.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dout):
        (x2,) = ctx.saved_tensors
        R, K, N, has_bias, has_base, bshape = ctx.cfg
        d2 = dout.reshape(R, N).contiguous()
        nblk = (R + 63) // 64
        slabs = torch.empty((nblk, N * 32 + N), dtype=torch.float32, device=dout.device)
        _ext.call("vlp3d_smallk_bwd", d2, x2, x2.shape[1], R, K, N, slabs)
        dW = torch.empty((N, K), dtype=torch.float32, device=dout.device)
        db = torch.empty((N,), dtype=torch.float32, device=dout.device)
        _ext.reduce_slabs(slabs, nblk, dW, N * 32, 32, K, db, N, ncol_out=K)
        # VIEWS on purpose (SlabReduceQueue): the queue still references dW / db, and autograd copies — at once, before the
        # deferred sum has run — a gradient tensor that anything else references
        return None, dW.view(N, K), (db.view(N) if has_bias else None), (dout.reshape(bshape) if has_base else None)


def smallk_supported(x, weight):
    return (x.is_cuda and x.dtype == torch.float32 and weight.shape[1] <= 32 and weight.shape[0] in (64, 128, 256)
            and not x.requires_grad)


def small_linear(x, weight, bias=None, base=None):
    """base + F.linear(x, weight, bias) for weight (N, K <= 32): one launch forward, one + the batched slab sum backward."""
    return _SmallK.apply(x, weight, bias, base)


class _RowDot(Function):
    """F.linear(x, weight (1,K), bias (1)) -> (R,) (csrc/glue.hip: rowdot)."""
    ROWS_PER_BLOCK = 64

    @staticmethod
    def forward(ctx, x, weight, bias):
        K = x.shape[-1]
        x2 = x.reshape(-1, K).contiguous()
        R = x2.shape[0]
        w = weight.reshape(K).contiguous()
        y = torch.empty((R,), dtype=torch.float32, device=x.device)
        _ext.call("vlp3d_rowdot_fwd", x2, w, bias, R, K, y)
        ctx.save_for_backward(x2, w)
        ctx.cfg = (R, K, bias is not None, x.shape, weight.shape)
        return y.view(*x.shape[:-1], 1)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        R, K, has_bias, xshape, wshape = ctx.cfg
        d = dy.reshape(R).contiguous()
        rpb = _RowDot.ROWS_PER_BLOCK
        nblk = (R + rpb - 1) // rpb
        slabs = torch.empty((nblk, K + 4), dtype=torch.float32, device=dy.device)
        dx = torch.empty((R, K), dtype=torch.float32, device=dy.device) if ctx.needs_input_grad[0] else None
        _ext.call("vlp3d_rowdot_bwd", d, x2, w, R, K, rpb, dx, slabs)
        dwb = torch.empty((K + 4,), dtype=torch.float32, device=dy.device)
        _ext.reduce_slabs(slabs, nblk, dwb, K + 4, K + 4, K + 4)
        return (None if dx is None else dx.view(xshape)), dwb[:K].view(wshape), (dwb[K:K + 1] if has_bias else None)


def rowdot_supported(x, weight):
    K = x.shape[-1]
    return x.is_cuda and x.dtype == torch.float32 and weight.shape[0] == 1 and K % 4 == 0 and 4 <= K <= 128


def rowdot(x, weight, bias=None):
    return _RowDot.apply(x, weight, bias)
