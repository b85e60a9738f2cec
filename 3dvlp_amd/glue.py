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
def relation_inputs(pc, seed_inds, vote_inds, corners):
    """-> obj_feat (B,K,128), manual_bbox_feat (B,K,27), centre (B,K,3); no gradient (all inputs are detached data)."""
    B, N, Cpc = pc.shape
    K = corners.shape[1]
    dev = pc.device
    obj_feat = torch.empty((B, K, 128), dtype=torch.float32, device=dev)
    bbox = torch.empty((B, K, 27), dtype=torch.float32, device=dev)
    centre = torch.empty((B, K, 3), dtype=torch.float32, device=dev)
    _ext.call("vlp3d_relation_inputs", pc.contiguous(), Cpc, N, seed_inds.contiguous().int(), seed_inds.shape[1],
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
