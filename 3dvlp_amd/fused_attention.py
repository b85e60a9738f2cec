"""Autograd wrapper of the fused HIP attention core (csrc/sdpa.hip, vlp3d_sdpa_fwd/bwd).

Computes softmax(q k^T / sqrt(d) [+bias | *weights] [key mask]) v per head from the projected
(b, n, h*d) tensors — the part of models/transformer/attention.py:63-75 between the fc_q/k/v and
fc_o linears — without ever materialising the (b,h,nq,nk) attention matrix.
"""
import os

import torch
from torch.autograd import Function

from . import _lib as _ext

_ext.load()

# False: exact-fp32 MFMA (parity configuration).  True: bf16 MFMA operands, fp32 accumulation/softmax — set by the
# step driver together with the bf16 grouped MLPs (GroundingStep(sa_dtype=torch.bfloat16)).
BF16_MMA = False


def _key_mask(attention_mask, b, nk):
    """Only a 4-D (b,1,1,nk) mask is a pure per-batch key mask; None when there is no mask, False for any other shape.
    (A 2-D mask is NOT accepted: the reference's masked_fill broadcasts (x,nk) over (b,h,nq,nk) as a (nq,nk) mask, so
    reading a (b,nk) tensor as a key mask would silently differ whenever b == nq.)"""
    if attention_mask is None:
        return None
    m = attention_mask
    if m.dim() == 4 and m.shape[1] == 1 and m.shape[2] == 1 and m.shape[0] == b and m.shape[3] == nk:
        return m.reshape(b, nk)
    return False  # a mask, but not one the kernel takes


def supported(d_k, d_v, attention_mask, nk, b=None):
    """Shapes the fused core takes.  `b` = batch size of the queries (masks are checked against it); callers fall back
    to the unfused formulation when this returns False."""
    if d_k != 32 or d_v != 32:
        return False
    if attention_mask is None:
        return True
    return _key_mask(attention_mask, attention_mask.shape[0] if b is None else b, nk) is not False


class _SDPA(Function):
    @staticmethod
    def forward(ctx, q, k, v, bias, H, bias_mode, mask, bf16_mma):
        # q / k / v may be column blocks of one merged projection output (row-strided views): no copies
        q, k, v = (t if _ext._row_stride_ok(t) else t.contiguous() for t in (q, k, v))
        bias = bias.contiguous() if bias is not None else None
        out, lse = _ext.sdpa_fwd(q, k, v, H, bias, bias_mode, mask, bf16_mma)
        ctx.save_for_backward(q, k, v, bias, mask, out, lse)
        ctx.H, ctx.bias_mode, ctx.bf16_mma = H, bias_mode, bf16_mma
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, bias, mask, out, lse = ctx.saved_tensors
        need_dbias = bias is not None and ctx.needs_input_grad[3]
        dq, dk, dv, dbias = _ext.sdpa_bwd(q, k, v, ctx.H, bias, ctx.bias_mode, mask, out, lse, dout.contiguous(),
                                          need_dbias, ctx.bf16_mma)
        return dq, dk, dv, dbias, None, None, None, None


class _SDPAMerged(Function):
    """Same core on MERGED projections: `a` is (b, n, 3*h*32) = [q | k | v] (self-attention, `b_` None) or `a` = q
    (b, nq, h*32) and `b_` = [k | v] (b, nk, 2*h*32).  The column blocks go to the kernels as row-strided views and the
    backward returns ONE gradient tensor per merged input — directly the dY of the merged linear layer (no split /
    cat copies on either side)."""

    @staticmethod
    def forward(ctx, a, b_, bias, H, bias_mode, mask, bf16_mma):
        HD = H * 32
        a = a.contiguous()
        if b_ is None:
            q, k, v = a[..., :HD], a[..., HD:2 * HD], a[..., 2 * HD:]
        else:
            b_ = b_.contiguous()
            q, k, v = a, b_[..., :HD], b_[..., HD:]
        bias = bias.contiguous() if bias is not None else None
        out, lse = _ext.sdpa_fwd(q, k, v, H, bias, bias_mode, mask, bf16_mma)
        ctx.save_for_backward(a, b_, bias, mask, out, lse)
        ctx.H, ctx.bias_mode, ctx.bf16_mma = H, bias_mode, bf16_mma
        return out

    @staticmethod
    def backward(ctx, dout):
        a, b_, bias, mask, out, lse = ctx.saved_tensors
        HD = ctx.H * 32
        if b_ is None:
            q, k, v = a[..., :HD], a[..., HD:2 * HD], a[..., 2 * HD:]
        else:
            q, k, v = a, b_[..., :HD], b_[..., HD:]
        need_dbias = bias is not None and ctx.needs_input_grad[2]
        dq, dk, dv, dbias = _ext.sdpa_bwd(q, k, v, ctx.H, bias, ctx.bias_mode, mask, out, lse, dout.contiguous(),
                                          need_dbias, ctx.bf16_mma)
        if b_ is None:
            da, db = dq._base if dq._base is not None else torch.cat([dq, dk, dv], -1), None
        else:
            da, db = dq, (dk._base if dk._base is not None else torch.cat([dk, dv], -1))
        return da, db, dbias, None, None, None, None


def rows_supported(nq, nk, batch_heads):
    """Shapes the bf16-rows cores take (vlp3d_sdpa_fwd_io / _bwd_io: the LDS kernels of the bf16-MFMA configuration)."""
    return nk <= 288 and nq <= 512 and batch_heads >= int(os.environ.get("VLP3D_SDPA_LDS_MIN_BH", 64))


class _SDPARows(Function):
    """The cores of the match decoder on bf16 ROWS (vlp3d_sdpa_fwd_io / _bwd_io): q, out (and k, v of a merged self-attention
    projection) cross memory once, as bf16 — SURVEY.md §8(d)'s bytes.  The bf16-MFMA kernels round q / k / v to bf16 anyway, so
    the forward numbers are those of the fp32-row cores.

    Autograd sees fp32 SHELLS: `a_shell` and the returned out shell are fp32 tensors of the right shape whose storage is never
    read or written.  (The engine converts a gradient to the dtype of the tensor it belongs to: a bf16 tensor in the graph
    would receive its gradient through an extra conversion launch, rounded.)  The values travel beside the shells as bf16 rows:
    `a_rows` in — q (b, nq, h*32) with `b_` = [k | v] fp32, or the merged [q | k | v] (b, n, 3*h*32) with `b_` None — and
    `out_rows` out (non-differentiable)."""

    @staticmethod
    def forward(ctx, a_shell, a_rows, b_, H, mask, b_rows=None):
        HD = H * 32
        if b_ is None:
            q, k, v = a_rows[..., :HD], a_rows[..., HD:2 * HD], a_rows[..., 2 * HD:]
        else:
            # b_rows: [k | v] as bf16 rows too (b_ is then their shell): every operand of the core at SURVEY 8(d)'s size
            b_ = b_.contiguous() if b_rows is None else b_rows.contiguous()
            q, k, v = a_rows, b_[..., :HD], b_[..., HD:]
        out_rows, lse = _ext.sdpa_fwd_rows(q, k, v, H, mask, True)
        ctx.save_for_backward(a_rows, b_, mask, out_rows, lse)
        ctx.H = H
        ctx.mark_non_differentiable(out_rows)
        ctx.set_materialize_grads(False)  # (no zero tensor for the rows output's absent gradient)
        return torch.empty(out_rows.shape, dtype=torch.float32, device=out_rows.device), out_rows

    @staticmethod
    def backward(ctx, dout, _drows):
        if dout is None:
            return None, None, None, None, None, None
        a_rows, b_, mask, out_rows, lse = ctx.saved_tensors
        HD = ctx.H * 32
        if b_ is None:
            q, k, v = a_rows[..., :HD], a_rows[..., HD:2 * HD], a_rows[..., 2 * HD:]
        else:
            q, k, v = a_rows, b_[..., :HD], b_[..., HD:]
        dq, dk, dv = _ext.sdpa_bwd_rows(q, k, v, ctx.H, mask, out_rows, lse, dout.contiguous())
        if b_ is None:
            return dq._base, None, None, None, None, None
        return dq, None, dk._base, None, None, None


def sdpa_rows(a_shell, a_rows, b_, h, attention_mask=None, b_rows=None):
    """-> (out shell fp32, out rows bf16); see _SDPARows.  Check rows_supported() first."""
    nk = a_rows.shape[1] if b_ is None else b_.shape[1]
    mask = _key_mask(attention_mask, a_rows.shape[0], nk)
    if mask is False:
        raise RuntimeError("fused sdpa: unsupported attention_mask shape")
    if mask is not None:
        mask = mask.to(torch.float32).contiguous()
    return _SDPARows.apply(a_shell, a_rows, b_, h, mask, b_rows)


def sdpa_merged(a, b_, h, attention_weights=None, way="add", attention_mask=None, bf16_mma=None):
    """a = [q | k | v] merged (b_ None) or a = q, b_ = [k | v] merged; see _SDPAMerged."""
    nk = a.shape[1] if b_ is None else b_.shape[1]
    mask = _key_mask(attention_mask, a.shape[0], nk)
    if mask is False:
        raise RuntimeError("fused sdpa: unsupported attention_mask shape")
    if mask is not None:
        mask = mask.to(torch.float32).contiguous()
    bias_mode = 0 if attention_weights is None else (1 if way == "add" else 2)
    return _SDPAMerged.apply(a, b_, attention_weights, h, bias_mode, mask, BF16_MMA if bf16_mma is None else bool(bf16_mma))


def sdpa(q, k, v, h, attention_weights=None, way="add", attention_mask=None, bf16_mma=None):
    """q (b,nq,h*32), k/v (b,nk,h*32) fp32 CUDA tensors -> (b,nq,h*32).  bf16_mma=None: module default BF16_MMA."""
    mask = _key_mask(attention_mask, q.shape[0], k.shape[1])
    if mask is False:
        raise RuntimeError("fused sdpa: unsupported attention_mask shape")
    if mask is not None:
        mask = mask.to(torch.float32).contiguous()
    bias_mode = 0 if attention_weights is None else (1 if way == "add" else 2)
    return _SDPA.apply(q, k, v, attention_weights, h, bias_mode, mask, BF16_MMA if bf16_mma is None else bool(bf16_mma))
