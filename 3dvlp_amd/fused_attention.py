"""Autograd wrapper of the fused HIP attention core (csrc/sdpa.hip, vlp3d_sdpa_fwd/bwd).

Computes softmax(q k^T / sqrt(d) [+bias | *weights] [key mask]) v per head from the projected
(b, n, h*d) tensors — the part of models/transformer/attention.py:63-75 between the fc_q/k/v and
fc_o linears — without ever materialising the (b,h,nq,nk) attention matrix.
"""
import torch
from torch.autograd import Function

from . import _lib as _ext

_ext.load()

# False: exact-fp32 MFMA (parity configuration).  True: bf16 MFMA operands, fp32 accumulation/softmax — set by the
# step driver together with the bf16 grouped MLPs (GroundingStep(sa_dtype=torch.bfloat16)).
BF16_MMA = False


def _key_mask(attention_mask, b, nk):
    """Accept the mask shapes that are pure key masks ((b,1,1,nk) / (b,nk)); None otherwise."""
    if attention_mask is None:
        return None
    m = attention_mask
    if m.dim() == 4 and m.shape[1] == 1 and m.shape[2] == 1 and m.shape[0] == b and m.shape[3] == nk:
        return m.reshape(b, nk)
    if m.dim() == 2 and tuple(m.shape) == (b, nk):
        return m
    return False  # a mask, but not one the kernel takes


def supported(d_k, d_v, attention_mask, nk):
    if d_k != 32 or d_v != 32:
        return False
    if attention_mask is None:
        return True
    m = attention_mask
    return (m.dim() == 4 and m.shape[1] == 1 and m.shape[2] == 1 and m.shape[3] == nk) or \
           (m.dim() == 2 and m.shape[1] == nk)


class _SDPA(Function):
    @staticmethod
    def forward(ctx, q, k, v, bias, H, bias_mode, mask, bf16_mma):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
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


def sdpa(q, k, v, h, attention_weights=None, way="add", attention_mask=None, bf16_mma=None):
    """q (b,nq,h*32), k/v (b,nk,h*32) fp32 CUDA tensors -> (b,nq,h*32).  bf16_mma=None: module default BF16_MMA."""
    mask = _key_mask(attention_mask, q.shape[0], k.shape[1])
    if mask is False:
        raise RuntimeError("fused sdpa: unsupported attention_mask shape")
    if mask is not None:
        mask = mask.to(torch.float32).contiguous()
    bias_mode = 0 if attention_weights is None else (1 if way == "add" else 2)
    return _SDPA.apply(q, k, v, attention_weights, h, bias_mode, mask, BF16_MMA if bf16_mma is None else bool(bf16_mma))
