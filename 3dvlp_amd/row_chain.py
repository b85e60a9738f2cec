"""Row-local runs of the decoder layers as ONE forward launch each (csrc/rows_chain.hip, include/vlp3d.h vlp3d_rows_chain).

A chain is a list of stages over the rows of a matrix X; stage s maps the current rows t_s to t_{s+1}:

    Linear(W, b)                                     t_{s+1} = t_s W^T + b
    Linear + act ("relu" | "gelu") + dropout p       t_{s+1} = dropout_p(act(t_s W^T + b))
    Linear + add & norm                              t_{s+1} = LayerNorm(res + dropout_p(t_s W^T + b)),  res = a tensor given by
                                                     the caller or an earlier t_j of the same chain

i.e. attention.py:75 fc_o -> :128-130 add & norm -> the next projection; mmattention.py:36-50 FFN -> :84-86 add & norm;
match_module.py:40-47 Linear/GELU/Dropout.  The forward pass of a chain is one kernel launch whose 64-row tiles stay in LDS
between the stages; every t_{s+1} (and what backward needs: pre-activations, xhat, rstd) is stored on the way.  Backward walks
the stages in reverse on the existing entry points (vlp3d_sum_norm_bwd, vlp3d_act_dropout, vlp3d_linear_dgrad with the
residual gradient as `base`, the queued weight gradients) — the launches autograd would have made for the unfused modules.

bf16 MFMA operands only (the timing configuration of the step driver): callers check `supported()` and use the unfused
modules otherwise — that is a different sequence of the same HIP entry points, never a library or CPU path.
"""
import os

import torch
from torch.autograd import Function

from . import _lib as _ext
from . import add_norm, mfma_linear

ENABLED = os.environ.get("VLP3D_ROW_CHAIN", "1") != "0"
_ACTS = {"relu": 0, "gelu": 1}


def linear(weight, bias=None, act=None, p=0.0):
    """Stage: Linear [+ act + dropout_p]."""
    return {"W": weight, "b": bias, "act": act, "p": float(p), "ln": None}


def linear_add_norm(weight, bias, norm, res, p=0.0):
    """Stage: Linear -> LayerNorm `norm`(res + dropout_p(.)).  res: a tensor (R, 128) or ("tile", j) = this chain's t_j."""
    return {"W": weight, "b": bias, "act": None, "p": float(p), "ln": norm, "res": res}


def supported(x, stages):
    if not (ENABLED and mfma_linear.BF16_MMA and x.is_cuda and x.dtype == torch.float32
            and not torch.is_autocast_enabled("cuda")):
        return False
    R = x.numel() // x.shape[-1]
    if R % 64 or not 1 <= len(stages) <= 6:
        return False
    K = x.shape[-1]
    for s, st in enumerate(stages):
        N, Kw = st["W"].shape
        last = s + 1 == len(stages)
        if Kw != K or K not in (128, 256) or N % 128 or N > (384 if last else 256) or R * N >= 2 ** 32:
            return False
        if st["W"].dtype != torch.float32 or not mfma_linear.supported(x.new_empty((R, K)), st["W"]):
            return False
        if st["ln"] is not None:
            ln = st["ln"]
            if N != 128 or tuple(ln.normalized_shape) != (128,) or not ln.elementwise_affine or ln.bias is None:
                return False
        K = N
    return True


class _Chain(Function):
    @staticmethod
    def forward(ctx, spec, seed, X, *tensors):
        # spec: per stage (iW, ib, act_kind, act_p, act_call, has_ln, res, ig, ibeta, ln_p, ln_call, eps); indices into
        # `tensors`; res = ("ext", index) | ("tile", j)
        R = X.shape[0]
        dev = X.device
        tiles, zs, xhats, rstds, descs = [X], [], [], [], []
        for (iW, ib, act_kind, act_p, act_call, has_ln, res, ig, ibeta, ln_p, ln_call, eps) in spec:
            W = tensors[iW]
            N, K = W.shape
            d = {"W": W, "bias": None if ib is None else tensors[ib], "N": N, "K": K, "act_kind": act_kind, "act_p": act_p,
                 "act_call": act_call, "has_ln": int(has_ln), "ln_p": ln_p, "ln_call": ln_call, "eps": eps}
            out = torch.empty((R, N), dtype=torch.float32, device=dev)
            z = xh = rs = None
            if has_ln:
                xh = torch.empty((R, N), dtype=torch.float32, device=dev)
                rs = torch.empty((R,), dtype=torch.float32, device=dev)
                d.update(res=tensors[res[1]] if res[0] == "ext" else tiles[res[1]], gamma=tensors[ig], beta=tensors[ibeta],
                         ln_out=out, xhat=xh, rstd=rs)
            elif act_kind >= 0:
                z = torch.empty((R, N), dtype=torch.float32, device=dev)
                d.update(v_out=z, h_out=out)
            else:
                d.update(v_out=out)
            descs.append(d)
            tiles.append(out)
            zs.append(z)
            xhats.append(xh)
            rstds.append(rs)
        _ext.rows_chain(X, descs, seed)
        ctx.spec = spec
        ctx.n_tensors = len(tensors)
        keep = [seed, X] + list(tensors) + tiles[1:] + [t for t in zs + xhats + rstds if t is not None]
        ctx.layout = ([t is not None for t in zs], [t is not None for t in xhats])
        ctx.save_for_backward(*keep)
        ctx.set_materialize_grads(False)
        return tuple(tiles[1:])

    @staticmethod
    def backward(ctx, *douts):
        spec = ctx.spec
        n = len(spec)
        saved = list(ctx.saved_tensors)
        seed, X = saved[0], saved[1]
        tensors = saved[2:2 + ctx.n_tensors]
        tiles = [X] + saved[2 + ctx.n_tensors:2 + ctx.n_tensors + n]
        rest = iter(saved[2 + ctx.n_tensors + n:])
        has_z, has_ln = ctx.layout
        zs = [next(rest) if h else None for h in has_z]
        xhats = [next(rest) if h else None for h in has_ln]
        rstds = [next(rest) if h else None for h in has_ln]
        R = X.shape[0]
        grads = [None] * ctx.n_tensors
        pend = [None] + [None if d is None else d.reshape(R, -1).contiguous() for d in douts]

        def accumulate(slot, g):
            return g if slot is None else slot + g

        for s in reversed(range(n)):
            g = pend[s + 1]
            if g is None:
                continue
            (iW, ib, act_kind, act_p, act_call, has, res, ig, ibeta, ln_p, ln_call, eps) = spec[s]
            W = tensors[iW]
            N, K = W.shape
            if has:
                dres, g, dgam, dbet = add_norm.norm_backward(g, None, xhats[s], rstds[s], None, tensors[ig], R, N, ln_p, seed,
                                                             ln_call, True)
                grads[ig], grads[ibeta] = dgam, dbet
                if res[0] == "ext":
                    grads[res[1]] = accumulate(grads[res[1]], dres)
                else:
                    pend[res[1]] = accumulate(pend[res[1]], dres)
            elif act_kind >= 0:
                gz = torch.empty_like(g)
                _ext.call("vlp3d_act_dropout", zs[s], g, g.numel(), act_kind, act_p, seed, act_call, gz, None)
                g = gz
            want_db = ib is not None and ctx.needs_input_grad[3 + ib]
            if ctx.needs_input_grad[3 + iW]:
                dw, db = mfma_linear.weight_grad(g, tiles[s], N, K, want_db, 1)
                grads[iW] = dw
                if want_db:
                    grads[ib] = db
            if s > 0 or ctx.needs_input_grad[2]:
                dx = torch.empty((R, K), dtype=torch.float32, device=g.device)
                _ext.call("vlp3d_linear_dgrad", g, W, R, N, K, dx, pend[s], 1)  # base: the gradient already waiting for t_s
                pend[s] = dx
        gX = pend[0]
        for i, t in enumerate(tensors):  # residual tensors keep the caller's shape
            if grads[i] is not None and grads[i].shape != t.shape:
                grads[i] = grads[i].view(t.shape)
        return (None, None, gX) + tuple(grads)


def run(x, stages, training=True):
    """x (..., K0) -> tuple of t_1..t_n, each (R, N_s) with R = rows of x.  Check `supported(x, stages)` first."""
    X = x.reshape(-1, x.shape[-1]).contiguous()
    R = X.shape[0]
    tensors, spec = [], []

    def slot(t):
        tensors.append(t)
        return len(tensors) - 1

    for st in stages:
        iW = slot(st["W"].contiguous())
        ib = None if st["b"] is None else slot(st["b"].contiguous())
        p = st["p"] if training else 0.0
        if st["ln"] is not None:
            ln = st["ln"]
            res = st["res"]
            if isinstance(res, torch.Tensor):
                res = ("ext", slot(res.reshape(R, -1).contiguous()))
            spec.append((iW, ib, -1, 0.0, 0, True, res, slot(ln.weight.contiguous()), slot(ln.bias.contiguous()), p,
                         add_norm.next_call(), float(ln.eps)))
        elif st["act"] is not None:
            spec.append((iW, ib, _ACTS[st["act"]], p, add_norm.next_call(), False, None, None, None, 0.0, 0, 0.0))
        else:
            spec.append((iW, ib, -1, 0.0, 0, False, None, None, None, 0.0, 0, 0.0))
    return _Chain.apply(tuple(spec), add_norm.state(X.device), X, *tensors)
