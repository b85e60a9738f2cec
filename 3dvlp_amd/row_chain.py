"""Row-local runs of the decoder layers as ONE forward launch each (csrc/rows_chain.hip, include/vlp3d.h vlp3d_rows_chain).

A chain is a list of stages over the rows of a matrix X; stage s maps the current rows t_s to t_{s+1}:

    Linear(W, b)                                     t_{s+1} = t_s W^T + b
    Linear + act ("relu" | "gelu") + dropout p       t_{s+1} = dropout_p(act(t_s W^T + b))
    Linear + add & norm                              t_{s+1} = LayerNorm(res + dropout_p(t_s W^T + b)),  res = a tensor given by
                                                     the caller or an earlier t_j of the same chain

i.e. attention.py:75 fc_o -> :128-130 add & norm -> the next projection; mmattention.py:36-50 FFN -> :84-86 add & norm;
match_module.py:40-47 Linear/GELU/Dropout.  The forward pass of a chain is one kernel launch whose 32-row tiles stay in LDS
between the stages; every t_{s+1} (and what backward needs: pre-activations, xhat, rstd) is stored on the way.  Backward is
ONE launch too for the stages up to 256 columns wide (vlp3d_rows_chain_bwd: input-gradient products on the transposed weights,
add & norm / activation / dropout backward and the residual adds at the points between them, the gradients the queued
weight-gradient kernels read stored on the way); a wider top stage (the merged q|k|v projection) and passes without transposed
weights at hand walk the stages on the unfused entry points (vlp3d_sum_norm_bwd, vlp3d_act_dropout, vlp3d_linear_dgrad).

bf16 MFMA operands only (the timing configuration of the step driver): callers check `supported()` and use the unfused
modules otherwise — that is a different sequence of the same HIP entry points, never a library or CPU path.
"""
import os

import torch
from torch.autograd import Function

from . import _lib as _ext
from . import add_norm, mfma_linear, row_mlp

ENABLED = os.environ.get("VLP3D_ROW_CHAIN", "1") != "0"
FUSED_BACKWARD = os.environ.get("VLP3D_ROW_CHAIN_BWD", "1") != "0"  # the chain's backward as one launch (+ the weight gradients)
_ACTS = {"relu": 0, "gelu": 1}


def linear(weight, bias=None, act=None, p=0.0):
    """Stage: Linear [+ act + dropout_p]."""
    return {"W": weight, "b": bias, "act": act, "p": float(p), "ln": None}


def linear_add_norm(weight, bias, norm, res, p=0.0):
    """Stage: Linear -> LayerNorm `norm`(res + dropout_p(.)).  res: a tensor (R, 128) or ("tile", j) = this chain's t_j."""
    return {"W": weight, "b": bias, "act": None, "p": float(p), "ln": norm, "res": res}


def supported(x, stages, rows=None):
    """x: the chain's input (or any tensor of its device / dtype with `rows` = the row count and x.shape[-1] = the input width)."""
    if not (ENABLED and mfma_linear.BF16_MMA and x.is_cuda and x.dtype == torch.float32
            and not torch.is_autocast_enabled("cuda")):
        return False
    R = x.numel() // x.shape[-1] if rows is None else int(rows)
    if R % 64 or not 1 <= len(stages) <= 6:
        return False
    K = x.shape[-1]
    for s, st in enumerate(stages):
        N, Kw = st["W"].shape
        last = s + 1 == len(stages)
        if Kw != K or K not in (128, 256) or N % 128 or N > (384 if last else 256) or R * N >= 2 ** 32:
            return False
        if st["W"].dtype != torch.float32 or not mfma_linear.shape_supported(R, K, N):
            return False
        if st["ln"] is not None:
            ln = st["ln"]
            if N != 128 or tuple(ln.normalized_shape) != (128,) or not ln.elementwise_affine or ln.bias is None:
                return False
        K = N
    return True


_T0 = 6  # position of the first of `tensors` among _Chain.forward's inputs


class _Chain(Function):
    @staticmethod
    def forward(ctx, spec, seed, X, x_rows, last_rows, compact_acts, *tensors):
        # spec: per stage (iW, ib, act_kind, act_p, act_call, has_ln, res, ig, ibeta, ln_p, ln_call, eps); indices into
        # `tensors`; res = ("ext", index) | ("tile", j)
        # x_rows (bf16, or None): the VALUES of the input rows — X is then an fp32 shell autograd routes the gradient through,
        # its storage never read (an attention core's bf16 output, fused_attention.sdpa_rows).  last_rows: the last stage (a
        # plain projection) stores bf16 rows, returned as an extra non-differentiable output; its fp32 tile is a shell.
        # compact_acts: an activation stage that is not the last one and whose output is nobody's residual keeps what backward
        # reads and nothing else — its output h as bf16 rows (the values the next stage and the weight gradient multiply; the
        # returned tile is that bf16 tensor, non-differentiable) and, for ReLU, NO pre-activation (h > 0 <=> z > 0 wherever the
        # dropout mask kept the element): 1.5 KB less per row and 256-wide stage, the launch is bound by its stores.
        R = X.shape[0]
        res_tiles = {st[6][1] for st in spec if st[5] and st[6][0] == "tile"}
        nondiff = []
        dev = X.device
        if x_rows is not None:
            X = x_rows
        tiles, zs, xhats, rstds, descs = [X], [], [], [], []
        out_rows = None
        for si, (iW, ib, act_kind, act_p, act_call, has_ln, res, ig, ibeta, ln_p, ln_call, eps) in enumerate(spec):
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
                if compact_acts and si + 1 < len(spec) and (si + 1) not in res_tiles:
                    out = torch.empty((R, N), dtype=torch.bfloat16, device=dev)
                    nondiff.append(out)
                    d.update(h_out=out, h_out_bf16=1)
                    if act_kind != 0:  # GELU: its derivative needs z itself
                        z = torch.empty((R, N), dtype=torch.float32, device=dev)
                        d.update(v_out=z)
                else:
                    z = torch.empty((R, N), dtype=torch.float32, device=dev)
                    d.update(v_out=z, h_out=out)
            elif last_rows and si + 1 == len(spec):
                out_rows = torch.empty((R, N), dtype=torch.bfloat16, device=dev)
                d.update(v_out=out_rows, v_out_bf16=1)
            else:
                d.update(v_out=out)
            descs.append(d)
            tiles.append(out)
            zs.append(z)
            xhats.append(xh)
            rstds.append(rs)
        _ext.rows_chain(X, descs, seed)
        # the transposed weights the one-launch backward multiplies with: the K-major copies of the step's PreparedWeights
        # (refreshed once per forward pass; a weight seen for the first time is served from the next pass on — this pass then
        # walks the unfused entry points), or a transposed copy made here when no PreparedWeights context is open
        ctx.wt = None
        ctx.prep = ctx.prep_pass = None
        if FUSED_BACKWARD:
            prep = row_mlp._ACTIVE
            if prep is not None:
                ctx.prep, ctx.prep_pass = prep, prep.pass_id
            ctx.wt = [(prep.lookup(tensors[st[0]]) if prep is not None else tensors[st[0]].detach().t().contiguous())
                      if max(tensors[st[0]].shape) <= 256 else None for st in spec]
        ctx.spec = spec
        ctx.n_tensors = len(tensors)
        keep = [seed, X] + list(tensors) + tiles[1:] + [t for t in zs + xhats + rstds if t is not None]
        ctx.layout = ([t is not None for t in zs], [t is not None for t in xhats])
        ctx.save_for_backward(*keep)
        ctx.set_materialize_grads(False)
        if out_rows is not None:
            nondiff.append(out_rows)
        if nondiff:
            ctx.mark_non_differentiable(*nondiff)
        if out_rows is not None:
            return tuple(tiles[1:]) + (out_rows,)
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
        pend = [None] + [None if d is None else d.reshape(R, -1).contiguous() for d in douts[:n]]

        def accumulate(slot, g):
            return g if slot is None else slot + g

        def unfused_stage(s):
            g = pend[s + 1]
            if g is None:
                return
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
                # (compact stash of a ReLU stage: h stands in for z — h > 0 <=> z > 0 where the mask kept the element)
                zz = zs[s] if zs[s] is not None else tiles[s + 1].float()
                _ext.call("vlp3d_act_dropout", zz, g, g.numel(), act_kind, act_p, seed, act_call, gz, None)
                g = gz
            want_db = ib is not None and ctx.needs_input_grad[_T0 + ib]
            if ctx.needs_input_grad[_T0 + iW]:
                dw, db = mfma_linear.weight_grad(g, tiles[s], N, K, want_db, 1)
                grads[iW] = dw
                if want_db:
                    grads[ib] = db
            if s > 0 or ctx.needs_input_grad[2]:
                dx = torch.empty((R, K), dtype=torch.float32, device=g.device)
                _ext.call("vlp3d_linear_dgrad", g, W, R, N, K, dx, pend[s], 1)  # base: the gradient already waiting for t_s
                pend[s] = dx

        def chainable(top):
            """stages top..0 as ONE backward launch: transposed weights at hand, widths the tile kernel stages, residuals
            that are tiles of the chain above t_0."""
            if ctx.wt is None or pend[top + 1] is None or not ctx.needs_input_grad[2]:
                return False
            if ctx.prep is not None and not ctx.prep.current(ctx.prep_pass):
                return False   # the K-major copies were refreshed by a later forward pass: walk the unfused stages (W itself)
            for s in range(top + 1):
                st = spec[s]
                N, K = tensors[st[0]].shape
                if ctx.wt[s] is None or N > 256 or K > 256 or (st[5] and st[6][0] == "tile" and not 1 <= st[6][1] <= s):
                    return False
            return True

        def fused_run(top):
            G = pend[top + 1]
            dev = G.device
            points, gemms, gouts, lns = [], [], {}, []
            nblk = _ext.rows_chain_bwd_blocks(R)
            kept_for = {}  # tile index -> an add & norm above keeps its residual gradient for it
            for j, s in enumerate(range(top, -1, -1)):
                (iW, ib, act_kind, act_p, act_call, has, res, ig, ibeta, ln_p, ln_call, eps) = spec[s]
                N, K = tensors[iW].shape
                P = {"base": pend[s + 1] if j > 0 else None, "add_kept": int(kept_for.pop(s + 1, False))}
                if has:
                    part = torch.empty((nblk, 2, N), dtype=torch.float32, device=dev)
                    g = torch.empty((R, N), dtype=torch.float32, device=dev)
                    P.update(op=1, aux=xhats[s], rstd=rstds[s], gamma=tensors[ig], p=ln_p, call=ln_call, g_out=g, part=part)
                    if res[0] == "ext":
                        dres = torch.empty((R, N), dtype=torch.float32, device=dev)
                        P["dres_out"] = dres
                        grads[res[1]] = accumulate(grads[res[1]], dres)
                    else:
                        P["keep"] = 1
                        kept_for[res[1]] = True
                    lns.append((s, part))
                elif act_kind >= 0:
                    g = torch.empty((R, N), dtype=torch.float32, device=dev)
                    P.update(op=2, aux=zs[s], act_kind=act_kind, p=act_p, call=act_call, g_out=g)
                    if zs[s] is None:  # compact stash (ReLU): the stage's bf16 output in place of its pre-activation
                        P.update(aux=tiles[s + 1], aux_bf16=1)
                elif j == 0:
                    g = G  # the incoming gradient is what this stage's weight gradient reads
                else:
                    g = torch.empty((R, N), dtype=torch.float32, device=dev)
                    P.update(op=0, g_out=g)
                gouts[s] = g
                points.append(P)
                gemms.append({"Wt": ctx.wt[s], "N": K, "K": N})
            gX = torch.empty((R, tensors[spec[0][0]].shape[1]), dtype=torch.float32, device=dev)
            points.append({"base": pend[0], "add_kept": int(kept_for.pop(0, False)), "op": 0, "g_out": gX})
            assert not kept_for
            _ext.rows_chain_bwd(G, points, gemms, seed)
            for s, part in lns:
                ig, ibeta = spec[s][7], spec[s][8]
                N = part.shape[2]
                dgb = torch.empty((2, N), dtype=torch.float32, device=dev)
                _ext.reduce_slabs(part, nblk, dgb, 2 * N, 2 * N, 2 * N)
                grads[ig], grads[ibeta] = dgb[0], dgb[1]
            for s in range(top, -1, -1):
                iW, ib = spec[s][0], spec[s][1]
                N, K = tensors[iW].shape
                want_db = ib is not None and ctx.needs_input_grad[_T0 + ib]
                if ctx.needs_input_grad[_T0 + iW]:
                    dw, db = mfma_linear.weight_grad(gouts[s], tiles[s], N, K, want_db, 1)
                    grads[iW] = dw
                    if want_db:
                        grads[ib] = db
            pend[0] = gX

        s = n - 1
        while s >= 0 and not chainable(s):
            unfused_stage(s)
            s -= 1
        if s >= 0:
            fused_run(s)
        gX = pend[0]
        for i, t in enumerate(tensors):  # residual tensors keep the caller's shape
            if grads[i] is not None and grads[i].shape != t.shape:
                grads[i] = grads[i].view(t.shape)
        return (None, None, gX, None, None, None) + tuple(grads)


def run(x, stages, training=True, x_rows=None, last_rows=False, compact_acts=False):
    """x (..., K0) -> tuple of t_1..t_n, each (R, N_s) with R = rows of x.  Check `supported(x, stages)` first.
    x_rows: the input's values as bf16 rows (x is then an fp32 shell, see _Chain.forward); last_rows: t_n's values come back
    as bf16 rows in an extra last element of the tuple (t_n itself is then a shell).  compact_acts: see _Chain.forward — only
    for callers that do not differentiate through (or read as fp32) the outputs of the activation stages in the middle."""
    X = x.reshape(-1, x.shape[-1]).contiguous()
    R = X.shape[0]
    if x_rows is not None:
        x_rows = x_rows.reshape(R, -1).contiguous()
        if x_rows.dtype != torch.bfloat16 or x_rows.shape != X.shape:
            raise RuntimeError("row_chain.run: x_rows must be the bf16 rows of x")
    if last_rows and (stages[-1]["ln"] is not None or stages[-1]["act"] is not None):
        raise RuntimeError("row_chain.run: last_rows needs a plain projection as the last stage")
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
    return _Chain.apply(tuple(spec), add_norm.state(X.device), X, x_rows, bool(last_rows), bool(compact_acts), *tensors)
