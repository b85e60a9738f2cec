"""nn.Linear on the hand-written MFMA kernels (csrc/sa_mlp.hip: row_gemm<PLAIN,BIAS>, wgrad<PLAIN>).

The projections / FFNs / heads of the grounding path are 2 048..16 384-row by 128..256 GEMMs that the BLAS
library runs at ~5 TFLOP/s; here they use the same exact-fp32 MFMA kernels as the grouped MLP.  `linear(x, w, b)`
has F.linear semantics; shapes the kernels do not cover (rows not a multiple of 32, tiny widths) go to F.linear.
"""
import collections
import os
import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib as _ext

_ext.load()

BF16_MMA = False       # timing configuration (set by the step driver together with the bf16 grouped MLPs): bf16 MFMA operands
_ROWS_PER_BLOCK = 64   # rows a workgroup of the weight-gradient kernel accumulates before writing its slab
WGRAD_BLOCKS = int(os.environ.get("VLP3D_LINEAR_WGRAD_BLOCKS", 128))     # at most this many workgroups (= partial [dW | db] slabs)
BATCH_WGRAD = os.environ.get("VLP3D_LINEAR_WGRAD_BATCH", "1") != "0"  # queue the weight gradients, one launch for all
BATCH_WGRAD_BLOCKS = int(os.environ.get("VLP3D_LINEAR_WGRAD_BATCH_BLOCKS", 32))  # row groups per layer inside a batch
_FWD_N = (32, 64, 128, 160, 256, 288)
_WGRAD_N = (64, 128, 256, 384, 512)  # > 256: 128-column workgroup blocks (merged q/k/v projections)
_WGRAD_K = (32, 64, 128, 256)  # K/4 a power of two (the weight-gradient staging indexes rows by shifts)
_SLICED_K = (512,)             # wider inputs (the captioner's w_2, transformer_captioner.py:95-104): the weight gradient as 256-column slices


# Library (rocBLAS / hipBLASLt) GEMMs taken for shapes the MFMA kernels do not cover, keyed "R x K -> N".  Nothing falls back
# silently: bench.py prints the number per step (0 in the cfg2 step), tests assert on it.
FALLBACKS = collections.Counter()


def note_fallback(rows, K, N, where="linear"):
    FALLBACKS["%s: %d x %d -> %d" % (where, rows, K, N)] += 1


def supported(x, weight):
    R = x.numel() // x.shape[-1]
    N, K = weight.shape
    return x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and shape_supported(R, K, N)


def shape_supported(R, K, N):
    """R rows of K columns through a weight (N, K): shapes the forward, input-gradient and weight-gradient kernels all cover."""
    if K in _SLICED_K:  # forward / input gradient take any K % 32 == 0; the weight gradient runs per 256-column slice of X
        return N % 64 == 0 and N <= 512 and shape_supported(R, 256, N)
    if not (R % 32 == 0 and R >= 32 and K % 8 == 0 and N in _WGRAD_N and K in _WGRAD_K):
        return False
    kt = (K + 31) // 32
    blocked = N > 256 or (N // 32) * kt > 36  # the weight gradient then runs as 128-column workgroup blocks
    nb = 128 if blocked else N
    return (not blocked or N % 128 == 0) and (nb // 32) * kt <= 36 and 32 * (nb + kt * 32) * 4 <= 65536


def weight_grad(dy2, x2, N, K, want_db, bf):
    """(dW (N, K), db (N) or None) of y = x W^T + b from dy2 (R, N), x2 (R, K): launched now, or queued when a deferred
    slab-reduce queue is open (one batched launch for all queued layers at the end of backward)."""
    R = x2.shape[0]
    q = _ext.slab_queue()
    if K in _SLICED_K:
        return _weight_grad_sliced(dy2, x2, N, K, want_db, bf, q)
    batched = q is not None and bf and BATCH_WGRAD and N % 64 == 0 and N <= 512 and K <= 256
    if x2.dtype == torch.bfloat16 and not batched:
        x2 = x2.float()  # bf16 rows (an attention core's output) are an operand form of the batched kernel only
    # dW and (when asked for) the bias gradient come out of ONE kernel pair: [dW | db] contiguous
    dwb = torch.empty((N * K + (N if want_db else 0),), dtype=torch.float32, device=dy2.device)
    nblk = max(16, min(WGRAD_BLOCKS, R // _ROWS_PER_BLOCK))  # few slabs for few rows: the slab sum reads nblk*N*K floats
    if batched:  # the batch supplies the parallelism: fewer, longer row groups per layer = a quarter of the slab traffic
        nblk = max(8, min(BATCH_WGRAD_BLOCKS, R // _ROWS_PER_BLOCK))
    part = torch.empty((nblk, dwb.numel()), dtype=torch.float32, device=dy2.device)
    if batched:
        # not launched now: up to 48 of these run as ONE launch when the queue is flushed (end of backward)
        q.add_linear_wgrad(dy2, x2, part, R, K, N, nblk, want_db, _ext.wgrad_slabs(R, nblk), dwb,
                           dwb[N * K:] if want_db else None)
    else:
        _ext.call("vlp3d_linear_wgrad", dy2, x2, R, K, N, dwb, part, nblk, int(want_db), int(q is not None), int(bf))
        if q is not None:
            q.add(part, _ext.wgrad_slabs(R, nblk), dwb, N * K, K, K, dwb[N * K:] if want_db else None,
                  N if want_db else 0)
    return dwb[:N * K].view(N, K), (dwb[N * K:] if want_db else None)


def _weight_grad_sliced(dy2, x2, N, K, want_db, bf, q):
    """dW (N, K) for K > 256 as K / 256 jobs of the batched rows weight gradient (vlp3d_rows_wgrad_batch: X + offset with row
    stride K, dW[:, offset:] with row stride K) when the step's deferred queue is open; outside it (eager unit tests) a counted
    library product."""
    R = x2.shape[0]
    dev = dy2.device
    if q is None or not (bf and BATCH_WGRAD):
        note_fallback(R, K, N, "linear weight gradient")
        return dy2.t() @ x2.float(), (dy2.sum(0) if want_db else None)
    dW = torch.empty((N, K), dtype=torch.float32, device=dev)
    db = torch.empty((N,), dtype=torch.float32, device=dev) if want_db else None
    x2 = x2.float() if x2.dtype != torch.float32 else x2
    ks = 256
    nblk = max(8, min(BATCH_WGRAD_BLOCKS, R // _ROWS_PER_BLOCK))
    for off in range(0, K, ks):
        part = torch.empty((nblk, N * ks + N), dtype=torch.float32, device=dev)
        db_ = db if off == 0 else None
        q.add_rows_wgrad(dict(G=dy2, Ypre=None, ldg=N, bn5=None, X=(x2, off), lda=K, a_scale=None, a_shift=None, R=R, K=ks, N=N,
                              partials=part, max_blocks=nblk, with_bias=int(db_ is not None)),
                         (dW, db), (part, _ext.wgrad_slabs(R, nblk), dW[:, off:], N * ks, ks, K, db_, N if db_ is not None else 0))
    return dW, db


class _Linear(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, bf16_mma, with_residual=False):
        ctx.bf = int(bool(bf16_mma))
        ctx.with_residual = bool(with_residual)
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        R, K = x2.shape
        N = weight.shape[0]
        w = weight.contiguous()
        y = torch.empty((R, N), dtype=torch.float32, device=x.device)
        _ext.call("vlp3d_linear_fwd", x2, w, bias, R, K, N, y, ctx.bf)
        ctx.save_for_backward(x2, w)
        ctx.has_bias = bias is not None
        ctx.xshape = x.shape
        if with_residual:  # x handed back as a second output: the caller routes its residual connection through it, and
            # backward receives both gradients of x together (the add is then fused into the dX kernel)
            return y.view(*x.shape[:-1], N), x.view_as(x)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy, dres=None):
        x2, w = ctx.saved_tensors
        R, K = x2.shape
        N = w.shape[0]
        dy2 = dy.reshape(R, N).contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((R, K), dtype=torch.float32, device=dy.device)
            base = None
            if dres is not None and ctx.bf and K % 32 == 0 and N % 16 == 0:
                base = dres.reshape(R, K).contiguous()
            if K % 32 == 0:
                _ext.call("vlp3d_linear_dgrad", dy2, w, R, N, K, dx, base, ctx.bf)  # reads W (N,K) as stored: no transposed copy
            else:
                _ext.call("vlp3d_linear_fwd", dy2, w.t().contiguous(), None, R, N, K, dx, ctx.bf)
            dx = dx.view(ctx.xshape)
            if dres is not None and base is None:
                dx = dx + dres
        elif dres is not None:
            dx = dres
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            dw, db = weight_grad(dy2, x2, N, K, want_db, ctx.bf)
        elif want_db:
            db = dy2.sum(0)
        return dx, dw, db, None, None


class _LinearRows16(Function):
    """linear(x, w, b, bf16_mma=True, with_residual=True) whose result is STORED as bf16 rows (vlp3d_linear_fwd_rows16): the
    query projection in front of an attention core that reads bf16 rows (fused_attention.sdpa_rows).  Returns (y shell — an
    fp32 tensor autograd routes the gradient through, storage untouched —, x_res, y rows bf16 (non-differentiable))."""

    @staticmethod
    def forward(ctx, x, weight, bias, with_residual=True):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        R, K = x2.shape
        N = weight.shape[0]
        w = weight.contiguous()
        rows = _ext.linear_fwd_rows16(x2, w, bias)
        ctx.save_for_backward(x2, w)
        ctx.has_bias = bias is not None
        ctx.xshape = x.shape
        ctx.with_residual = bool(with_residual)
        rows = rows.view(*x.shape[:-1], N)
        ctx.mark_non_differentiable(rows)
        ctx.set_materialize_grads(False)  # (no zero tensor for the rows output's absent gradient)
        shell = torch.empty(rows.shape, dtype=torch.float32, device=x.device)
        return (shell, x.view_as(x), rows) if with_residual else (shell, rows)

    @staticmethod
    def backward(ctx, dy, *rest):
        dres = rest[0] if ctx.with_residual else None
        return _LinearRows16._backward(ctx, dy, dres) + (None,)

    @staticmethod
    def _backward(ctx, dy, dres):
        if dy is None:
            return dres, None, None
        x2, w = ctx.saved_tensors
        R, K = x2.shape
        N = w.shape[0]
        dy2 = dy.reshape(R, N).contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((R, K), dtype=torch.float32, device=dy.device)
            base = None if dres is None else dres.reshape(R, K).contiguous()
            _ext.call("vlp3d_linear_dgrad", dy2, w, R, N, K, dx, base, 1)
            dx = dx.view(ctx.xshape)
        elif dres is not None:
            dx = dres
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            dw, db = weight_grad(dy2, x2, N, K, want_db, 1)
        elif want_db:
            db = dy2.sum(0)
        return dx, dw, db


def rows16_supported(x, weight):
    R = x.numel() // x.shape[-1]
    N, K = weight.shape
    return supported(x, weight) and K % 32 == 0 and N % 64 == 0 and R % 32 == 0 and not torch.is_autocast_enabled("cuda")


def linear_rows16(x, weight, bias=None, with_residual=True):
    """-> (y shell, x_res, y rows bf16), or (y shell, y rows bf16) without the residual route; see _LinearRows16.  Check
    rows16_supported() first."""
    return _LinearRows16.apply(x, weight, bias, with_residual)


class bf16_mma:
    """Context: linear() calls inside use bf16 MFMA operands (forward; each call's backward follows its forward)."""

    def __init__(self, enabled=True):
        self.enabled = bool(enabled)

    def __enter__(self):
        global BF16_MMA
        self._old, BF16_MMA = BF16_MMA, self.enabled

    def __exit__(self, *exc):
        global BF16_MMA
        BF16_MMA = self._old
        return False


def linear(x, weight, bias=None, bf16_mma=None, with_residual=False):
    """F.linear on the MFMA kernels.  bf16_mma=None: the module default BF16_MMA.
    with_residual=True returns (y, x_res): x_res is x, to be used for the residual connection around the layer — backward
    then receives both gradients of x in one place and adds them inside the dX kernel (bf16 configuration)."""
    if supported(x, weight) and not torch.is_autocast_enabled("cuda"):
        return _Linear.apply(x, weight, bias, BF16_MMA if bf16_mma is None else bf16_mma, with_residual)
    if x.is_cuda:
        note_fallback(x.numel() // x.shape[-1], weight.shape[1], weight.shape[0])
    if with_residual:
        return F.linear(x, weight, bias), x
    return F.linear(x, weight, bias)
