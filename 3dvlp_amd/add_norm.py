"""out = LayerNorm(x + dropout(y)) — the add & norm closing every attention / FFN block
(models/transformer/attention.py:128-130, mmattention.py:84-86) — on the fused HIP kernels of csrc/add_norm.hip.

The dropout mask is a counter-based hash of (seed word, call id, element index) recomputed in backward, never stored.
`seed` is one int64 in device memory per device: `advance(device)` adds 1 (the step driver does that once per step,
inside the replayed graph, so replays draw fresh masks); `call_id` is a process-wide counter that distinguishes call
sites (baked into a captured graph, ever increasing in eager mode).
"""
import torch
from torch.autograd import Function

from . import _lib as _ext

_ext.load()

_STATE = {}
_CALLS = [0]
_DIMS = (64, 128, 256)


def state(device):
    """The device's dropout seed word (created from torch's default generator, so torch.manual_seed controls it)."""
    key = torch.device(device)
    if key.index is None:
        key = torch.device(key.type, torch.cuda.current_device())
    if key not in _STATE:
        _STATE[key] = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).to(key)
    return _STATE[key]


def advance(device):
    state(device).add_(1)


def next_call():
    """A fresh call id for one dropout site (process-wide counter; baked into a captured graph)."""
    _CALLS[0] = (_CALLS[0] + 1) & 0xFFFFF
    return _CALLS[0]


def norm_backward(d2, r2, xhat, rstd, kappa, gamma, R, D, p, seed, call_id, has_y):
    """Backward of LayerNorm(x + dropout_p(y)) (vlp3d_sum_norm_bwd): d2 (R, D) = gradient of the normalised rows, r2 = gradient
    arriving at the sum directly (or None).  Returns (dx, dy or None, dgamma, dbeta); [dgamma | dbeta] is summed now, or with
    the other slabs of the backward pass when a deferred queue is open."""
    dx = torch.empty_like(d2)
    dy = torch.empty_like(d2) if has_y else None
    nblk = int(_ext.load().vlp3d_add_norm_blocks(R))
    part = torch.empty((nblk, 2, D), dtype=torch.float32, device=d2.device)
    dgb = torch.empty((2, D), dtype=torch.float32, device=d2.device)
    q = _ext.slab_queue()
    _ext.call("vlp3d_sum_norm_bwd", d2, r2, xhat, rstd, kappa, gamma.contiguous(), R, D, p, seed, call_id, dx, dy, part,
              dgb, int(q is not None))
    if q is not None:
        q.add(part, nblk, dgb, 2 * D, 2 * D, 2 * D)
    return dx, dy, dgb[0], dgb[1]


def supported(x, y, norm):
    D = x.shape[-1]
    return (x.is_cuda and x.dtype == torch.float32 and y.dtype == torch.float32 and x.shape == y.shape and D in _DIMS
            and tuple(norm.normalized_shape) == (D,) and norm.elementwise_affine and norm.bias is not None
            and x.numel() < 2 ** 32)


class _AddNorm(Function):
    @staticmethod
    def forward(ctx, x, y, gamma, beta, p, eps, call_id, seed, mask):
        D = x.shape[-1]
        x2, y2 = x.reshape(-1, D).contiguous(), y.reshape(-1, D).contiguous()
        R = x2.shape[0]
        out, xhat = torch.empty_like(x2), torch.empty_like(x2)
        rstd = torch.empty((R,), dtype=torch.float32, device=x.device)
        _ext.call("vlp3d_add_norm_fwd", x2, y2, gamma.contiguous(), beta.contiguous(), R, D, float(p), seed, call_id,
                  float(eps), out, xhat, rstd, mask)
        ctx.save_for_backward(xhat, rstd, gamma, seed)
        ctx.cfg = (R, D, float(p), call_id, x.shape)
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, dout):
        xhat, rstd, gamma, seed = ctx.saved_tensors
        R, D, p, call_id, shape = ctx.cfg
        dx, dy, dgam, dbet = norm_backward(dout.reshape(R, D).contiguous(), None, xhat, rstd, None, gamma, R, D, p, seed, call_id,
                                           True)
        return dx.view(shape), dy.view(shape), dgam, dbet, None, None, None, None, None


def add_norm(x, y, norm, p=0.0, training=True, mask_out=None):
    """LayerNorm `norm` of x + dropout_p(y).  mask_out: optional uint8 tensor like x receiving the keep mask (tests)."""
    p = float(p) if training else 0.0
    _CALLS[0] = (_CALLS[0] + 1) & 0xFFFFF
    return _AddNorm.apply(x, y, norm.weight, norm.bias, p, norm.eps, _CALLS[0], state(x.device), mask_out)


class _AddNormRep(Function):
    """LayerNorm(x + dropout(y)) on `rep` replicas of every group of `seq` rows of x / y (csrc/add_norm.hip
    vlp3d_add_norm_rep_fwd): (G*seq, D) -> (G*rep*seq, D); backward = the plain add & norm backward on the replicated rows,
    then ONE launch sums the replicas' dx and dy back onto the source rows."""

    @staticmethod
    def forward(ctx, x, y, gamma, beta, rep, seq, p, eps, call_id, seed):
        D = x.shape[-1]
        x2, y2 = x.reshape(-1, D).contiguous(), y.reshape(-1, D).contiguous()
        Rs = x2.shape[0]
        R = Rs * rep
        out = torch.empty((R, D), dtype=torch.float32, device=x.device)
        xhat = torch.empty_like(out)
        rstd = torch.empty((R,), dtype=torch.float32, device=x.device)
        _ext.call("vlp3d_add_norm_rep_fwd", x2, y2, gamma.contiguous(), beta.contiguous(), R, D, int(rep), int(seq), float(p),
                  seed, call_id, float(eps), out, xhat, rstd)
        ctx.save_for_backward(xhat, rstd, gamma, seed)
        ctx.cfg = (R, Rs, D, float(p), call_id, int(rep), int(seq), x.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        xhat, rstd, gamma, seed = ctx.saved_tensors
        R, Rs, D, p, call_id, rep, seq, shape = ctx.cfg
        d2 = dout.reshape(R, D).contiguous()
        dx, dy = torch.empty_like(d2), torch.empty_like(d2)
        nblk = int(_ext.load().vlp3d_add_norm_blocks(R))
        part = torch.empty((nblk, 2, D), dtype=torch.float32, device=dout.device)
        dgb = torch.empty((2, D), dtype=torch.float32, device=dout.device)
        q = _ext.slab_queue()
        _ext.call("vlp3d_sum_norm_bwd", d2, None, xhat, rstd, None, gamma.contiguous(), R, D, p, seed, call_id, dx, dy, part,
                  dgb, int(q is not None))
        if q is not None:
            q.add(part, nblk, dgb, 2 * D, 2 * D, 2 * D)
        sx = torch.empty((Rs, D), dtype=torch.float32, device=dout.device)
        sy = torch.empty_like(sx)
        _ext.call("vlp3d_rep_sum2", dx, dy, Rs, D, rep, seq, sx, sy)
        return sx.view(shape), sy.view(shape), dgb[0], dgb[1], None, None, None, None, None, None


def add_norm_rep(x, y, norm, rep, p=0.0, training=True):
    """x, y (G, seq, D) -> (G*rep, seq, D): every group replicated `rep` times, then LayerNorm(x + dropout_p(y)) with an
    independent dropout mask per replica (== add_norm(x.repeat_interleave(rep, 0), y.repeat_interleave(rep, 0), ...))."""
    p = float(p) if training else 0.0
    G, seq, D = x.shape
    _CALLS[0] = (_CALLS[0] + 1) & 0xFFFFF
    out = _AddNormRep.apply(x, y, norm.weight, norm.bias, int(rep), int(seq), p, norm.eps, _CALLS[0], state(x.device))
    return out.view(G * rep, seq, D)


class _SumNorm(Function):
    """(s, n) = (x + dropout_p(y), norm(s)) — the pre-norm residual stream of the caption decoder (csrc/add_norm.hip
    vlp3d_sum_norm_*).  y None: s = x (first norm of a stack; s is then x itself, no copy)."""

    @staticmethod
    def forward(ctx, x, y, gamma, beta, p, eps, std_mode, call_id, seed, mask):
        D = x.shape[-1]
        x2 = x.reshape(-1, D).contiguous()
        y2 = None if y is None else y.reshape(-1, D).contiguous()
        R = x2.shape[0]
        out, xhat = torch.empty_like(x2), torch.empty_like(x2)
        ssum = None if y is None else torch.empty_like(x2)
        rstd = torch.empty((R,), dtype=torch.float32, device=x.device)
        kappa = torch.empty((R,), dtype=torch.float32, device=x.device)
        _ext.call("vlp3d_sum_norm_fwd", x2, y2, gamma.contiguous(), beta.contiguous(), R, D, float(p), seed, call_id,
                  float(eps), int(std_mode), ssum, out, xhat, rstd, kappa, mask)
        ctx.save_for_backward(xhat, rstd, kappa, gamma, seed)
        ctx.set_materialize_grads(False)  # unused outputs arrive as None (the kernels take NULL), not as zero fills
        ctx.cfg = (R, D, float(p), call_id, x.shape, y is not None)
        if y is None:
            return x.view_as(x), out.view(x.shape)
        return ssum.view(x.shape), out.view(x.shape)

    @staticmethod
    def backward(ctx, dsum, dout):
        xhat, rstd, kappa, gamma, seed = ctx.saved_tensors
        R, D, p, call_id, shape, has_y = ctx.cfg
        if dout is None:
            dout = torch.zeros(shape, dtype=torch.float32, device=xhat.device)
        d2 = dout.reshape(R, D).contiguous()
        r2 = None if dsum is None else dsum.reshape(R, D).contiguous()
        dx = torch.empty_like(d2)
        dy = torch.empty_like(d2) if has_y else None
        nblk = int(_ext.load().vlp3d_add_norm_blocks(R))
        part = torch.empty((nblk, 2, D), dtype=torch.float32, device=d2.device)
        dgb = torch.empty((2, D), dtype=torch.float32, device=d2.device)
        q = _ext.slab_queue()
        _ext.call("vlp3d_sum_norm_bwd", d2, r2, xhat, rstd, kappa, gamma.contiguous(), R, D, p, seed, call_id, dx, dy, part,
                  dgb, int(q is not None))
        if q is not None:
            q.add(part, nblk, dgb, 2 * D, 2 * D, 2 * D)
        return dx.view(shape), (dy.view(shape) if has_y else None), dgb[0], dgb[1], None, None, None, None, None, None


def sum_norm_supported(x):
    return x.is_cuda and x.dtype == torch.float32 and x.shape[-1] in _DIMS and x.numel() < 2 ** 32


def sum_norm(x, y, gamma, beta, eps, p=0.0, training=True, std_mode=True, mask_out=None):
    """Returns (s, n): s = x + dropout_p(y) (y may be None), n = norm(s) with scale gamma / shift beta.
    std_mode=True: the captioner's LayerNorm (transformer_captioner.py:117-129); False: nn.LayerNorm."""
    p = float(p) if (training and y is not None) else 0.0
    _CALLS[0] = (_CALLS[0] + 1) & 0xFFFFF
    return _SumNorm.apply(x, y, gamma, beta, p, eps, std_mode, _CALLS[0], state(x.device), mask_out)


class _ActDropout(Function):
    @staticmethod
    def forward(ctx, z, kind, p, call_id, seed, mask):
        z2 = z.contiguous()
        out = torch.empty_like(z2)
        _ext.call("vlp3d_act_dropout", z2, None, z2.numel(), kind, float(p), seed, call_id, out, mask)
        ctx.save_for_backward(z2, seed)
        ctx.cfg = (kind, float(p), call_id)
        return out

    @staticmethod
    def backward(ctx, dout):
        z2, seed = ctx.saved_tensors
        kind, p, call_id = ctx.cfg
        dz = torch.empty_like(z2)
        _ext.call("vlp3d_act_dropout", z2, dout.contiguous(), z2.numel(), kind, p, seed, call_id, dz, None)
        return dz, None, None, None, None, None


def act_dropout_supported(z):
    return z.is_cuda and z.dtype == torch.float32 and z.numel() % 4 == 0 and 4 <= z.numel() < 2 ** 32


def act_dropout(z, kind, p=0.0, training=True, mask_out=None):
    """dropout_p(act(z)), act = "relu" | "gelu" (erf) — one launch each way (csrc/add_norm.hip), mask never stored."""
    p = float(p) if training else 0.0
    _CALLS[0] = (_CALLS[0] + 1) & 0xFFFFF
    return _ActDropout.apply(z, {"relu": 0, "gelu": 1}[kind], p, _CALLS[0], state(z.device), mask_out)
