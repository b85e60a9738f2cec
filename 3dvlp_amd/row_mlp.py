"""1x1 Conv(+bias) -> BatchNorm -> ReLU stacks on point-major rows (csrc/rows_mlp.hip), exact fp32.

`row_stack(X, layers)` runs X (R, K0) through `layers` = [(weight, bias, bn), ...]:
  * every layer but possibly the last has a BatchNorm (`bn` = the nn.BatchNorm1d/2d module, train or eval mode) followed by
    ReLU; its conv bias (if any) cancels inside a train-mode BatchNorm and only moves running_mean;
  * the last layer may be a plain biased linear layer (`bn` None): its output is returned raw.
This is the arithmetic of SharedMLP / Conv1d+BatchNorm1d+ReLU chains in PointnetFPModule (pointnet2_modules.py:403-416),
VotingModule (voting_module.py:33-60) and StandardROIHeads (roi_heads.py:15-147), on (rows, channels) matrices instead of
(B, C, n[, 1]) tensors: no NCHW convolution, layout transposes, library BatchNorm or separate ReLU launches, and the
autograd backward is the same handful of kernels run in reverse (BatchNorm / ReLU backward folded into the products).
`fp_rows` is three_interpolate + concat of the FP module on point-major features.
"""
import os
import torch
from torch.autograd import Function

from . import _lib as _ext
from . import ddp, mfma_linear

_ext.load()
_ZEROS = {}
ROWS_WGRAD_BLOCKS = int(os.environ.get("VLP3D_ROWS_WGRAD_BLOCKS", 64))
BATCH_WGRAD = os.environ.get("VLP3D_ROWS_WGRAD_BATCH", "1") != "0"  # queue the weight gradients (vlp3d_rows_wgrad_batch)
WGRAD_K = 256  # K-slice of one weight-gradient launch (the staging of csrc/sa_mlp.hip: wgrad_kernel covers K <= 288)


_ACTIVE = None  # the PreparedWeights whose forward pass is running


class PreparedWeights:
    """K-major copies of the weights the rows stacks of ONE model use, refreshed by one launch at the start of every forward
    pass (vlp3d_transpose_batch) — `with prepared:` around the model's forward.  The forward product then reads its weight
    fragments coalesced (vlp3d_rows_fwd_wt) instead of re-staging a 128 x K weight block through LDS for every 32-row tile:
    30 -> 13 us for the 8192 x 256 x 256 layers of the FP / voting stacks.  A weight is registered the first time a stack
    sees it (that pass runs the row-major kernel) and served from the next pass on; outside the context nothing is served,
    so a stale copy can never be read.  The parameters keep the reference's (N, K[, 1]) storage and state_dict layout."""

    def __init__(self):
        self.entries = {}   # (data_ptr, shape) -> (weight view (Np, K), K-major copy (K, Np))
        self.fresh = set()  # registered since the last refresh: not served yet
        self.pass_id = 0    # number of refreshes: the copies handed out belong to THIS pass

    def __enter__(self):
        global _ACTIVE
        if self.entries:
            _ext.transpose_batch([t for _, t in self.entries.values()], [w for w, _ in self.entries.values()])
        self.fresh.clear()
        self.pass_id += 1
        self._outer, _ACTIVE = _ACTIVE, self
        return self

    def current(self, pass_id):
        """True while the copies served in pass `pass_id` have not been overwritten by a later refresh: a backward that kept
        them (row_chain: ctx.wt) must check this before multiplying — after another forward pass under this context (gradient
        accumulation, an interleaved eval forward after a parameter update) they hold the transposes of the NEW weights."""
        return pass_id == self.pass_id

    def __exit__(self, *exc):
        global _ACTIVE
        _ACTIVE = self._outer
        return False

    def lookup(self, w):
        """w: contiguous (Np, K) view of a parameter's storage -> its K-major copy (K, Np), or None (then registered)."""
        key = (w.data_ptr(), tuple(w.shape))   # two views of one address with different shapes are two entries
        e = self.entries.get(key)
        if e is None:
            self.entries[key] = (w.detach(), torch.zeros((w.shape[1], w.shape[0]), dtype=torch.float32, device=w.device))
            self.fresh.add(key)
            return None
        return None if key in self.fresh else e[1]


def _zeros(n, device):
    key = (n, str(device))
    if key not in _ZEROS:
        _ZEROS[key] = torch.zeros(n, dtype=torch.float32, device=device)
    return _ZEROS[key]


def _round_up(x, m):
    return (x + m - 1) // m * m


def supported(x, layers):
    """fp32 CUDA rows, R % 32 == 0, every BatchNorm layer 64 / 128 / 256 / 512 / 1024 wide, input widths the weight-gradient
    kernel slices."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] % 32 == 0 and x.shape[0] >= 32):
        return False
    k = x.shape[1]
    for i, (w, b, bn) in enumerate(layers):
        n = w.shape[0]
        if w.shape[1] != k or not (k in (32, 64, 128, 256) or (k > WGRAD_K and k % WGRAD_K == 0)):
            return False
        if bn is None:
            if i != len(layers) - 1:
                return False
        elif n not in (64, 128, 256, 512, 1024) or not bn.affine:  # n / 4 divides 256: the BatchNorm loaders' staging
            return False
        k = n
    return not torch.is_autocast_enabled("cuda")


class _RowStack(Function):
    @staticmethod
    def forward(ctx, x, bns, keep_pad, slope, *params):
        L = len(bns)
        W, bias, gam, bet = params[0::4], params[1::4], params[2::4], params[3::4]
        R, dev = x.shape[0], x.device
        x = x.contiguous()
        lib = _ext.load()
        bf = int(mfma_linear.BF16_MMA)  # bf16 MFMA operands: the step driver's timing configuration (mfma_linear.bf16_mma)
        nslab = int(lib.vlp3d_rows_slabs(R))
        Ys, vecs, Wp = [], [], []
        A, lda, a_vec = x, x.shape[1], None
        training = [bn is not None and (bn.training or not bn.track_running_stats) for bn in bns]
        for l in range(L):
            N, K = W[l].shape
            Np = _round_up(N, 64)
            w = W[l].contiguous()
            b = bias[l]
            if Np != N:  # only a final plain layer (259 = 3 + 256 vote channels, 28 ROI predictor channels)
                if ddp.padded_rows(w) >= Np and (b is None or ddp.padded_rows(b) >= Np):
                    # the parameter storage itself carries the zero rows (ddp.FlatParams reserved them): no per-step pad
                    w = w.as_strided((Np, K), (K, 1))
                    b = None if b is None else b.as_strided((Np,), (1,))
                else:
                    w = torch.nn.functional.pad(w, (0, 0, 0, Np - N))
                    b = None if b is None else torch.nn.functional.pad(b, (0, Np - N))
            Wp.append(w)
            y = torch.empty((R, Np), dtype=torch.float32, device=dev)
            bn = bns[l]
            # K-major copy of the weight made at the start of this forward pass (PreparedWeights), when the step driver runs one
            wt = _ACTIVE.lookup(w) if (_ACTIVE is not None and w.data_ptr() == W[l].data_ptr()) else None

            def product(bias_, stats_):
                if wt is not None:
                    _ext.call("vlp3d_rows_fwd_wt", A, lda, R, K, a_vec, wt, Np, bias_, Np, y, Np, stats_, bf)
                else:
                    _ext.call("vlp3d_rows_fwd", A, lda, R, K, a_vec, w, bias_, Np, y, Np, stats_, bf)
            if bn is None:
                product(b, None)
                vec = None
            else:
                vec = torch.empty((4, Np), dtype=torch.float32, device=dev)
                if training[l]:
                    st = torch.empty((nslab, 2, Np), dtype=torch.float64, device=dev)
                    product(None, st)
                    track = bn.track_running_stats and bn.training
                    if track:
                        if bn.num_batches_tracked is not None:  # None: the step driver increments all counters at once
                            bn.num_batches_tracked.add_(1)
                        mom = 1.0 / float(bn.num_batches_tracked) if bn.momentum is None else bn.momentum
                    else:
                        mom = 0.0
                    # (the bias shifts the batch mean the running estimate tracks, and nothing else: folded into the same launch)
                    _ext.call("vlp3d_sa_bn_fold_shift", st, nslab, gam[l], bet[l], bn.running_mean if track else None,
                              bn.running_var if track else None, Np, R, float(bn.eps), float(mom), 1, vec,
                              b.detach() if (track and b is not None) else None)
                else:  # eval: y includes the bias, the running statistics normalise it
                    product(b, None)
                    _ext.call("vlp3d_sa_bn_fold", None, 1, gam[l], bet[l], bn.running_mean, bn.running_var, Np, R,
                              float(bn.eps), 0.0, 0, vec)
            Ys.append(y)
            vecs.append(vec)
            A, lda, a_vec = y, Np, vec
        if bns[-1] is not None:
            out = torch.empty_like(Ys[-1])
            if slope is not None and slope.numel() != Ys[-1].shape[1]:
                raise ValueError("final PReLU slope needs one value per (64-aligned) output channel")
            _ext.call("vlp3d_rows_act", Ys[-1], R, Ys[-1].shape[1], vecs[-1], slope, out)
        else:
            out = Ys[-1] if keep_pad else Ys[-1][:, :W[-1].shape[0]]
        ctx.save_for_backward(x, *Ys, *[v for v in vecs if v is not None], *Wp, *[g for g in gam if g is not None],
                              *([slope] if slope is not None else []))
        ctx.meta = (L, [v is not None for v in vecs], training, [b is not None for b in bias],
                    [tuple(w.shape) for w in W], bf, slope is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        L, has_bn, training, has_bias, wshapes, bf, has_slope = ctx.meta
        sv = list(ctx.saved_tensors)
        slope = sv.pop() if has_slope else None
        dslope = None
        x, Ys = sv[0], sv[1:1 + L]
        nbn = sum(has_bn)
        vlist, Wp, glist = sv[1 + L:1 + L + nbn], sv[1 + L + nbn:1 + 2 * L + nbn], sv[1 + 2 * L + nbn:]
        vecs, gam, it_v, it_g = [], [], iter(vlist), iter(glist)
        for l in range(L):
            vecs.append(next(it_v) if has_bn[l] else None)
            gam.append(next(it_g) if has_bn[l] else None)
        R, dev = x.shape[0], x.device
        lib = _ext.load()
        nslab = int(lib.vlp3d_rows_slabs(R))
        grads = [None] * (4 * L)
        last = L - 1
        Np = Ys[last].shape[1]
        if has_bn[last]:
            G = torch.empty((R, Np), dtype=torch.float32, device=dev)
            tn = int(lib.vlp3d_rows_act_slabs(R))
            t = torch.empty((tn, 2, Np), dtype=torch.float64, device=dev)
            ds = torch.empty((tn, Np), dtype=torch.float64, device=dev) if has_slope else None
            _ext.call("vlp3d_rows_act_bwd", dout.contiguous(), Ys[last], R, Np, vecs[last], slope, G, t, ds)
            if has_slope:
                dslope = ds.sum(0).float()
        else:
            N = wshapes[last][0]
            G = dout if dout.shape[1] == Np else torch.nn.functional.pad(dout, (0, Np - N))
            G = G.contiguous()
            t, tn = None, 0
        dx = None
        for l in range(L - 1, -1, -1):
            N, K = wshapes[l]
            Np = Ys[l].shape[1]
            bn5 = None
            if has_bn[l]:
                bn5 = torch.empty((5, Np), dtype=torch.float32, device=dev)
                dg = torch.empty((Np,), dtype=torch.float32, device=dev)
                db = torch.empty((Np,), dtype=torch.float32, device=dev)
                _ext.call("vlp3d_sa_bn_bwd_consts", vecs[l], gam[l], t, tn, Np, R, int(training[l]), bn5, dg, db)
                grads[4 * l + 2], grads[4 * l + 3] = dg, db
                if has_bias[l]:  # cancels inside a train-mode BatchNorm; in eval mode it is d(beta) scaled by gamma*rstd
                    grads[4 * l + 1] = _zeros(N, dev) if training[l] else vecs[l][0] * db
            A = x if l == 0 else Ys[l - 1]
            lda = A.shape[1]
            pv = None if l == 0 else vecs[l - 1]
            dW = torch.empty((Np, K), dtype=torch.float32, device=dev)
            want_db = (not has_bn[l]) and has_bias[l]
            dbias = torch.empty((Np,), dtype=torch.float32, device=dev) if want_db else None
            ks = min(K, WGRAD_K)
            nblk = max(8, min(ROWS_WGRAD_BLOCKS, R // 64))
            q = _ext.slab_queue()
            for off in range(0, K, ks):
                if q is not None or off == 0:  # deferred: every K-slice keeps its own slabs until the batched sum
                    part = torch.empty((nblk, Np * ks + Np), dtype=torch.float32, device=dev)
                db_ = dbias if off == 0 else None
                if q is not None and bf and BATCH_WGRAD and Np <= 512:
                    # not launched now: the rows-stack and linear weight gradients of the whole backward pass run as a few
                    # launches when the queue is flushed (vlp3d_rows_wgrad_batch)
                    q.add_rows_wgrad(dict(G=G, Ypre=Ys[l] if has_bn[l] else None, ldg=Np, bn5=bn5, X=(A, off), lda=lda,
                                          a_scale=None if pv is None else (pv, off),
                                          a_shift=None if pv is None else (pv, pv.shape[1] + off), R=R, K=ks, N=Np,
                                          partials=part, max_blocks=nblk, with_bias=int(db_ is not None)),
                                     (dW, dbias), (part, _ext.wgrad_slabs(R, nblk), dW[:, off:], Np * ks, ks, K, db_,
                                                   Np if db_ is not None else 0))
                    continue
                _ext.call("vlp3d_rows_wgrad", G, Ys[l] if has_bn[l] else None, Np, bn5, A[:, off:], lda,
                          None if pv is None else pv[0, off:], None if pv is None else pv[1, off:], R, ks, Np,
                          dW[:, off:], K, db_, part, nblk, int(q is not None), bf)
                if q is not None:
                    q.add(part, _ext.wgrad_slabs(R, nblk), dW[:, off:], Np * ks, ks, K, db_, Np if db_ is not None else 0)
            grads[4 * l] = dW[:N]
            if want_db:
                grads[4 * l + 1] = dbias[:N]
            if l > 0:
                Gp = torch.empty((R, K), dtype=torch.float32, device=dev)
                tp = torch.empty((nslab, 2, K), dtype=torch.float64, device=dev)
                _ext.call("vlp3d_rows_dgrad", G, Ys[l] if has_bn[l] else None, Np, bn5, Wp[l], R, Np, K, Ys[l - 1], K,
                          vecs[l - 1], Gp, K, tp, bf)
                G, t, tn = Gp, tp, nslab
            elif ctx.needs_input_grad[0]:
                dx = torch.empty((R, K), dtype=torch.float32, device=dev)
                _ext.call("vlp3d_rows_dgrad", G, Ys[l] if has_bn[l] else None, Np, bn5, Wp[l], R, Np, K, None, 0, None,
                          dx, K, None, bf)
        return (dx, None, None, dslope, *grads)


def row_stack(x, layers, keep_pad=False, final_slope=None):
    """x (R, K0) fp32 CUDA; layers = [(weight (N,K[,1[,1]]), bias or None, bn module or None), ...] -> (R, N_last).
    final_slope: per-channel PReLU weight (N_last,) replacing the ReLU after the LAST BatchNorm layer.
    keep_pad: return the final plain layer's full 64-aligned buffer (R, round_up(N, 64)); the columns past N are zero
    and take no gradient — lets a consumer kernel address column blocks without a slice copy either way."""
    bns = [bn for _, _, bn in layers]
    params = []
    for w, b, bn in layers:
        params += [w.reshape(w.shape[0], w.shape[1]), b, None if bn is None else bn.weight, None if bn is None else bn.bias]
    return _RowStack.apply(x, bns, bool(keep_pad), final_slope, *params)


class _FPRows(Function):
    """[three_interpolate(known, idx, weight) | unknown] on point-major features: (B,m,C1), (B,n,C2) -> (B*n, C1+C2)."""

    @staticmethod
    def forward(ctx, known, unknown, idx, weight, inv):
        known, unknown = known.contiguous().float(), unknown.contiguous().float()
        B, m, C1 = known.shape
        n, C2 = unknown.shape[1:]
        X = torch.empty((B * n, C1 + C2), dtype=torch.float32, device=known.device)
        _ext.call("vlp3d_fp_rows", known, unknown, idx.contiguous(), weight.contiguous(), B, n, m, C1, C2, X)
        ctx.save_for_backward(idx, weight, *(inv if inv is not None else ()))
        ctx.dims = (B, n, m, C1, C2)
        return X

    @staticmethod
    def backward(ctx, dX):
        idx, weight = ctx.saved_tensors[:2]
        inv = ctx.saved_tensors[2:]
        B, n, m, C1, C2 = ctx.dims
        dX = dX.contiguous()
        dk = du = None
        if ctx.needs_input_grad[0]:
            dk = torch.empty((B, m, C1), dtype=torch.float32, device=dX.device)
            if len(inv) == 2:  # inverse of the three_nn map (built with the geometry): a wave per known point, no atomics
                _ext.call("vlp3d_fp_rows_grad_csr", dX, weight.contiguous(), inv[0], inv[1], B, m, C1, C1 + C2, dk)
            else:
                _ext.call("vlp3d_fp_rows_grad", dX, idx.contiguous(), weight.contiguous(), B, n, m, C1, C1 + C2, dk)
        if ctx.needs_input_grad[1]:
            du = dX.view(B, n, C1 + C2)[:, :, C1:]
        return dk, du, None, None, None


def fp_rows_supported(known_pm, unknown_pm):
    return (known_pm.is_cuda and known_pm.shape[2] % 16 == 0 and unknown_pm.shape[2] % 4 == 0
            and known_pm.shape[1] <= 1024)


def fp_rows(known_pm, unknown_pm, idx, weight, inv=None):
    """inv: (inv_start, inv_refs) = _lib.sa_inverse(idx (B, n, 3), m) — the atomic-free backward (PointnetFPModule.compute_geometry)."""
    return _FPRows.apply(known_pm, unknown_pm, idx, weight, inv)
