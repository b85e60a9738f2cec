"""Scene-level data parallelism: one process per GPU, ONE flat gradient all-reduce per step.

The reference only has single-process nn.DataParallel (scripts/joint_scripts/train_3dvlp.py:124-126).
Here every rank runs its own shard of scenes; gradients live as views into a single pre-zeroed flat
fp32 buffer (so parameters that receive no gradient in a step — many do, SURVEY.md §5 — contribute
zeros without any bookkeeping) and are summed with one RCCL all-reduce over xGMI (~1.8 M elements =
7 MB: latency-bound, so one bucket instead of many).  BatchNorm statistics stay per rank, exactly
like the reference's DataParallel replicas.

`FlatParams` additionally moves the PARAMETERS themselves into one flat buffer (same element order as the gradient
buffer): projections that are evaluated as one merged product (q|k|v, k|v, the five ROI predictors) then read their
concatenated weight as a VIEW (`merge_adjacent`, no torch.cat per step), and AdamW is one launch over the flat buffers
(`FlatAdamW`, csrc/glue.hip) instead of the multi-tensor optimiser's six.
"""
import math
import weakref

import torch
import torch.distributed as dist
from torch.autograd import Function


def adjacency_groups(module):
    """Parameter groups that should sit back to back in the flat buffer (each group in concatenation order)."""
    groups = []
    for m in module.modules():
        name = type(m).__name__
        if name == "ScaledDotProductAttention":
            groups.append([m.fc_q.weight, m.fc_k.weight, m.fc_v.weight])
            groups.append([m.fc_q.bias, m.fc_k.bias, m.fc_v.bias])
        elif name == "StandardROIHeads":
            heads = [m.heading_reg_predictor, m.heading_cls_predictor, m.box_predictor, m.objectness_predictor]
            if m.num_class:
                heads.append(m.sem_cls_predictor)
            groups.append([h.weight for h in heads])
            groups.append([h.bias for h in heads])
        elif name == "RelationModule":  # the packed parameter block of each pairwise-bias MLP (csrc/relation_bias.hip)
            for fc in m.self_attn_fc:
                groups.append(list(fc.parameters()))
    return groups


def padding_requests(module):
    """[(first parameter, last parameter, row multiple)]: parameter runs (one tensor, or an adjacency group from first to
    last) whose LEADING dimension a kernel wants rounded up to a multiple — FlatParams then reserves zero rows behind the
    run, so that the kernel reads a padded weight straight from the parameter storage instead of F.pad-ing it every step
    (the final plain layers of the voting module: 259 -> 320 rows, and of the merged ROI predictors: 28 -> 64)."""
    req = []
    for m in module.modules():
        name = type(m).__name__
        if name == "VotingModule":
            req += [(m.conv3.weight, m.conv3.weight, 64), (m.conv3.bias, m.conv3.bias, 64)]
        elif name == "StandardROIHeads":
            heads = [m.heading_reg_predictor, m.heading_cls_predictor, m.box_predictor, m.objectness_predictor]
            if m.num_class:
                heads.append(m.sem_cls_predictor)
            req += [(heads[0].weight, heads[-1].weight, 64), (heads[0].bias, heads[-1].bias, 64)]
    return req


PADDED_ROWS = {}  # data_ptr of a run's first parameter -> (rows available incl. the reserved zero rows, weakref to the flat buffer)


def padded_rows(t):
    """Rows readable at t's address when t starts a padded parameter run of a LIVE FlatParams buffer (else 0).  The entry
    is trusted only while the flat buffer it was made for is alive and t really lives in that buffer's storage — an
    address alone says nothing once a model has been freed."""
    hit = PADDED_ROWS.get(t.data_ptr())
    if hit is None:
        return 0
    rows, ref = hit
    flat = ref()
    if flat is None or t.untyped_storage().data_ptr() != flat.untyped_storage().data_ptr():
        return 0
    return rows


def ordered_parameters(module):
    """module.parameters() (trainable ones) reordered so that every adjacency group is contiguous."""
    placed, out = set(), []
    group_of = {}
    for g in adjacency_groups(module):
        for p in g:
            group_of[id(p)] = g
    for p in module.parameters():
        if not p.requires_grad or id(p) in placed:
            continue
        for q in group_of.get(id(p), [p]):
            if id(q) not in placed:
                placed.add(id(q))
                out.append(q)
    return out


class FlatParams:
    """Re-homes the parameters of `module` (already on its device) in one flat fp32 buffer, in `ordered_parameters`
    order; every group start is 16-byte aligned, members of an adjacency group are packed without gaps.  The
    parameters keep their names, shapes and values (state_dict / load_state_dict work as before)."""

    def __init__(self, module):
        self.params = ordered_parameters(module)
        starts = {id(g[0]) for g in adjacency_groups(module)}
        members = {id(p) for g in adjacency_groups(module) for p in g}
        pos = {id(p): i for i, p in enumerate(self.params)}
        pad_after, run_first = {}, {}
        for first, last, mult in padding_requests(module):
            if id(first) not in pos or id(last) not in pos:
                continue
            run = self.params[pos[id(first)]:pos[id(last)] + 1]
            rows = sum(q.shape[0] for q in run)
            row_elems = first.numel() // first.shape[0]
            want = (rows + mult - 1) // mult * mult
            pad_after[id(last)] = (want - rows) * row_elems
            run_first[id(first)] = want
        offs, off = [], 0
        for p in self.params:
            if id(p) in starts or id(p) not in members:
                off = (off + 3) // 4 * 4
            offs.append(off)
            off += p.numel() + pad_after.get(id(p), 0)   # reserved rows stay zero: no gradient, masked out of AdamW
        self.numel = (off + 3) // 4 * 4
        ref = self.params[0]
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=ref.device)
        self.offsets = offs
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                view = self.flat[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                if id(p) in run_first:
                    PADDED_ROWS[p.data_ptr()] = (run_first[id(p)], weakref.ref(self.flat))


class _MergeAdjacent(Function):
    """cat(tensors, 0) of tensors that ARE back to back in memory: a view forward, views of the gradient backward."""

    @staticmethod
    def forward(ctx, *ts):
        ctx.rows = [t.shape[0] for t in ts]
        first = ts[0]
        shape = (sum(ctx.rows),) + tuple(first.shape[1:])
        return first.as_strided(shape, first.stride(), first.storage_offset())

    @staticmethod
    def backward(ctx, g):
        return tuple(torch.split(g, ctx.rows, dim=0))


def merge_adjacent(tensors):
    """torch.cat(tensors, 0) without a copy when the tensors are contiguous and adjacent in one storage (FlatParams
    arranges that for the merged projections); plain torch.cat otherwise."""
    ok = all(t.is_contiguous() and t.dtype == tensors[0].dtype for t in tensors)
    if ok:
        st = tensors[0].untyped_storage().data_ptr()
        off = tensors[0].storage_offset()
        for t in tensors:
            if t.untyped_storage().data_ptr() != st or t.storage_offset() != off or t.shape[1:] != tensors[0].shape[1:]:
                ok = False
                break
            off += t.numel()
    return _MergeAdjacent.apply(*tensors) if ok else torch.cat(tensors, 0)


class FlatGradBucket:
    """Flat fp32 gradient buffer; after `collect()` every parameter's .grad is a view into it.

    Autograd adds into an existing .grad with one small kernel per parameter (~300 launches per step here), so
    the gradients are left undefined during backward (autograd then just keeps the produced tensors) and are
    gathered afterwards with one multi-tensor copy.  Parameters that received no gradient keep zeros.
    `layout`: a FlatParams whose element order the buffer follows (then flat gradient i belongs to flat parameter i)."""

    def __init__(self, module, process_group=None, layout=None):
        self.group = process_group
        if layout is not None:
            self.params, offs, n = layout.params, layout.offsets, layout.numel
        else:
            self.params = [p for p in module.parameters() if p.requires_grad]
            offs, n = [], 0
            for p in self.params:
                offs.append(n)
                n += p.numel()
        self.offsets = offs
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=torch.float32, device=ref.device)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for p, o in zip(self.params, offs)]
        for p, v in zip(self.params, self.views):
            p.grad = v
        self.touched = [True] * len(self.params)

    def zero(self):
        """Call before backward: zero the flat buffer and detach the .grad views from the parameters."""
        self.flat.zero_()
        for p in self.params:
            p.grad = None

    def collect(self):
        """Call after backward: copy the produced gradients into the flat buffer, re-attach the views.

        A parameter that received NO gradient in this backward keeps ``.grad = None`` (its slice of the flat buffer
        stays zero, so the all-reduce is unaffected): the optimiser then skips it exactly like the reference's
        (torch.optim skips grad-None parameters) — no weight decay on, and no optimiser state for, sub-modules that
        exist only so that reference checkpoints load (lang_emb_proj, box_con_proj, NCELoss.tau, ...).  All ranks run
        the same graph, so the touched set is the same everywhere."""
        self._collect(None)

    def collect_subset(self, params):
        """collect() for some of the parameters only (the launch goes to the CURRENT stream): the step driver's split
        backward completes the gradients of two disjoint parameter sets on two streams, and each set is copied into its
        slices of the flat buffer on the stream that produced it.  Every parameter must be covered by exactly one call."""
        self._collect({id(p) for p in params})

    def _collect(self, only):
        dst, src = [], []
        if only is None or len(getattr(self, "touched", ())) != len(self.params):
            self.touched = [False] * len(self.params)
        for i, (p, v) in enumerate(zip(self.params, self.views)):
            if only is not None and id(p) not in only:
                continue
            got = p.grad is not None
            if got and p.grad.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(p.grad.to(torch.float32) if p.grad.dtype != torch.float32 else p.grad)
            p.grad = v if got else None
            self.touched[i] = got
        if dst:
            torch._foreach_copy_(dst, src)

    def _distributed(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def _use_avg(self):
        """RCCL (backend "nccl"): ReduceOp.AVG — the division rides inside the all-reduce, no second launch.  Probed ONCE on a
        one-element tensor (no recorded run had executed that branch; a build that rejects AVG must not fail the
        first step): on any error, and on gloo (no AVG), SUM followed by one in-place division."""
        if getattr(self, "_avg_ok", None) is None:
            ok = False
            if dist.get_backend(self.group) == "nccl":
                try:
                    probe = torch.ones(1, dtype=torch.float32, device=self.flat.device)
                    dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=self.group)
                    ok = bool(abs(float(probe.item()) - 1.0) < 1e-6)
                except Exception:  # noqa: BLE001 — whatever the backend raises for an unsupported op
                    ok = False
            self._avg_ok = ok
        return self._avg_ok

    def all_reduce(self):
        """Average over ranks with ONE collective on the flat buffer.  No-op without an initialised process group."""
        if self._distributed():
            self.all_reduce_range(0, self.flat.numel()).wait()

    def param_range(self, params):
        """[a, b) of the flat buffer covered by `params` if they are exactly one contiguous run of it, else None."""
        ids = {id(p) for p in params}
        idx = [i for i, p in enumerate(self.params) if id(p) in ids]
        if not idx or len(idx) != len(ids) or idx != list(range(idx[0], idx[-1] + 1)):
            return None
        a = self.offsets[idx[0]]
        b = self.offsets[idx[-1] + 1] if idx[-1] + 1 < len(self.params) else self.flat.numel()
        return a, b

    def all_reduce_range(self, a, b):
        """Start the averaging all-reduce of flat[a:b] and return a handle whose wait() completes it (for the CURRENT stream:
        with RCCL the collective runs on the process group's own stream, ordered behind the stream that was current when it was
        issued — issue it under `torch.cuda.stream(s)` right after the kernels that complete those gradients and it overlaps
        whatever the other streams still do).  The step driver reduces the head parameters' slice as soon as the deferred
        graph has completed it, beside the backward of SA2 / SA1, and the rest afterwards: same sums as ONE all-reduce of the
        whole buffer (tests/test_ddp_gloo.py).  Without a process group: a no-op handle."""
        if not self._distributed() or b <= a:
            return _Done()
        piece = self.flat[a:b]
        if self._use_avg():
            return _Pending(dist.all_reduce(piece, op=dist.ReduceOp.AVG, group=self.group, async_op=True), None, 1)
        return _Pending(dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True), piece,
                        dist.get_world_size(self.group))


class _Done:
    def wait(self):
        return None


class _Pending:
    """An all-reduce in flight (+ the division that completes the average where the backend has no AVG)."""

    def __init__(self, work, piece, world):
        self.work, self.piece, self.world = work, piece, world

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
            if self.piece is not None:
                self.piece.div_(self.world)


class FlatAdamW:
    """torch.optim.AdamW over (FlatParams, FlatGradBucket with the same layout), one launch per run of parameters that share
    a step count (csrc/glue.hip) — ONE launch in the steady state.  Like torch.optim.AdamW the step count that enters the
    bias correction is PER PARAMETER and starts when the parameter first receives a gradient (a parameter that becomes
    active later — the contrast projections at epoch 50, a restart with a different touched set — gets its own count, and
    the launch splits at the boundaries).  `state_dict()` / `load_state_dict()` carry m, v and the step counts."""

    def __init__(self, layout, bucket, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        assert bucket.flat.numel() == layout.flat.numel()
        self.layout, self.bucket = layout, bucket
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.m = torch.zeros_like(layout.flat)
        self.v = torch.zeros_like(layout.flat)
        self.active = torch.zeros(layout.numel, dtype=torch.uint8, device=layout.flat.device)
        self._active_key = None
        self.steps = [0] * len(layout.params)   # per parameter, like torch.optim's state[p]["step"]
        self._runs = None

    @property
    def t(self):
        """Largest step count (the only one in the steady state)."""
        return max(self.steps) if self.steps else 0

    def _set_active(self):
        key = tuple(self.bucket.touched)
        if key != self._active_key:  # changes only when the step's graph changes (e.g. the contrast losses switch on)
            a = torch.zeros(self.layout.numel, dtype=torch.uint8)
            for p, o, got in zip(self.layout.params, self.layout.offsets, self.bucket.touched):
                if got:
                    a[o:o + p.numel()] = 1
            self.active.copy_(a)
            self._active_key = key
            self._runs = None

    def _step_runs(self):
        """[(start, end, index of one touched parameter of the run)] over the flat buffer: maximal runs of consecutive
        touched parameters with one step count (untouched parameters are masked by `active` and join a neighbouring run).
        All touched parameters advance together, so the partition only changes with the touched set."""
        runs = []
        for i, (p, o, got) in enumerate(zip(self.layout.params, self.layout.offsets, self.bucket.touched)):
            if not got:
                continue
            if runs and self.steps[runs[-1][2]] == self.steps[i]:
                runs[-1][1] = o + p.numel()
            else:
                runs.append([o, o + p.numel(), i])
        if runs:
            runs[0][0] = 0
            runs[-1][1] = self.layout.numel
            for a, b in zip(runs, runs[1:]):
                a[1] = b[0]
        return runs

    def step(self):
        from . import _lib as _ext
        self._set_active()
        if self._runs is None:
            self._runs = self._step_runs()
        for i, got in enumerate(self.bucket.touched):
            if got:
                self.steps[i] += 1
        b1, b2 = self.betas
        L, G = self.layout.flat, self.bucket.flat
        for a, b, i in self._runs:
            t = self.steps[i]
            _ext.call("vlp3d_adamw_flat", L[a:b], G[a:b], self.m[a:b], self.v[a:b], self.active[a:b], b - a,
                      float(self.lr), float(b1), float(b2), float(self.eps), float(self.weight_decay), 1.0 - b1 ** t,
                      math.sqrt(1.0 - b2 ** t))

    def state_dict(self):
        return {"m": self.m.clone(), "v": self.v.clone(), "steps": list(self.steps), "lr": self.lr, "betas": self.betas,
                "eps": self.eps, "weight_decay": self.weight_decay}

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.steps = list(sd["steps"])
        self.lr, self.betas, self.eps, self.weight_decay = sd["lr"], tuple(sd["betas"]), sd["eps"], sd["weight_decay"]
        self._runs = None


def _flat_broadcast(tensors, src, group):
    """One broadcast for a list of same-dtype tensors: pack, broadcast, unpack."""
    if not tensors:
        return
    flat = torch.cat([t.detach().reshape(-1) for t in tensors])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    with torch.no_grad():
        for t in tensors:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()


def broadcast_parameters(module, src=0, process_group=None, layout=None):
    """One-time parameter + buffer broadcast from rank `src` so that all replicas start identical: ONE collective for the
    parameters (the FlatParams buffer itself when `layout` is given — every parameter is a view of it — else a packed copy)
    and one per buffer dtype (BatchNorm statistics, counters) instead of one per tensor (~300 at this model)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return
    if layout is not None:
        dist.broadcast(layout.flat, src=src, group=process_group)
        rest = [p for p in module.parameters() if not p.requires_grad]  # (not re-homed by FlatParams)
    else:
        rest = list(module.parameters())
    by_dtype = {}
    for t in rest + list(module.buffers()):
        by_dtype.setdefault(t.dtype, []).append(t.data)
    for ts in by_dtype.values():
        _flat_broadcast(ts, src, process_group)


def shard_range(num_scenes, rank, world_size):
    """Contiguous shard of scene ids for this rank (global batch = per-rank batch * world_size)."""
    if num_scenes % world_size:
        raise ValueError("shard_range: %d scenes do not divide over %d ranks (the step assumes equal shards: "
                         "per-rank BatchNorm statistics and a plain 1/world gradient average)" % (num_scenes, world_size))
    per = num_scenes // world_size
    return rank * per, (rank + 1) * per
