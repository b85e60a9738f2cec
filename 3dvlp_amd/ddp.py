"""Scene-level data parallelism: one process per GPU, ONE flat gradient all-reduce per step.

The reference only has single-process nn.DataParallel (scripts/joint_scripts/train_3dvlp.py:124-126).
Here every rank runs its own shard of scenes; gradients live as views into a single pre-zeroed flat
fp32 buffer (so parameters that receive no gradient in a step — many do, SURVEY.md §5 — contribute
zeros without any bookkeeping) and are summed with one RCCL all-reduce over xGMI (≈6 M elements =
24 MB: latency-bound, so one bucket instead of many).  BatchNorm statistics stay per rank, exactly
like the reference's DataParallel replicas.
"""
import torch
import torch.distributed as dist


class FlatGradBucket:
    """Flat fp32 gradient buffer; after `collect()` every parameter's .grad is a view into it.

    Autograd adds into an existing .grad with one small kernel per parameter (~300 launches per step here), so
    the gradients are left undefined during backward (autograd then just keeps the produced tensors) and are
    gathered afterwards with one multi-tensor copy.  Parameters that received no gradient keep zeros."""

    def __init__(self, module, process_group=None):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.group = process_group
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=torch.float32, device=ref.device)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
            p.grad = self.views[-1]

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
        dst, src = [], []
        self.touched = []
        for p, v in zip(self.params, self.views):
            got = p.grad is not None
            if got and p.grad.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(p.grad.to(torch.float32) if p.grad.dtype != torch.float32 else p.grad)
            p.grad = v if got else None
            self.touched.append(got)
        if dst:
            torch._foreach_copy_(dst, src)

    def all_reduce(self):
        """Sum over ranks then average. No-op without an initialised process group."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(dist.get_world_size(self.group))


def broadcast_parameters(module, src=0, process_group=None):
    """One-time parameter + buffer broadcast from rank `src` so that all replicas start identical."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=process_group)


def shard_range(num_scenes, rank, world_size):
    """Contiguous shard of scene ids for this rank (global batch = per-rank batch * world_size)."""
    if num_scenes % world_size:
        raise ValueError("shard_range: %d scenes do not divide over %d ranks (the step assumes equal shards: "
                         "per-rank BatchNorm statistics and a plain 1/world gradient average)" % (num_scenes, world_size))
    per = num_scenes // world_size
    return rank * per, (rank + 1) * per
