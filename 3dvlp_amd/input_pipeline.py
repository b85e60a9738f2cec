"""Input pipeline of the grounding step — SURVEY.md §8f-4.

  sample_scene   lib/joint/dataset.py:603-612: optional height channel (z minus the 0.99th percentile of z), then
                 `num_points` point indices by rng.choice (with replacement only when the scene has fewer points), applied
                 to the cloud and to every per-point label array.
  Prefetcher     lib/joint/prefetcher.py:2-22, same interface (`Prefetcher(loader).next()` -> data_dict or None): the next
                 batch is uploaded on a copy stream while the current step runs.  Differences that are the point of having
                 it here: host batches are staged through PINNED buffers (reused, so `non_blocking=True` really overlaps —
                 the reference calls .cuda(non_blocking=True) on pageable tensors, which is synchronous), numpy arrays are
                 accepted as they come out of a collate function, and the batch-only derived tensors the kernels read
                 (grounding_step.prepare_batch: dtypes, decoded referred-box sizes, K/V token slice) are produced on the copy
                 stream too, off the step's critical path.
A 40 000-point cfg2 batch is 8 x 40 000 x 135 fp32 = 173 MB: ~3.5 ms of PCIe gen5 per step if not overlapped, against a
~9 ms step — overlapped it is free; bench.py --host-batches measures exactly that.
"""
import os

import numpy as np
import torch

from . import synth


def sample_scene(point_cloud, num_points, rng, use_height=True, per_point=()):
    """point_cloud (n, C) float array; per_point: label arrays of length n.  Returns (cloud (num_points, C[+1]), labels...)."""
    if use_height:
        floor_height = np.percentile(point_cloud[:, 2], 0.99)
        point_cloud = np.concatenate([point_cloud, (point_cloud[:, 2] - floor_height)[:, None]], 1)
    replace = point_cloud.shape[0] < num_points
    choices = rng.choice(point_cloud.shape[0], num_points, replace=replace)
    return (point_cloud[choices],) + tuple(a[choices] for a in per_point)


class Prefetcher:
    """Iterates `loader` (an iterable of dicts of numpy arrays / CPU tensors / plain Python values), keeping ONE batch in
    flight on a copy stream.  `prepare(batch_dict_on_device) -> batch_dict` runs on that stream after the upload."""

    def __init__(self, loader, device="cuda", prepare=None, stream=None, max_ahead=3):
        self.loader = iter(loader)
        self.device = torch.device(device)
        # The uploads never run more than `max_ahead` next() calls ahead of the consumer's STREAM: an asynchronous consumer (a
        # captured step: ~0.5 ms of host time per 4 ms of device time) otherwise lets the host race ahead until the launch queues
        # push back — ~60 batches of 176 MB in flight at cfg2, 11 GB of device memory that the allocator cannot reuse
        # (tools/soak_host_feed.py).  The wait is on the host, for the consumer's position `max_ahead` calls ago.
        self.max_ahead = max_ahead
        self._marks = []
        # (stream: reuse a copy stream created earlier — every new HIP stream takes one of the few hardware queues, and a
        # stream created late in a process can land on the queue of the step's main or side stream and serialise with it)
        self.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)
        self.prepare = prepare
        self._pinned = [{}, {}]  # two sets of staging buffers: the copy of batch t+1 may still read its set while t+2 is staged
        self._uploaded = [None, None]  # per set: event recorded after its H2D copies; waited for before the set is overwritten
        self._flip = 0
        self.data_dict = None
        self.preload()

    def _stage(self, key, value):
        t = torch.from_numpy(np.ascontiguousarray(value)) if isinstance(value, np.ndarray) else value
        if not torch.is_tensor(t):
            return value
        if t.is_pinned():  # e.g. DataLoader(pin_memory=True): already page-locked, upload straight from it
            return t
        pool = self._pinned[self._flip]
        buf = pool.get(key)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            pool[key] = buf
        buf.copy_(t)
        return buf

    PACK_BELOW = 1 << 20   # bytes: tensors smaller than this travel in one packed upload

    def _stage_all(self, host):
        """-> ({key: pinned tensor or plain value} for the large tensors, (pinned byte buffer, layout) for the small ones).
        A batch has ~45 tensors of which 3 carry 99 % of the bytes; 45 separate copies kept the command processor busy for
        ~0.15 ms per step that the launch stream's dispatches waited for (tools/host_feed_timeline.py)."""
        small, staged = [], {}
        for k, v in host.items():
            t = torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v
            if torch.is_tensor(t) and not t.is_cuda and t.numel() > 0 and t.numel() * t.element_size() < self.PACK_BELOW \
                    and os.environ.get("VLP3D_PREFETCH_PACK", "1") != "0":
                small.append((k, t.contiguous()))
            else:
                staged[k] = self._stage(k, v)
        if not small:
            return staged, None
        layout, total = [], 0
        for k, t in small:
            n = t.numel() * t.element_size()
            layout.append((k, total, n, t.dtype, tuple(t.shape)))
            total += (n + 255) & ~255
        pool = self._pinned[self._flip]
        buf = pool.get("__packed__")
        if buf is None or buf.numel() < total:
            buf = torch.empty((total,), dtype=torch.uint8, pin_memory=torch.cuda.is_available())
            pool["__packed__"] = buf
        for (k, t), (_, off, n, _, _) in zip(small, layout):
            buf[off:off + n].copy_(t.reshape(-1).view(torch.uint8))
        return staged, (buf[:total], layout)

    def preload(self):
        try:
            host = next(self.loader)
        except StopIteration:
            self.data_dict = None
            return
        used = self._flip
        if self._uploaded[used] is not None:
            self._uploaded[used].synchronize()  # the copies that read this staging set two batches ago have finished
        staged, packed = self._stage_all(host)
        self._flip ^= 1
        with torch.cuda.stream(self.stream):
            dev = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in staged.items()}
            if packed is not None:   # the small tensors: ONE upload, carved into views on the device
                buf, layout = packed
                dbuf = buf.to(self.device, non_blocking=True)
                for k, off, n, dtype, shape in layout:
                    dev[k] = dbuf[off:off + n].view(dtype).view(shape)
            ev = torch.cuda.Event()
            ev.record(self.stream)
            self._uploaded[used] = ev
            if self.prepare is not None:
                dev = self.prepare(dev)
            self._ready = torch.cuda.Event()
            self._ready.record(self.stream)
        self.data_dict = dev

    def next(self):
        """The prefetched batch (device tensors, safe to use on the current stream), or None when the loader is exhausted."""
        # order the consumer's stream behind the upload + prepare — unless they have already finished (no packet at all then)
        ready = getattr(self, "_ready", None)
        if ready is None:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
        elif not ready.query():
            torch.cuda.current_stream(self.device).wait_event(ready)
        data_dict = self.data_dict
        cur = torch.cuda.current_stream(self.device)
        if data_dict is not None:
            for v in data_dict.values():
                if torch.is_tensor(v):
                    v.record_stream(cur)
        if self.max_ahead:
            mark = torch.cuda.Event()
            mark.record(cur)           # everything the consumer has enqueued before asking for this batch
            self._marks.append(mark)
            if len(self._marks) > self.max_ahead:
                self._marks.pop(0).synchronize()
        self.preload()
        return data_dict


def compress_cloud(host_batch):
    """Host batch -> the same batch with `point_clouds` (B,N,3+C) fp32 replaced by `k/xyz` (B,N,3) fp32, `k/feat_bf`
    (B,N,round_up(C,8)) bf16 (zero padded) and `k/feat_c` = C: loader-side work (a DataLoader worker / collate_fn) that halves the bytes of the step's largest input on the PCIe
    link — at cfg2 the cloud is 173 MB of a batch's 176 MB, and the host-fed step is bound by that copy (bench.py
    ms_per_step_host_batches).  ONLY for the bf16 configuration: its first grouped-MLP layer rounds the gathered features to
    bf16 (round to nearest even, as here) before the matrix product, so the step computes the same bits; the exact-fp32
    parity configuration and the device-side augmentation (which rewrites the cloud) need the fp32 cloud.
    The bf16 kernels read k/feat_bf as it is (grounding_step.prepare_batch leaves it alone)."""
    pc = host_batch["point_clouds"]
    pc = torch.from_numpy(np.ascontiguousarray(pc)) if isinstance(pc, np.ndarray) else pc
    if pc.shape[-1] <= 3:
        return host_batch
    out = {k: v for k, v in host_batch.items() if k != "point_clouds"}
    C = pc.shape[-1] - 3
    out["k/xyz"] = pc[..., :3].contiguous()
    rows = torch.zeros(pc.shape[:-1] + ((C + 7) // 8 * 8,), dtype=torch.bfloat16)   # 16-byte rows for the gather kernels
    rows[..., :C] = pc[..., 3:]
    out["k/feat_bf"], out["k/feat_c"] = rows, C
    return out


# ---- training-time augmentation (lib/joint/dataset.py:653-690, utils/utils_fn.py:28-142) --------------------------------
def draw_augment_params(rng, batch_size):
    """The reference's random draws for `batch_size` scenes, in its call order per scene (flip x, flip y, three angles, the
    3x3 scale draw of which the diagonal is used, three translations): (B, 24) float32 — the layout csrc/augment.hip reads.
    Host work: 11 numbers and one 3x3 product per scene."""
    out = np.zeros((batch_size, 24), np.float64)
    grid = np.arange(-0.5, 0.501, 0.001)
    for b in range(batch_size):
        fx, fy = float(rng.random() > 0.7), float(rng.random() > 0.7)
        ang = [rng.random() * np.pi / 18 - np.pi / 36 for _ in range(3)]
        sc = np.exp(rng.uniform(-0.1, 0.1, (3, 3)))
        t = [rng.choice(grid, size=1)[0] for _ in range(3)]
        (cx, sx), (cy, sy), (cz, sz) = [(np.cos(a), np.sin(a)) for a in ang]
        rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        out[b, :11] = [fx, fy, *ang, sc[0, 0], sc[1, 1], sc[2, 2], *t]
        out[b, 12:21] = (rx.T @ ry.T @ rz.T).reshape(-1)
    return out.astype(np.float32)


@torch.no_grad()
def augment_on_device(batch, params, mean_size_arr=None, height_col=None):
    """Apply the reference's training-time augmentation to a DEVICE batch in place of the loader workers' numpy code.
    batch: the reference's keys + `instance_labels` (B,N) int32, `instance_valid` (B,I) uint8, `box_sizes` (B,M,3);
    params: (B,24) from draw_augment_params (host array or device tensor); height_col: column of point_clouds holding the
    height channel (negative: from the end; None: no height channel — dataset.py's use_height off).  Rewrites point_clouds (xyz + height column),
    vote_label, vote_label_mask, center_label, size_residual_label, ref_center_label_list, ref_size_residual_label_list —
    the votes from the AUGMENTED cloud's per-instance point boxes, the box labels from the augmented GT boxes
    (dataset.py:653-690).  Three launches over the cloud / boxes + a few label gathers; runs on the caller's stream (the
    Prefetcher's copy stream)."""
    from . import _lib as _ext
    pc = batch["point_clouds"]
    if not pc.is_cuda:
        raise RuntimeError("CPU not supported")
    dev = pc.device
    B, N, C = pc.shape
    p = torch.as_tensor(params, dtype=torch.float32, device=dev).contiguous()
    inst = batch["instance_labels"].to(torch.int32).contiguous()
    valid = batch["instance_valid"].to(torch.uint8).contiguous()
    I = valid.shape[1]
    ibox = torch.empty((B, I, 6), dtype=torch.int32, device=dev)
    if not pc.is_contiguous():
        pc = batch["point_clouds"] = pc.contiguous()
    hcol = -1 if height_col is None else int(height_col) % C
    _ext.call("vlp3d_augment_points", pc, B, N, C, hcol, p, inst, I, ibox)
    vote = torch.empty((B, N, 9), dtype=torch.float32, device=dev)
    vm = batch["vote_label_mask"]
    mask_i = torch.empty((B, N), dtype=torch.int64, device=dev) if not vm.is_floating_point() else None
    mask_f = torch.empty((B, N), dtype=torch.float32, device=dev) if vm.is_floating_point() else None
    _ext.call("vlp3d_augment_votes", pc, B, N, C, inst, I, ibox, valid, vote, mask_f, mask_i)
    batch["vote_label"], batch["vote_label_mask"] = vote, (mask_f if mask_i is None else mask_i)
    present = batch["box_label_mask"].to(torch.float32).unsqueeze(-1)
    # absent GT rows are ZERO boxes before the augmentation (dataset.py:631,649-650) and go through it like every other row
    boxes = (torch.cat([batch["center_label"][..., :3].float(), batch["box_sizes"].float()], -1) * present).contiguous()
    M = boxes.shape[1]
    out = torch.empty_like(boxes)
    _ext.call("vlp3d_augment_boxes", boxes, B, M, p, out)
    mean = torch.as_tensor(synth.mean_size_arr() if mean_size_arr is None else mean_size_arr, dtype=torch.float32, device=dev)
    # dataset.py:823 exports target_bboxes[:, 0:3] UNMASKED: an absent row's centre is the translation vector (utils_fn.py:
    # 137-139 adds it to all MAX_NUM_OBJ rows), not the origin — it takes part in nn_distance(aggregated_vote_xyz, gt_center)
    # (loss_detection.py:88-92).  Sizes of absent rows are zero boxes' sizes (zero), their residuals stay zero (dataset.py:688).
    batch["center_label"] = out[..., :3].contiguous()
    batch["box_sizes"] = out[..., 3:6] * present
    batch["size_residual_label"] = (out[..., 3:6] - mean[batch["size_class_label"]]) * present
    tgt = batch["ref_box_label_list"].long().unsqueeze(-1).expand(-1, -1, 6)
    ref = torch.gather(out, 1, tgt)
    batch["ref_center_label_list"] = ref[..., :3].contiguous()
    batch["ref_size_residual_label_list"] = (ref[..., 3:6] - mean[batch["ref_size_class_label_list"]]).contiguous()
    return batch


AUGMENT_ONLY_KEYS = ("instance_labels", "instance_valid", "box_sizes")


def augmenting_prepare(rng, prepare=None, mean_size_arr=None, height_col=3):
    """-> a `prepare` callable for Prefetcher: draw this batch's parameters (host, reference call order), augment on the
    device, drop the loader-only arrays the step never reads, then `prepare` (grounding_step.prepare_batch).
    height_col: the column that scale_augment multiplies by the z scale.  Default 3 = the reference AS SHIPPED
    (utils_fn.py:119-120 scales point_cloud[:, 3]; dataset.py:603-607 appends the height as the LAST column, so with normals /
    multiview features the reference scales the first feature channel and leaves the height alone); pass -1 to scale the
    true height channel instead, None for no height scaling (use_height off)."""
    def run(batch):
        B = batch["point_clouds"].shape[0]
        batch = augment_on_device(batch, draw_augment_params(rng, B), mean_size_arr, height_col)
        for k in AUGMENT_ONLY_KEYS:
            batch.pop(k, None)
        return prepare(batch) if prepare is not None else batch
    return run
