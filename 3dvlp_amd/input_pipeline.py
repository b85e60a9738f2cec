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
import numpy as np
import torch


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

    def __init__(self, loader, device="cuda", prepare=None):
        self.loader = iter(loader)
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.prepare = prepare
        self._pinned = [{}, {}]  # two sets of staging buffers: the copy of batch t+1 may still read its set while t+2 is staged
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

    def preload(self):
        try:
            host = next(self.loader)
        except StopIteration:
            self.data_dict = None
            return
        staged = {k: self._stage(k, v) for k, v in host.items()}
        self._flip ^= 1
        with torch.cuda.stream(self.stream):
            dev = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in staged.items()}
            if self.prepare is not None:
                dev = self.prepare(dev)
        self.data_dict = dev

    def next(self):
        """The prefetched batch (device tensors, safe to use on the current stream), or None when the loader is exhausted."""
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        data_dict = self.data_dict
        if data_dict is not None:
            for v in data_dict.values():
                if torch.is_tensor(v):
                    v.record_stream(torch.cuda.current_stream(self.device))
        self.preload()
        return data_dict
