"""The grounding hot path end to end: network graph, the reference's training loss, one optimisation step.

GroundingNet is the hot-path subset of the reference's JointNet (models/jointnet/jointnet.py:112-220):
backbone -> voting (+L2 norm) -> proposal (vote clustering, ROI heads, decode) -> relation -> match
(proposal<->token cross-attention) -> contrast (OCC/OSC).  Sub-module attribute names equal JointNet's
(backbone_net, vgen, proposal, relation, match, constrast) so that a reference checkpoint's keys map
1:1.  The frozen BERT encoder (`lang`, out of scope) is replaced by its outputs `lang_fea`/`lang_emb`
in the data_dict.

`grounding_loss` is the reference's loss for this path (lib/loss_helper/loss_joint.py:26-227 with detection +
reference + DIoU + OCC/OSC as run.sh:1 configures it): 3dvlp_amd/losses.py, fused in csrc/joint_loss.hip.
"""
import gc
import importlib
import os
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as _ext
from . import mfma_linear
from . import add_norm, losses, row_mlp, synth
from .ddp import FlatAdamW, FlatGradBucket, FlatParams
from .detection import Pointnet2Backbone, ProposalModule, RelationModule, VotingModule
from .grounding import ContrastModule, MatchModule


class GroundingNet(nn.Module):
    def __init__(self, num_class=18, num_heading_bin=1, num_size_cluster=18, mean_size_arr=None,
                 input_feature_dim=132, num_proposal=256, vote_factor=1, sampling="vote_fps", use_con=True,
                 use_answer=False, num_answers=0, use_caption=False, caption_kwargs=None):
        super().__init__()
        mean_size_arr = synth.mean_size_arr() if mean_size_arr is None else mean_size_arr
        assert mean_size_arr.shape[0] == num_size_cluster
        self.num_class, self.num_heading_bin, self.num_size_cluster = num_class, num_heading_bin, num_size_cluster
        self.mean_size_arr = mean_size_arr
        self.dataset_config = SimpleNamespace(mean_size_arr=mean_size_arr, num_heading_bin=num_heading_bin,
                                              num_size_cluster=num_size_cluster, num_class=num_class)
        self.use_con = use_con
        self.backbone_net = Pointnet2Backbone(input_feature_dim=input_feature_dim)
        self.vgen = VotingModule(vote_factor, 256)
        self.proposal = ProposalModule(num_class, num_heading_bin, num_size_cluster, mean_size_arr, num_proposal,
                                       sampling)
        self.relation = RelationModule(num_proposals=num_proposal, det_channel=128)
        if use_con:
            self.constrast = ContrastModule(config=self.dataset_config)
        self.match = MatchModule(num_proposals=num_proposal, lang_size=256, det_channel=128)
        self.use_caption = use_caption
        if use_caption:  # the Scan2Cap head on the shared proposal features (jointnet.py:103-104, 214-215; BASELINE cfg4)
            from .caption import TransformerDecoderModel
            self.caption = TransformerDecoderModel(30522, **(caption_kwargs or {}))
        self.use_answer = use_answer
        if use_answer:  # the ScanQA head of the joint QA + grounding task (jointnet.py:109-110, 217-218; BASELINE cfg5)
            from .answer import AnswerModule
            self.answer = AnswerModule(num_answers=num_answers)

    def forward(self, data_dict):
        data_dict = self.backbone_net(data_dict)
        xyz, features = data_dict["fp2_xyz"], data_dict["fp2_features"]
        data_dict["seed_inds"], data_dict["seed_xyz"], data_dict["seed_features"] = data_dict["fp2_inds"], xyz, features
        fused = self.vgen.forward_normalized(xyz, features)  # votes + jointnet.py:148-149's L2 norm in one epilogue kernel
        if fused is not None:
            xyz, features = fused
        else:
            xyz, features = self.vgen(xyz, features)
            features = features.div(torch.norm(features, p=2, dim=1).unsqueeze(1))
        data_dict["vote_xyz"], data_dict["vote_features"] = xyz, features
        data_dict = self.proposal(xyz, features, data_dict)
        data_dict = self.relation(data_dict)
        data_dict = self.match(data_dict)
        if self.use_con:
            data_dict = self.constrast(data_dict)
        if self.use_caption:
            data_dict = self.caption(data_dict)
        if self.use_answer:
            data_dict = self.answer(data_dict)
        return data_dict


LOSS_IMPL = None  # None -> losses.DEFAULT_IMPL ("hip": csrc/joint_loss.hip); "torch" = batched op-by-op form (tests)


def grounding_loss(d, config, args=None, impl=None, caption=False):
    """The reference's training loss for this path: losses.get_joint_loss == lib/loss_helper/loss_joint.py:26-227 with
    detection + reference (+ DIoU + OCC/OSC, run.sh:1); the language-classification term belongs to the out-of-scope
    language encoder and is included only when the batch carries `lang_scores`."""
    losses.get_joint_loss(args, d, config=config, caption=caption, impl=impl or LOSS_IMPL)
    return d["loss"]


_MEAN_SIZE_CACHE = {}


def prepare_batch(out, mean_size_arr=None, feat_bf16=False):
    """Device batch (the reference's keys and dtypes) -> the same dict plus the KERNEL-READY forms of pure input data
    that the step would otherwise re-derive with a dozen small launches every iteration (loader work: it depends on the
    batch only, never on the model; input_pipeline.Prefetcher runs it on the copy stream):
      k/vote_label_mask f32, k/{heading_class,size_class,sem_cls}_label i32, k/lang_num i32 — dtypes the loss kernel
      reads (lib/joint/dataset.py hands them over as int64);  k/ref_size — decoded size of the referred boxes
      (class2size, model_util_scannet.py:183-185; consumed by the DIoU loss and the contrast module);
      k/lang_kv — lang_fea[:, 1:] contiguous (the K/V tokens of match_module.py:134);
      k/xyz, k/feat_pm — the cloud split into coordinates and point-major features (backbone_module.py:73-80 slices
      and copies them inside forward: 173 MB per step at cfg2).  feat_bf16 (bf16 configuration only): the features as BF16
      rows instead — k/feat_bf (B,N,round_up(C,8)) zero padded + k/feat_c = C — which the first grouped-MLP layer and its
      weight gradient read directly (the fp32 rows are rounded to exactly these values on their way into LDS).  A batch that
      arrives as (k/xyz, k/feat_bf, k/feat_c) — input_pipeline.compress_cloud on the host — is left as it is."""
    device = (out["point_clouds"] if "point_clouds" in out else out["k/xyz"]).device
    out.setdefault("istrain", [1])
    # (cached per device: a pageable host -> device copy is SYNCHRONOUS — issued on the copy stream behind a batch's upload it
    # blocked the host for the whole upload, 2.8 ms per step: tools/host_feed_timeline.py)
    key = (str(device), None if mean_size_arr is None else id(mean_size_arr))
    mean = _MEAN_SIZE_CACHE.get(key)
    if mean is None:
        mean = torch.as_tensor(synth.mean_size_arr() if mean_size_arr is None else mean_size_arr, dtype=torch.float32,
                               device=device)
        if mean_size_arr is None:
            _MEAN_SIZE_CACHE[key] = mean
    out["k/vote_label_mask"] = out["vote_label_mask"].float()
    for k in ("heading_class_label", "size_class_label", "sem_cls_label", "lang_num"):
        if k in out:
            out["k/" + k] = out[k].to(torch.int32)
    out["k/ref_size"] = (mean[out["ref_size_class_label_list"]] + out["ref_size_residual_label_list"]).float().contiguous()
    out["k/lang_kv"] = out["lang_fea"][:, 1:].contiguous()
    if "k/feat_bf" in out and "k/xyz" in out:
        # the loader sent the cloud already split, its feature channels as bf16 rows (input_pipeline.compress_cloud: half the
        # PCIe bytes of the step's largest input): the bf16 kernels read them as they are
        return out
    pc = out["point_clouds"]
    if pc.size(-1) > 3:
        out["k/xyz"] = pc[..., :3].contiguous()
        if feat_bf16:   # the same split made on the device (resident batches, or behind the device-side augmentation)
            C = pc.size(-1) - 3
            rows = torch.zeros(pc.shape[:-1] + ((C + 7) // 8 * 8,), dtype=torch.bfloat16, device=pc.device)
            rows[..., :C] = pc[..., 3:]
            out["k/feat_bf"], out["k/feat_c"] = rows, C
        else:
            out["k/feat_pm"] = pc[..., 3:].contiguous()
    return out


def batch_to_device(batch, device, mean_size_arr=None, feat_bf16=False):
    """Host batch (numpy) -> device tensors + prepare_batch (synchronous form; see input_pipeline.Prefetcher)."""
    return prepare_batch({k: torch.from_numpy(v).to(device) for k, v in batch.items()}, mean_size_arr, feat_bf16)


def _detached(out):
    """The step's data_dict without its autograd graph: same storages (so the allocator keeps them), no grad_fn chain — a
    graph kept alive across iterations keeps last iteration's AccumulateGrad nodes, bound to that iteration's stream."""
    if os.environ.get("VLP3D_DEBUG_KEEP_AUTOGRAD") == "1":
        return out
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}


class _deferred_bn_counters:
    """While active, training-mode BatchNorm layers with a fixed momentum skip their own
    ``num_batches_tracked.add_(1)`` (the buffer is hidden, torch.nn.modules.batchnorm then leaves the counter
    alone); on exit every hidden counter is restored and all are incremented by ONE fused launch.  Layers with
    ``momentum=None`` need the counter's value during forward and keep their own increment; layers a module reports
    as `unused_batchnorms()` (present only so that reference checkpoints load) are left alone, like the reference
    leaves them (round 3: the whole-step comparison with CpuStep found match.lang_emb_proj's counters advancing)."""

    def __init__(self, model):
        idle = {id(b) for m in model.modules() if hasattr(m, "unused_batchnorms") for b in m.unused_batchnorms()}
        self.mods = [m for m in model.modules()
                     if isinstance(m, nn.modules.batchnorm._BatchNorm) and m.training and m.track_running_stats
                     and m.momentum is not None and m.num_batches_tracked is not None and id(m) not in idle]

    def __enter__(self):
        self.counters = [m.num_batches_tracked for m in self.mods]
        for m in self.mods:
            m._buffers["num_batches_tracked"] = None

    def __exit__(self, *exc):
        for m, c in zip(self.mods, self.counters):
            m._buffers["num_batches_tracked"] = c
        if self.counters and exc[0] is None:
            torch._foreach_add_(self.counters, 1)
        return False


class _BatchTag:
    """(tensor object, in-place version); equal only to a tag of the SAME live tensor at the same version."""
    __slots__ = ("src", "version")

    def __init__(self, t):
        self.src, self.version = t, t._version

    def __eq__(self, other):
        return isinstance(other, _BatchTag) and self.src is other.src and self.version == other.version

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None


class GroundingStep:
    """Owns model + optimiser + flat gradient bucket; `run(batch)` = fwd + loss + bwd + all-reduce + AdamW.

    With `use_graph=True` the forward+loss+backward of a (static) batch is captured once into a HIP graph and
    replayed: the step issues ~2500 kernel launches, which eager PyTorch cannot enqueue as fast as the GPU
    retires them.  The step contains no host-side decision or sync (copy-paste augmentation, contrast losses and
    box decode are all fixed-shape device code), which is what makes the capture legal.  The gradient
    all-reduce and the optimiser step stay outside the graph."""

    def __init__(self, device, epoch=50, lr=1e-3, autocast_dtype=None, seed=0, use_graph=False, pipeline=False,
                 sa_dtype=None, use_answer=False, num_answers=0, side_stream=None, use_caption=False, caption_kwargs=None):
        """use_caption: the caption head (3dvlp_amd.caption.TransformerDecoderModel(30522), jointnet.py:104) is part of the
        model — its parameters live in the flat parameter / gradient / AdamW buffers, `cap_loss` is added inside the captured
        graph (loss_joint.py:222-223), ONE backward, the same all-reduce (BASELINE cfg4).  The batch then carries `input_ids`
        (B, L, T) token ids."""
        torch.manual_seed(seed)
        self.device = device
        self.use_caption = bool(use_caption)
        self.model = GroundingNet(use_answer=use_answer, num_answers=num_answers, use_caption=use_caption,
                                  caption_kwargs=caption_kwargs).to(device)
        self.loss_args = type("Args", (losses._Args,), {"use_answer": bool(use_answer)})
        self.model.train()
        if torch.device(device).type == "cuda":
            # parameters, gradients and AdamW moments as three flat buffers with one layout: merged projections read their
            # concatenated weights as views, the optimiser is one launch (ddp.FlatParams / FlatAdamW)
            self.layout = FlatParams(self.model)
            self.bucket = FlatGradBucket(self.model, layout=self.layout)
            self.opt = FlatAdamW(self.layout, self.bucket, lr=lr, weight_decay=1e-5)
        else:
            self.layout = None
            self.bucket = FlatGradBucket(self.model)
            self.opt = torch.optim.AdamW(self.model.parameters(), lr=lr, weight_decay=1e-5)
        self._wprep = row_mlp.PreparedWeights()  # K-major weight copies of the rows stacks, refreshed once per forward pass
        self.epoch = epoch
        self.autocast_dtype = autocast_dtype
        self.sa_dtype = sa_dtype
        # bf16 for the grouped-MLP kernels only (the dense work that matters); everything else stays fp32, which
        # removes the ~350 per-step cast kernels autocast would launch around the many small layers
        for m in self.model.modules():
            if hasattr(m, "mlp_dtype"):
                m.mlp_dtype = sa_dtype
            if hasattr(m, "bf16_mma"):  # fused attention cores follow the same switch
                m.bf16_mma = sa_dtype == torch.bfloat16
            if hasattr(m, "fused_norm"):  # dropout + residual + LayerNorm kernels; their seed word advances per step
                m.fused_norm = True
        self.use_graph = use_graph
        # geometry pipeline: the backbone's coordinate-only stage (FPS / ball query / three_nn) of the NEXT batch
        # runs on a side stream while the dense layers of the current batch run (one workgroup per scene = 8 CUs)
        self.pipeline = pipeline
        # (side_stream: reuse another step's stream — every new HIP stream takes one of the few hardware queues, and a side
        # stream that lands on the main stream's queue serialises the two: a THIRD step object in one process ran 9.8 ms)
        # (VLP3D_SIDE_PRIORITY: experiment knob — stream priority of the geometry / deferred-work stream; default 0 = normal)
        self._side = (side_stream or torch.cuda.Stream(device=device, priority=int(os.environ.get("VLP3D_SIDE_PRIORITY", 0)))) \
            if pipeline else None
        self._geom_cur = self._geom_next = None
        self._geom_tag = None      # eager pipeline: the batch _geom_next was prepared for
        self._geom_for = None      # graph pipeline: the batch _geom_next was prepared for
        self._static_tag = self._next_src_tag = None  # sources the static graph buffers were last filled from
        self._graph = self._gD = self._gM2 = None
        self._regime = None        # epoch < 50 at capture: the loss configuration the captured graph holds
        self.on_capture = None     # optional callable run right before the graphs are captured (after the warm-up passes)
        self._static_batch = self._static_next = None
        self._static_loss = None
        self._last_out = self._static_out = None
        # gradient all-reduce in two pieces (split backward only): the head parameters' slice of the flat buffer is complete when
        # the deferred graph gD ends — SA2 / SA1 backward (gM2, ~0.95 ms) still runs then — and is reduced beside it; the tail's
        # slice follows gM2.  Ranges of the flat buffer, or None (one all-reduce after the step, as in every other mode).
        self._head_range = self._tail_range = None
        self._pending_head = None
        self.comm_events = None    # set to [] to collect (event before, event after) the exposed all-reduce section of run()

    def forward_loss(self, batch, geometry=None):
        d = dict(batch)
        d["epoch"] = self.epoch
        if geometry is not None:
            d["backbone_geometry"] = geometry
        # 26 one-element `add_(1)` launches -> one multi-tensor add; plain linear layers follow the grouped MLPs' dtype
        with self._wprep, _deferred_bn_counters(self.model), mfma_linear.bf16_mma(self.sa_dtype == torch.bfloat16 and os.environ.get("VLP3D_LINEAR_BF16", "1") != "0"):
            if self.autocast_dtype is not None:
                with torch.autocast(device_type="cuda", dtype=self.autocast_dtype):
                    d = self.model(d)
            else:
                d = self.model(d)
        return grounding_loss(d, self.model.dataset_config, self.loss_args, caption=self.use_caption), d

    @staticmethod
    def _copy_geometry(dst, src):
        """All index / coordinate tensors of the backbone geometry in ONE launch (torch._foreach_copy_ issued 16 memcpy
        nodes and two kernels for them)."""
        d = [a for k in src for a in dst[k]]
        s_ = [b for k in src for b in src[k]]
        if d[0].is_cuda and all(b.is_contiguous() for b in s_):
            _ext.copy_batch(d, s_)
        else:
            torch._foreach_copy_(d, s_)

    @staticmethod
    def _tag(batch):
        """Identity of a batch's coordinates: the point_clouds tensor OBJECT + its in-place version.  The prepared geometry
        (and a captured graph's static inputs) are only ever used for the batch they were filled from.  The tag holds a
        strong reference to the tensor: an address-based identity is unsound, because the caching allocator hands the block
        of a freed batch to the next upload of the same size (`run(batch_to_device(...))` with a temporary would then look
        like the previous batch and the replay would silently train on stale data — ADVICE r2)."""
        pc = batch["point_clouds"] if "point_clouds" in batch else batch["k/xyz"]  # (the static inputs keep the split only)
        return _BatchTag(pc)

    def _fwd_bwd(self, batch, next_batch=None):
        """One forward+loss+backward.  With the pipeline on, uses the geometry prepared during the previous call IF
        it was prepared for this very batch (else computes it inline on the main stream), and prepares
        `next_batch`'s on the side stream meanwhile (default guess: the same batch comes again)."""
        geometry = None
        if self.pipeline:
            backbone = self.model.backbone_net
            cur = torch.cuda.current_stream()
            if self._geom_next is None or self._geom_tag != self._tag(batch):
                # nothing prepared, or prepared for another batch: geometry inline (correct, just not overlapped)
                fresh = backbone.compute_geometry(self._coords(batch))
                if self._geom_next is None:
                    self._geom_next = fresh
                    self._geom_cur = {k: tuple(t.clone() for t in v) for k, v in fresh.items()}
                else:
                    cur.wait_stream(self._side)
                    self._copy_geometry(self._geom_cur, fresh)
            else:
                self._copy_geometry(self._geom_cur, self._geom_next)  # tiny: indices + sampled coordinates
            nxt_batch = batch if next_batch is None else next_batch
            self._side.wait_stream(cur)  # fork: the side branch may overwrite _geom_next from here on
            with torch.cuda.stream(self._side):
                nxt = backbone.compute_geometry(self._coords(nxt_batch))
                self._copy_geometry(self._geom_next, nxt)
            self._geom_tag = self._tag(nxt_batch)
            geometry = self._geom_cur
        self.bucket.zero()
        # the step's data_dict stays referenced, detached (self._last_out; _static_out for a captured step): its tensors
        # are the replayed graph's outputs (losses, predictions, labels) that callers read after run()
        loss, out = self.forward_loss(batch, geometry)
        self._backward(loss)
        self._last_out = _detached(out)
        self.bucket.collect()
        add_norm.advance(self.device)  # fresh dropout masks next step (also when this is a captured graph)
        if self.pipeline:
            torch.cuda.current_stream().wait_stream(self._side)  # join
        return loss.detach()

    def _backward(self, loss):
        """backward with every weight-gradient slab sum of the pass deferred into one launch (51 -> 2 at cfg2).
        The deferred queue hands autograd VIEWS of buffers that are filled at the flush: that is only sound while
        AccumulateGrad steals the view, i.e. while the parameter has no gradient yet — checked here."""
        self._check_no_stale_grads()
        with _ext.deferred_slab_reduce():
            loss.backward()

    def _check_no_stale_grads(self):
        stale = [n for n, p in self.model.named_parameters() if p.grad is not None]
        if stale:
            raise RuntimeError("GroundingStep._backward: parameters still hold a gradient (%s, ...): call bucket.zero() first — "
                               "accumulating into an existing .grad would read the deferred buffers before they are filled"
                               % stale[0])

    def _persistent_state(self):
        """Everything a forward pass mutates besides the parameters: BatchNorm running statistics / counters and the
        add-norm dropout seed word."""
        return list(self.model.buffers()) + [add_norm.state(self.device)]

    def _capture(self, batch, next_batch):
        # The cyclic garbage collector stays OFF while the graphs are captured: a collection that runs on the capturing
        # (or the autograd) thread in the middle of a capture can finalise an unrelated object that owns HIP resources — a
        # CUDAGraph of an earlier step object kept alive by a reference cycle did, and its destructor's calls are illegal
        # under capture: the process aborted (round 3, seen once in three full test runs).  Everything collectable goes first.
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            self._capture_graphs(batch, next_batch)
        finally:
            if gc_was_on:
                gc.enable()

    def _capture_graphs(self, batch, next_batch):
        # static inputs of the captured graphs.  With the loader's split of the cloud (k/xyz, k/feat_pm) the step never reads
        # `point_clouds` itself: it is left out (173 MB less to refill per step when batches change); of the NEXT batch
        # only the coordinates are needed (its geometry).
        split = "k/xyz" in batch and "k/feat_pm" in batch
        clone = lambda b: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()
                           if not (split and k == "point_clouds")}
        self._static_batch = clone(batch)
        nb = batch if next_batch is None else next_batch
        ck = "k/xyz" if "k/xyz" in nb else "point_clouds"
        self._static_next = {ck: nb[ck].clone()}  # always its own buffer
        self._static_tag = self._tag(batch)
        self._next_src_tag = self._tag(batch if next_batch is None else next_batch)
        # the warm-up passes are real training forwards (BN momentum updates, counters, dropout seed): snapshot the
        # persistent state and put it back, so that a graph step leaves exactly the state an eager step leaves
        keep = [t.clone() for t in self._persistent_state()]
        warm = torch.cuda.Stream()
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):  # warm-up off the capture stream (allocator, library kernel selection)
            for _ in range(2):
                self._fwd_bwd(self._static_batch, self._static_next)
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        if self.on_capture is not None:   # instrumentation hook (bench.py: reset the in-step stamp log / fall-back counters)
            self.on_capture()
        if not self.pipeline:
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._static_loss = self._fwd_bwd(self._static_batch, self._static_next)
            self._static_out = self._last_out
        else:
            # Pipelined: THREE single-stream graphs instead of one graph with a forked branch.  ROCm launches a
            # linear graph in ~0.3 ms of host time but walks a multi-stream graph node by node (16 ms for the 1300
            # nodes of this step), which made the step host-bound.  The fork/join are two events outside the graphs:
            #   gC (main): hand the prepared geometry over;   gS (side): geometry of the next batch;
            #   gM (main): forward + loss + backward of the current batch.
            backbone = self.model.backbone_net
            self._gC, self._gS, self._gM = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._gC):
                self._copy_geometry(self._geom_cur, self._geom_next)
            with torch.cuda.graph(self._gS, stream=self._side):
                nxt = backbone.compute_geometry(self._coords(self._static_next))
                self._copy_geometry(self._geom_next, nxt)
            self._gD = self._gM2 = None
            boundary_keys = ("sa2_features",)
            if os.environ.get("VLP3D_SPLIT_BACKWARD", "1") != "0":
                # Backward in two autograd passes with the weight-gradient-only work between them on the side stream.
                # Everything the deferred queue holds when backward reaches the backbone's second set-abstraction level (the
                # batched weight gradients of the rows stacks and plain linear layers, the relation bias MLP's backward,
                # the slab sums: ~0.5 ms of launches that feed nothing but the optimiser) used to run at the END of the
                # main stream.  The side stream has finished the next batch's geometry by then and idles; so:
                #   gM  (main): forward + loss + backward down to d(sa2_features)   -> event
                #   gD  (side): flush of the deferred queue, concurrent with ...
                #   gM2 (main): ... the backward of SA2 and SA1 (d(sa2_features) as root), their slab sums, collect.
                # sa2_features is a CUT of the autograd graph (everything below it reaches the loss only through it; sa3 /
                # sa4 are not: `inputs=` does not stop the engine at a tensor, and d(sa3) already contains the path through
                # sa4), so the two passes run every node once and the gradients are those of the one-pass backward.  Still
                # LINEAR graphs only: ROCm walks a forked graph node by node (see above).
                self._gD, self._gM2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                head = [p for n, p in self.model.named_parameters()
                        if p.requires_grad and not n.startswith(("backbone_net.sa1.", "backbone_net.sa2."))]
                qctx = _ext.deferred_slab_reduce()
                with torch.cuda.graph(self._gM):
                    self.bucket.zero()
                    loss, out = self.forward_loss(self._static_batch, self._geom_cur)
                    boundary = [out[k] for k in boundary_keys]
                    self._check_no_stale_grads()
                    queue = qctx.__enter__()
                    # (retain_graph: without it the engine releases the saved tensors of the nodes it did NOT run as well)
                    # autograd.grad, not backward(inputs=...): the latter calls retain_grad() on the non-leaf boundary, whose
                    # hook CLONES d(sa2_features) into a channel-major .grad here and, firing again in the second pass, adds
                    # the cotangent onto it; the reshape in front of SA2's backward then copies it back to point-major rows:
                    # three 8 MB element-wise launches (13 + 6 + 19 us in the kernel trace) for nothing.  The captured
                    # gradient is the engine's own buffer — point-major like its producers (SA3 / FP2 input gradients).
                    # The head parameters' gradients come back as tensors and are attached by hand (no AccumulateGrad clone
                    # of the shared zero gradients of biases in front of a train-mode BatchNorm either).
                    grads = torch.autograd.grad([loss], head + boundary, retain_graph=True, allow_unused=True)
                    for p_, g_ in zip(head, grads[:len(head)]):
                        p_.grad = g_
                    bgrads = list(grads[len(head):])
                    del grads
                head_ids = {id(p) for p in head}
                tail = [p for p in self.model.parameters() if id(p) not in head_ids]
                rh, rt = self.bucket.param_range(head), self.bucket.param_range([p for p in tail if p.requires_grad])
                if (rh is not None and rt is not None and sorted([rh, rt])[0][1] == sorted([rh, rt])[1][0]
                        and rh[1] - rh[0] + rt[1] - rt[0] == self.bucket.flat.numel()
                        and os.environ.get("VLP3D_OVERLAP_ALLREDUCE", "1") != "0"):
                    self._head_range, self._tail_range = rh, rt
                with torch.cuda.graph(self._gD, stream=self._side):
                    queue.flush()
                    # the head parameters' gradients are complete HERE, on this stream: their copy into the flat buffer must
                    # not run on the main stream beside this graph (it would read buffers the flush is still filling)
                    self.bucket.collect_subset(head)
                with torch.cuda.graph(self._gM2):
                    torch.autograd.backward(boundary, bgrads)
                    qctx.__exit__(None, None, None)
                    del bgrads
                    self._static_out = _detached(out)
                    self.bucket.collect_subset(tail)
                    add_norm.advance(self.device)
                    self._static_loss = loss.detach()
                del boundary, out, loss
            else:
                with torch.cuda.graph(self._gM):
                    self.bucket.zero()
                    loss, out = self.forward_loss(self._static_batch, self._geom_cur)
                    self._backward(loss)
                    self._static_out = _detached(out)
                    self.bucket.collect()
                    add_norm.advance(self.device)
                    self._static_loss = loss.detach()
            self._graph = self._gM
            # _geom_next currently holds the geometry of _static_next (computed by the warm-up passes)
            self._geom_for = self._next_src_tag
        torch.cuda.synchronize()
        for t, k in zip(self._persistent_state(), keep):
            t.copy_(k)

    def _replay(self):
        if not self.pipeline:
            self._graph.replay()
            return
        cur = torch.cuda.current_stream()
        if self._geom_for != self._static_tag:
            # the prepared geometry belongs to another batch than the one about to run: recompute inline (eager)
            cur.wait_stream(self._side)
            fresh = self.model.backbone_net.compute_geometry(self._coords(self._static_batch))
            self._copy_geometry(self._geom_next, fresh)
        self._gC.replay()
        self._side.wait_stream(cur)          # fork: geometry of the next batch may overwrite _geom_next now
        with torch.cuda.stream(self._side):
            self._gS.replay()
        self._geom_for = self._next_src_tag
        self._gM.replay()
        if self._gD is not None:
            self._side.wait_stream(cur)      # the deferred weight-gradient work reads what gM left behind
            with torch.cuda.stream(self._side):
                self._gD.replay()
                if self._head_range is not None:
                    # issued under the side stream: the collective (RCCL: on the process group's own stream) is ordered behind
                    # gD and runs beside gM2; a no-op handle without a process group
                    self._pending_head = self.bucket.all_reduce_range(*self._head_range)
            self._gM2.replay()
        cur.wait_stream(self._side)          # join

    @staticmethod
    def _refill(static, batch):
        """New values into the captured graphs' static inputs: ONE launch for all tensors that are plain device copies (~40
        per batch: 40 `copy_` launches cost the host-fed step 0.1 ms of launch stream and 0.2 ms of host time)."""
        dst, src = [], []
        for k, sv in static.items():
            v = batch[k]
            if torch.is_tensor(v) and sv.data_ptr() != v.data_ptr():
                if (v.is_cuda and sv.is_cuda and v.dtype == sv.dtype and v.shape == sv.shape and v.is_contiguous()
                        and sv.is_contiguous() and v.numel() > 0):
                    dst.append(sv)
                    src.append(v)
                else:
                    sv.copy_(v, non_blocking=True)
        if dst:
            _ext.copy_batch(dst, src)

    @staticmethod
    def _coords(batch):
        """Coordinates of a batch for compute_geometry: the loader's k/xyz (B,N,3) when present, else the cloud."""
        return batch["k/xyz"] if "k/xyz" in batch else batch["point_clouds"]

    def run(self, batch, next_batch=None):
        if self.use_graph:
            if self._graph is not None and (self.epoch < 50) != self._regime:
                # the loss configuration baked into the captured graph (reference-loss weight 0.3 / 1.0, label smoothing below
                # epoch 50, OCC / OSC from epoch 50 on; loss_joint.py:208, loss_grounding.py) no longer matches: recapture
                self._graph = self._gC = self._gS = self._gM = self._gD = self._gM2 = None
                self._static_out = None
                self._head_range = self._tail_range = None
            if self._graph is None:
                self._regime = self.epoch < 50
                self._capture(batch, next_batch)
            else:
                # static buffers are refilled whenever the caller's batch is not the one they currently hold
                # (identity = address + in-place version of point_clouds)
                if self._tag(batch) != self._static_tag:
                    self._refill(self._static_batch, batch)
                    self._static_tag = self._tag(batch)
                nxt = batch if next_batch is None else next_batch
                if self._tag(nxt) != self._next_src_tag:
                    if self.pipeline:
                        torch.cuda.current_stream().wait_stream(self._side)
                    self._refill(self._static_next, nxt)
                    self._next_src_tag = self._tag(nxt)
            self._replay()
            loss = self._static_loss
        else:
            loss = self._fwd_bwd(batch, next_batch)
        ev = None
        if self.comm_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if self._pending_head is not None:
            self.bucket.all_reduce_range(*self._tail_range).wait()
            self._pending_head.wait()
            self._pending_head = None
        else:
            self.bucket.all_reduce()
        if ev is not None:   # what the step WAITS for the collectives: from the end of backward to the optimiser's start
            ev[1].record()
            self.comm_events.append(ev)
        self.opt.step()
        return loss


def smoke_step():
    """One tiny forward+backward of the whole path on cuda:0 (called by __graft_entry__.smoke)."""
    dev = torch.device("cuda:0")
    step = GroundingStep(dev)
    batch = batch_to_device(synth.make_batch(0, 2, num_points=8192, lang_num_max=2), dev)
    loss = step.run(batch)
    torch.cuda.synchronize()
    assert torch.isfinite(loss).item(), "non-finite loss"
    print("smoke step loss", float(loss.detach()))
