"""The grounding hot path end to end: network graph, reduced loss, one optimisation step.

GroundingNet is the hot-path subset of the reference's JointNet (models/jointnet/jointnet.py:112-220):
backbone -> voting (+L2 norm) -> proposal (vote clustering, ROI heads, decode) -> relation -> match
(proposal<->token cross-attention) -> contrast (OCC/OSC).  Sub-module attribute names equal JointNet's
(backbone_net, vgen, proposal, relation, match, constrast) so that a reference checkpoint's keys map
1:1.  The frozen BERT encoder (`lang`, out of scope) is replaced by its outputs `lang_fea`/`lang_emb`
in the data_dict.

`grounding_loss` is a REDUCED form of lib/loss_helper/loss_joint.py:26-227 (the full detection +
grounding loss stack is the "next" row §8f-1): vote loss and objectness loss as in
loss_detection.py:24-110 (both consume nn_distance), a centre/size regression of the assigned
proposals, the reference cross-entropy over cluster_ref, and 0.5*OCC + 2.5*OSC (loss_joint.py:208).
"""
import importlib
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as _ext
from . import add_norm, synth
from .ddp import FlatGradBucket
from .detection import Pointnet2Backbone, ProposalModule, RelationModule, VotingModule
from .grounding import ContrastModule, MatchModule
from .nn_distance import huber_loss, nn_distance

FAR_THRESHOLD = 0.6   # loss_detection.py:19-23
NEAR_THRESHOLD = 0.3
GT_VOTE_FACTOR = 3
OBJECTNESS_CLS_WEIGHTS = (0.2, 0.8)
_CLS_W = {}


class GroundingNet(nn.Module):
    def __init__(self, num_class=18, num_heading_bin=1, num_size_cluster=18, mean_size_arr=None,
                 input_feature_dim=132, num_proposal=256, vote_factor=1, sampling="vote_fps", use_con=True):
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

    def forward(self, data_dict):
        data_dict = self.backbone_net(data_dict)
        xyz, features = data_dict["fp2_xyz"], data_dict["fp2_features"]
        data_dict["seed_inds"], data_dict["seed_xyz"], data_dict["seed_features"] = data_dict["fp2_inds"], xyz, features
        xyz, features = self.vgen(xyz, features)
        features = features.div(torch.norm(features, p=2, dim=1).unsqueeze(1))
        data_dict["vote_xyz"], data_dict["vote_features"] = xyz, features
        data_dict = self.proposal(xyz, features, data_dict)
        data_dict = self.relation(data_dict)
        data_dict = self.match(data_dict)
        if self.use_con:
            data_dict = self.constrast(data_dict)
        return data_dict


def compute_vote_loss(d):
    """loss_detection.py:24-72."""
    B, S = d["seed_xyz"].shape[:2]
    seed_inds = d["seed_inds"].long()
    mask = torch.gather(d["vote_label_mask"], 1, seed_inds).float()
    gt_votes = torch.gather(d["vote_label"], 1, seed_inds.unsqueeze(-1).expand(-1, -1, 3 * GT_VOTE_FACTOR))
    gt_votes = gt_votes + d["seed_xyz"].repeat(1, 1, 3)
    _, _, dist2, _ = nn_distance(d["vote_xyz"].reshape(B * S, -1, 3), gt_votes.reshape(B * S, GT_VOTE_FACTOR, 3),
                                 l1=True)
    votes_dist = dist2.min(dim=1)[0].view(B, S)
    return torch.sum(votes_dist * mask) / (torch.sum(mask) + 1e-6)


def compute_objectness_loss(d):
    """loss_detection.py:74-110."""
    agg = d["aggregated_vote_xyz"]
    gt_center = d["center_label"][:, :, 0:3]
    dist1, ind1, _, _ = nn_distance(agg, gt_center)
    euc = torch.sqrt(dist1 + 1e-6)
    label = (euc < NEAR_THRESHOLD).long()
    mask = ((euc < NEAR_THRESHOLD) | (euc > FAR_THRESHOLD)).float()
    if agg.device not in _CLS_W:
        _CLS_W[agg.device] = torch.tensor(OBJECTNESS_CLS_WEIGHTS, device=agg.device)
    w = _CLS_W[agg.device]
    ce = F.cross_entropy(d["objectness_scores"].float().transpose(2, 1), label, weight=w, reduction="none")
    return torch.sum(ce * mask) / (torch.sum(mask) + 1e-6), label, mask, ind1


class _LossCore(torch.autograd.Function):
    """Fused vote + objectness + centre + reference loss (csrc/grounding_loss.hip): one forward kernel (+ finalize)
    and one backward kernel.  Returns out5 = [vote, objectness, centre, reference, weighted total]; only the total
    carries gradient (to vote_xyz, objectness_scores, pred_center, cluster_ref)."""
    CONSTS = (NEAR_THRESHOLD, FAR_THRESHOLD, OBJECTNESS_CLS_WEIGHTS[0], OBJECTNESS_CLS_WEIGHTS[1], 0.15, 0.1, 0.3)

    @staticmethod
    def forward(ctx, vote_xyz, obj_scores, pred_center, cluster_ref, seed_xyz, seed_inds, vote_label, vote_mask,
                agg_xyz, center_label, ref_center):
        cf = lambda t: t.contiguous().float()
        vote_xyz, obj_scores, pred_center, cluster_ref = cf(vote_xyz), cf(obj_scores), cf(pred_center), cf(cluster_ref)
        seed_xyz, vote_label, vote_mask, agg_xyz = cf(seed_xyz), cf(vote_label), cf(vote_mask), cf(agg_xyz)
        center_label, ref_center = cf(center_label), cf(ref_center)
        seed_inds = seed_inds.contiguous().int()
        B, S = seed_inds.shape
        N, K, G, L = vote_mask.shape[1], agg_xyz.shape[1], center_label.shape[1], ref_center.shape[1]
        nsum = int(_ext.load().vlp3d_grounding_loss_sums(B, S, K, L))
        sums = torch.empty((nsum,), dtype=torch.float64, device=vote_xyz.device)
        out = torch.empty((5,), dtype=torch.float32, device=vote_xyz.device)
        args = (vote_xyz, seed_xyz, seed_inds, vote_label, vote_mask, agg_xyz, center_label, obj_scores, pred_center,
                cluster_ref, ref_center, B, S, N, K, G, L, *_LossCore.CONSTS)
        _ext.call("vlp3d_grounding_loss_fwd", *args, sums, out)
        ctx.save_for_backward(vote_xyz, seed_xyz, seed_inds, vote_label, vote_mask, agg_xyz, center_label, obj_scores,
                              pred_center, cluster_ref, ref_center, sums)
        ctx.dims = (B, S, N, K, G, L)
        return out

    @staticmethod
    def backward(ctx, gout):
        sv = ctx.saved_tensors
        sums = sv[11]
        vote_xyz, obj_scores, pred_center, cluster_ref = sv[0], sv[7], sv[8], sv[9]
        g = gout[4:5].contiguous()  # only the total is differentiable (the components are reporting values)
        d_vote, d_obj = torch.empty_like(vote_xyz), torch.empty_like(obj_scores)
        d_center, d_ref = torch.empty_like(pred_center), torch.empty_like(cluster_ref)
        _ext.call("vlp3d_grounding_loss_bwd", *sv[:11], *ctx.dims, *_LossCore.CONSTS, sums, g, d_vote, d_obj, d_center,
                  d_ref)
        return d_vote, d_obj, d_center, d_ref, None, None, None, None, None, None, None


FUSED_LOSS = True  # csrc/grounding_loss.hip on CUDA tensors; False = the op-by-op form below (host tests)


def grounding_loss(d, mean_size_arr):
    if FUSED_LOSS and d["vote_xyz"].is_cuda and d["vote_xyz"].shape[1] == d["seed_xyz"].shape[1]:
        out = _LossCore.apply(d["vote_xyz"], d["objectness_scores"], d["pred_center"], d["cluster_ref"], d["seed_xyz"],
                              d["seed_inds"], d["vote_label"], d["vote_label_mask"], d["aggregated_vote_xyz"],
                              d["center_label"][:, :, 0:3], d["ref_center_label_list"][..., 0:3])
        loss = out[4]
        if "lang_con_loss" in d:
            loss = loss + 0.5 * d["lang_con_loss"] + 2.5 * d["iou_con_loss"]  # loss_joint.py:208
        comp = out.detach()
        d["vote_loss"], d["objectness_loss"], d["center_loss"], d["ref_loss"] = comp[0], comp[1], comp[2], comp[3]
        d["loss"] = loss
        return loss
    vote_loss = compute_vote_loss(d)
    obj_loss, obj_label, _, assign = compute_objectness_loss(d)
    B, K = obj_label.shape
    # regression of the assigned GT centre (proposals near an object only)
    gt_center = torch.gather(d["center_label"][:, :, 0:3], 1, assign.unsqueeze(-1).expand(-1, -1, 3))
    pos = obj_label.float()
    center_loss = (huber_loss(d["pred_center"] - gt_center, 0.15).sum(-1) * pos).sum() / (pos.sum() + 1e-6)
    size_loss = (d["pred_size"].mean(-1) * 0.0).sum()  # keeps the size head in the graph
    # reference loss: the proposal nearest to the referred GT centre is the target of cluster_ref
    L = d["ref_center_label_list"].shape[1]
    ref_c = d["ref_center_label_list"][..., 0:3]  # (B,L,3)
    dist = ((d["pred_center"].detach()[:, None, :, :] - ref_c[:, :, None, :]) ** 2).sum(-1)  # (B,L,K)
    target = dist.argmin(-1).reshape(B * L)
    ref_loss = F.cross_entropy(d["cluster_ref"].float(), target)
    loss = vote_loss + 0.1 * obj_loss + center_loss + size_loss + 0.3 * ref_loss
    if "lang_con_loss" in d:
        loss = loss + 0.5 * d["lang_con_loss"] + 2.5 * d["iou_con_loss"]  # loss_joint.py:208
    d["loss"] = loss
    return loss


def batch_to_device(batch, device):
    out = {k: torch.from_numpy(v).to(device) for k, v in batch.items()}
    out["istrain"] = [1]
    return out


class _deferred_bn_counters:
    """While active, training-mode BatchNorm layers with a fixed momentum skip their own
    ``num_batches_tracked.add_(1)`` (the buffer is hidden, torch.nn.modules.batchnorm then leaves the counter
    alone); on exit every hidden counter is restored and all are incremented by ONE fused launch.  Layers with
    ``momentum=None`` need the counter's value during forward and keep their own increment."""

    def __init__(self, model):
        self.mods = [m for m in model.modules()
                     if isinstance(m, nn.modules.batchnorm._BatchNorm) and m.training and m.track_running_stats
                     and m.momentum is not None and m.num_batches_tracked is not None]

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


class GroundingStep:
    """Owns model + optimiser + flat gradient bucket; `run(batch)` = fwd + loss + bwd + all-reduce + AdamW.

    With `use_graph=True` the forward+loss+backward of a (static) batch is captured once into a HIP graph and
    replayed: the step issues ~2500 kernel launches, which eager PyTorch cannot enqueue as fast as the GPU
    retires them.  The step contains no host-side decision or sync (copy-paste augmentation, contrast losses and
    box decode are all fixed-shape device code), which is what makes the capture legal.  The gradient
    all-reduce and the optimiser step stay outside the graph."""

    def __init__(self, device, epoch=50, lr=1e-3, autocast_dtype=None, seed=0, use_graph=False, pipeline=False,
                 sa_dtype=None):
        torch.manual_seed(seed)
        self.device = device
        self.model = GroundingNet().to(device)
        self.model.train()
        self.bucket = FlatGradBucket(self.model)
        # fused: the whole AdamW update is a couple of multi-tensor launches instead of ~30 (one per foreach op and chunk)
        self.opt = torch.optim.AdamW(self.model.parameters(), lr=lr, weight_decay=1e-5,
                                     fused=torch.device(device).type == "cuda")
        self.epoch = epoch
        self.autocast_dtype = autocast_dtype
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
        self._side = torch.cuda.Stream(device=device) if pipeline else None
        self._geom_cur = self._geom_next = None
        self._graph = None
        self._static_batch = self._static_next = None
        self._static_loss = None

    def forward_loss(self, batch, geometry=None):
        d = dict(batch)
        d["epoch"] = self.epoch
        if geometry is not None:
            d["backbone_geometry"] = geometry
        with _deferred_bn_counters(self.model):  # 26 one-element `add_(1)` launches -> one multi-tensor add
            if self.autocast_dtype is not None:
                with torch.autocast(device_type="cuda", dtype=self.autocast_dtype):
                    d = self.model(d)
            else:
                d = self.model(d)
        return grounding_loss(d, self.model.mean_size_arr), d

    @staticmethod
    def _copy_geometry(dst, src):
        for k in src:
            for a, b in zip(dst[k], src[k]):
                a.copy_(b)

    def _fwd_bwd(self, batch, next_batch=None):
        """One forward+loss+backward.  With the pipeline on, uses the geometry prepared during the previous call
        and prepares `next_batch`'s (default: the same batch again) on the side stream meanwhile."""
        geometry = None
        if self.pipeline:
            backbone = self.model.backbone_net
            cur = torch.cuda.current_stream()
            if self._geom_next is None:  # very first call: nothing prepared yet
                self._geom_next = backbone.compute_geometry(batch["point_clouds"])
                self._geom_cur = {k: tuple(t.clone() for t in v) for k, v in self._geom_next.items()}
            self._copy_geometry(self._geom_cur, self._geom_next)  # tiny: indices + sampled coordinates
            self._side.wait_stream(cur)  # fork: the side branch may overwrite _geom_next from here on
            with torch.cuda.stream(self._side):
                nxt = backbone.compute_geometry((next_batch or batch)["point_clouds"])
                self._copy_geometry(self._geom_next, nxt)
            geometry = self._geom_cur
        self.bucket.zero()
        loss, _ = self.forward_loss(batch, geometry)
        loss.backward()
        self.bucket.collect()
        add_norm.advance(self.device)  # fresh dropout masks next step (also when this is a captured graph)
        if self.pipeline:
            torch.cuda.current_stream().wait_stream(self._side)  # join
        return loss.detach()

    def _capture(self, batch, next_batch):
        self._static_batch = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        self._static_next = self._static_batch if next_batch is None else \
            {k: (v.clone() if torch.is_tensor(v) else v) for k, v in next_batch.items()}
        warm = torch.cuda.Stream()
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):  # warm-up off the capture stream (allocator, MIOpen/hipBLASLt selection)
            for _ in range(2):
                self._fwd_bwd(self._static_batch, self._static_next)
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        if not self.pipeline:
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._static_loss = self._fwd_bwd(self._static_batch, self._static_next)
            return
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
            nxt = backbone.compute_geometry(self._static_next["point_clouds"])
            self._copy_geometry(self._geom_next, nxt)
        with torch.cuda.graph(self._gM):
            self.bucket.zero()
            loss, _ = self.forward_loss(self._static_batch, self._geom_cur)
            loss.backward()
            self.bucket.collect()
            add_norm.advance(self.device)
            self._static_loss = loss.detach()
        self._graph = self._gM

    def _replay(self):
        if not self.pipeline:
            self._graph.replay()
            return
        cur = torch.cuda.current_stream()
        self._gC.replay()
        self._side.wait_stream(cur)          # fork: geometry of the next batch may overwrite _geom_next now
        with torch.cuda.stream(self._side):
            self._gS.replay()
        self._gM.replay()
        cur.wait_stream(self._side)          # join

    @staticmethod
    def _refill(static, batch):
        for k, v in batch.items():
            if torch.is_tensor(v) and static[k].data_ptr() != v.data_ptr():
                static[k].copy_(v, non_blocking=True)

    def run(self, batch, next_batch=None):
        if self.use_graph:
            if self._graph is None:
                self._capture(batch, next_batch)
            else:
                if batch is not self._static_batch:
                    self._refill(self._static_batch, batch)
                if next_batch is not None and next_batch is not self._static_next:
                    self._refill(self._static_next, next_batch)
            self._replay()
            loss = self._static_loss
        else:
            loss = self._fwd_bwd(batch, next_batch)
        self.bucket.all_reduce()
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
