"""Detection half of the grounding path: backbone, voting, vote clustering + ROI heads, relation module.

Interfaces, shapes and state_dict keys follow the reference so that its callers (jointnet.py /
refnet.py) and checkpoints work unchanged:
  Pointnet2Backbone  — models/base_module/backbone_module.py:11-135
  VotingModule       — models/base_module/voting_module.py:11-60
  StandardROIHeads   — models/proposal_module/ROI_heads/roi_heads.py:15-147
  ProposalModule     — models/proposal_module/proposal_module_fcos.py:21-144
  RelationModule     — models/proposal_module/relation_module.py:9-139
Differences that are deliberate: the box decode stays on the device (the reference
round-trips through numpy inside forward, proposal_module_fcos.py:127-130), and the relation
module's pairwise tensors are built by broadcasting instead of .repeat().
"""
import os
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as _ext
from .pointnet2_modules import PointnetFPModule, PointnetSAModuleVotes
from . import glue, row_mlp
from .ddp import merge_adjacent
from . import mfma_linear
from .mfma_linear import linear as _linear
from .transformer import MultiHeadAttention

# sa2..sa4 sample from the previous level's FPS-ordered new_xyz: prove "indices = 0..m-1" in parallel instead of running the
# sequential kernel (csrc/fps.hip vlp3d_fps_prefix_check; identical output).  VLP3D_FPS_PREFIX=0 turns the hint off.
FPS_PREFIX_HINT = os.environ.get("VLP3D_FPS_PREFIX", "1") != "0"


class Pointnet2Backbone(nn.Module):
    """4 set-abstraction + 2 feature-propagation layers; reads/writes the reference's data_dict keys."""

    def __init__(self, input_feature_dim=0):
        super().__init__()
        self.input_feature_dim = input_feature_dim
        self.sa1 = PointnetSAModuleVotes(npoint=2048, radius=0.2, nsample=64, mlp=[input_feature_dim, 64, 64, 128],
                                         use_xyz=True, normalize_xyz=True)
        self.sa2 = PointnetSAModuleVotes(npoint=1024, radius=0.4, nsample=32, mlp=[128, 128, 128, 256],
                                         use_xyz=True, normalize_xyz=True)
        self.sa3 = PointnetSAModuleVotes(npoint=512, radius=0.8, nsample=16, mlp=[256, 128, 128, 256],
                                         use_xyz=True, normalize_xyz=True)
        self.sa4 = PointnetSAModuleVotes(npoint=256, radius=1.2, nsample=16, mlp=[256, 128, 128, 256],
                                         use_xyz=True, normalize_xyz=True)
        for sa in (self.sa2, self.sa3, self.sa4):  # their input features take a gradient, their coordinates do not
            sa.csr_backward = True
        self.fp1 = PointnetFPModule(mlp=[256 + 256, 256, 256])
        self.fp2 = PointnetFPModule(mlp=[256 + 256, 256, 256])

    @staticmethod
    def _break_up_pc(pc):
        xyz = pc[..., :3].contiguous()
        # (B,C,N) like the reference, but as a transposed VIEW of one point-major copy: the fused SA layer
        # gathers whole point rows and takes it back without a second 169 MB transpose
        features = pc[..., 3:].contiguous().transpose(1, 2) if pc.size(-1) > 3 else None
        return xyz, features

    GEOMETRY_KEYS = ("sa1", "sa2", "sa3", "sa4", "fp1", "fp2")

    @torch.no_grad()
    def compute_geometry(self, point_clouds):
        """Everything in the backbone that depends only on the input coordinates (4x FPS + gather + ball query,
        2x three_nn): (B,N,3+C) -> {layer: tuple of tensors}.  No weights, no features, no gradient — a step
        driver can run it for the NEXT batch on a side stream while the dense layers of the current one run
        (SURVEY.md §7 hard part 2c); pass the result back as data_dict["backbone_geometry"]."""
        xyz = point_clouds[..., :3].contiguous()
        g = {}
        for name in ("sa1", "sa2", "sa3", "sa4"):
            g[name] = getattr(self, name).compute_geometry(xyz, fps_ordered=(name != "sa1" and FPS_PREFIX_HINT))
            xyz = g[name][1]
        g["fp1"] = PointnetFPModule.compute_geometry(g["sa3"][1], g["sa4"][1])
        g["fp2"] = PointnetFPModule.compute_geometry(g["sa2"][1], g["sa3"][1])
        return g

    def forward(self, data_dict):
        rows_bf16 = None
        if "k/xyz" in data_dict and "k/feat_bf" in data_dict:
            # ... and handed the feature channels over as bf16 rows (input_pipeline.compress_cloud / prepare_batch(feat_bf16=True))
            xyz, features, rows_bf16 = data_dict["k/xyz"], None, (data_dict["k/feat_bf"], int(data_dict["k/feat_c"]))
        elif "k/xyz" in data_dict and "k/feat_pm" in data_dict:
            # the loader already split the cloud (grounding_step.prepare_batch): no 173 MB copy inside the step
            xyz, features = data_dict["k/xyz"], data_dict["k/feat_pm"].transpose(1, 2)
        else:
            xyz, features = self._break_up_pc(data_dict["point_clouds"])
        geo = data_dict.get("backbone_geometry") or {}
        xyz, features, fps_inds = self.sa1(xyz, features, geometry=geo.get("sa1"), feat_rows_bf16=rows_bf16)
        data_dict["sa1_inds"], data_dict["sa1_xyz"], data_dict["sa1_features"] = fps_inds, xyz, features
        xyz, features, fps_inds = self.sa2(xyz, features, geometry=geo.get("sa2"))
        data_dict["sa2_inds"], data_dict["sa2_xyz"], data_dict["sa2_features"] = fps_inds, xyz, features
        xyz, features, fps_inds = self.sa3(xyz, features, geometry=geo.get("sa3"))
        data_dict["sa3_xyz"], data_dict["sa3_features"] = xyz, features
        xyz, features, fps_inds = self.sa4(xyz, features, geometry=geo.get("sa4"))
        data_dict["sa4_xyz"], data_dict["sa4_features"] = xyz, features

        features = self.fp1(data_dict["sa3_xyz"], data_dict["sa4_xyz"], data_dict["sa3_features"],
                            data_dict["sa4_features"], geometry=geo.get("fp1"))
        features = self.fp2(data_dict["sa2_xyz"], data_dict["sa3_xyz"], data_dict["sa2_features"], features,
                            geometry=geo.get("fp2"))
        data_dict["fp2_features"] = features
        data_dict["fp2_xyz"] = data_dict["sa2_xyz"]
        num_seed = data_dict["fp2_xyz"].shape[1]
        data_dict["fp2_inds"] = data_dict["sa1_inds"][:, 0:num_seed]  # indices into the input cloud
        return data_dict


class VotingModule(nn.Module):
    """seed (xyz, features) -> vote (xyz + offset, features + residual)."""

    def __init__(self, vote_factor, seed_feature_dim):
        super().__init__()
        self.vote_factor = vote_factor
        self.in_dim = seed_feature_dim
        self.out_dim = self.in_dim
        self.conv1 = nn.Conv1d(self.in_dim, self.in_dim, 1)
        self.conv2 = nn.Conv1d(self.in_dim, self.in_dim, 1)
        self.conv3 = nn.Conv1d(self.in_dim, (3 + self.out_dim) * self.vote_factor, 1)
        self.bn1 = nn.BatchNorm1d(self.in_dim)
        self.bn2 = nn.BatchNorm1d(self.in_dim)
        self.fused = True  # csrc/rows_mlp.hip on CUDA tensors; False = the reference's Conv1d / BatchNorm1d / ReLU sequence

    def _layers(self):
        return [(self.conv1.weight, self.conv1.bias, self.bn1), (self.conv2.weight, self.conv2.bias, self.bn2),
                (self.conv3.weight, self.conv3.bias, None)]

    def forward_normalized(self, seed_xyz, seed_features):
        """forward() followed by jointnet.py:148-149's `features / |features|_2` with the whole epilogue in one kernel
        (csrc/glue.hip: vote_epilogue); None when the fused path does not apply (the caller then runs forward())."""
        if not (self.fused and seed_features.is_cuda and self.vote_factor == 1):
            return None
        B, num_seed = seed_xyz.shape[:2]
        seed_pm = seed_features.float().transpose(1, 2).contiguous()
        X = seed_pm.view(B * num_seed, self.in_dim)
        if not row_mlp.supported(X, self._layers()):
            return None
        net = row_mlp.row_stack(X, self._layers(), keep_pad=True)           # (R, 320): [offset 3 | residual C | 0]
        vote_xyz, vote_features = glue.vote_epilogue(seed_xyz.float(), seed_pm, net)
        return vote_xyz, vote_features.transpose(2, 1)                      # (B,C,num_vote) view of point-major data

    def forward(self, seed_xyz, seed_features):
        B, num_seed = seed_xyz.shape[:2]
        num_vote = num_seed * self.vote_factor
        if self.fused and seed_features.is_cuda:
            seed_pm = seed_features.float().transpose(1, 2).contiguous()  # (B,S,C): free when the producer was point-major
            X = seed_pm.view(B * num_seed, self.in_dim)
            layers = [(self.conv1.weight, self.conv1.bias, self.bn1), (self.conv2.weight, self.conv2.bias, self.bn2),
                      (self.conv3.weight, self.conv3.bias, None)]
            if row_mlp.supported(X, layers):
                net = row_mlp.row_stack(X, layers).view(B, num_seed, self.vote_factor, 3 + self.out_dim)
                vote_xyz = (seed_xyz.unsqueeze(2) + net[..., 0:3]).reshape(B, num_vote, 3)
                vote_features = (seed_pm.unsqueeze(2) + net[..., 3:]).reshape(B, num_vote, self.out_dim)
                return vote_xyz, vote_features.transpose(2, 1)  # (B,C,num_vote) view of point-major data
        net = F.relu(self.bn1(self.conv1(seed_features)))
        net = F.relu(self.bn2(self.conv2(net)))
        net = self.conv3(net).transpose(2, 1).reshape(B, num_seed, self.vote_factor, 3 + self.out_dim)
        vote_xyz = (seed_xyz.unsqueeze(2) + net[..., 0:3]).reshape(B, num_vote, 3)
        vote_features = seed_features.transpose(2, 1).unsqueeze(2) + net[..., 3:]
        vote_features = vote_features.reshape(B, num_vote, self.out_dim).transpose(2, 1).contiguous()
        return vote_xyz, vote_features


class StandardROIHeads(nn.Module):
    """2x (Conv1d 128 + BN + ReLU) then objectness / box / class / heading predictors."""

    def __init__(self, num_heading_bin, num_class, seed_feat_dim=256, use_kl_loss=False):
        super().__init__()
        self.num_heading_bin = num_heading_bin
        self.num_class = num_class
        self.use_kl_loss = use_kl_loss
        self.fused = True  # csrc/rows_mlp.hip on CUDA tensors; False = the reference's Conv1d / BatchNorm1d / ReLU sequence
        convs = [nn.Conv1d(128, 128, kernel_size=1), nn.BatchNorm1d(128), nn.ReLU(inplace=True),
                 nn.Conv1d(128, 128, kernel_size=1), nn.BatchNorm1d(128), nn.ReLU(inplace=True)]
        self.convs = nn.Sequential(*convs)
        self.objectness_predictor = nn.Conv1d(128, 2, kernel_size=1)
        if self.use_kl_loss:
            self.alpha_predictor = nn.Conv1d(128, 6, kernel_size=1)
            self.alpha_activation = nn.Sigmoid()
        self.box_predictor = nn.Conv1d(128, 6, kernel_size=1)
        if self.num_class:
            self.sem_cls_predictor = nn.Conv1d(128, num_class, kernel_size=1)
        self.heading_cls_predictor = nn.Conv1d(128, num_heading_bin, kernel_size=1)
        self.heading_reg_predictor = nn.Conv1d(128, num_heading_bin, kernel_size=1)
        for layer in convs:
            if isinstance(layer, nn.Conv1d):
                nn.init.kaiming_normal_(layer.weight, mode="fan_out", nonlinearity="relu")
                nn.init.constant_(layer.bias, 0)
        for predictor in (self.objectness_predictor, self.box_predictor):
            nn.init.normal_(predictor.weight, std=0.001)
            nn.init.constant_(predictor.bias, 0)

    def forward(self, ROI_features, data_dict):
        heads = [self.heading_reg_predictor, self.heading_cls_predictor, self.box_predictor, self.objectness_predictor]
        if self.num_class:
            heads.append(self.sem_cls_predictor)
        if self.fused and ROI_features.is_cuda and not self.use_kl_loss:
            B, C, K = ROI_features.shape
            X = ROI_features.float().transpose(1, 2).contiguous().view(B * K, C)  # free for a point-major producer
            layers = [(self.convs[0].weight, self.convs[0].bias, self.convs[1]),
                      (self.convs[3].weight, self.convs[3].bias, self.convs[4]),
                      (merge_adjacent([h.weight for h in heads]), merge_adjacent([h.bias for h in heads]), None)]
            if row_mlp.supported(X, layers):
                out = row_mlp.row_stack(X, layers, keep_pad=True)              # (R, 64): the 28 predictor channels + 0
                (heading_reg, hres, hcls, rois, obj, sem, omask, sarg) = glue.roi_split(
                    out.view(B, K, -1), self.num_heading_bin, self.num_class)
                data_dict["sem_cls_scores"], data_dict["heading_scores"] = sem, hcls
                data_dict["heading_residuals_normalized"], data_dict["heading_residuals"] = heading_reg, hres
                data_dict["rois"], data_dict["objectness_scores"] = rois, obj
                data_dict["bbox_mask"], data_dict["pred_bbox_sems"] = omask, sarg
                return data_dict
        x = self.convs(ROI_features)
        if self.use_kl_loss:
            data_dict["alpha"] = self.alpha_activation(self.alpha_predictor(x).permute(0, 2, 1)) * 0.1 - 0.05
        # The five 1x1 predictors read the same features: ONE convolution with the concatenated weights (the
        # parameters stay separate tensors with the reference's names), then column slices — 5x fewer GEMM /
        # bias / convolution-backward launches for outputs of 1..18 channels each.
        out = F.conv1d(x, torch.cat([h.weight for h in heads], 0), torch.cat([h.bias for h in heads], 0))
        out = out.permute(0, 2, 1)
        return self._split(out, heads, data_dict)

    def _split(self, out, heads, data_dict):
        parts = torch.split(out, [h.weight.shape[0] for h in heads], dim=-1)
        heading_reg = parts[0]
        if self.num_class:
            data_dict["sem_cls_scores"] = parts[4]
        data_dict["heading_scores"] = parts[1]
        data_dict["heading_residuals_normalized"] = heading_reg
        data_dict["heading_residuals"] = heading_reg * (np.pi / self.num_heading_bin)
        data_dict["rois"] = parts[2].exp()  # distances to the 6 faces
        data_dict["objectness_scores"] = parts[3]
        data_dict["bbox_mask"] = data_dict["objectness_scores"].argmax(-1)
        return data_dict


# corner signs of utils/box_util.py:361-385 (get_3d_box_batch): x = +-l/2, y = +-w/2, z = +-h/2
_CORNER_SIGNS = ((1, 1, 1), (1, -1, 1), (-1, -1, 1), (-1, 1, 1), (1, 1, -1), (1, -1, -1), (-1, -1, -1), (-1, 1, -1))


_SIGN_CACHE = {}


def box_corners(box_size, heading, center):
    """Device restatement of get_3d_box_batch (utils/box_util.py:361-385), which rotates with
    roty_batch (:324-338): corners = (signs * size/2) @ R^T + center, R = [[c,0,s],[0,1,0],[-s,0,c]].
    box_size (...,3), heading (...), center (...,3) -> (...,8,3)."""
    key = (box_size.device, box_size.dtype)
    if key not in _SIGN_CACHE:
        _SIGN_CACHE[key] = torch.tensor(_CORNER_SIGNS, dtype=box_size.dtype, device=box_size.device)
    signs = _SIGN_CACHE[key]
    local = signs * (box_size.unsqueeze(-2) * 0.5)  # (...,8,3)
    c, s = torch.cos(heading).unsqueeze(-1), torch.sin(heading).unsqueeze(-1)
    x, y, z = local[..., 0], local[..., 1], local[..., 2]
    rotated = torch.stack([c * x + s * z, y, -s * x + c * z], dim=-1)
    return rotated + center.unsqueeze(-2)


class _BoxDecode(torch.autograd.Function):
    """Fused decode_pred_box + get_3d_box_batch (csrc/box_decode.hip, vlp3d_box_decode_fwd/bwd):
    (vote_xyz (B,K,3), heading_scores (B,K,NH), heading_residuals (B,K,NH), rois (B,K,6)) ->
    (heading (B,K), size (B,K,3), centre (B,K,3), corners (B,K,8,3) [no gradient, as in the reference])."""

    @staticmethod
    def forward(ctx, vote_xyz, heading_scores, heading_residuals, rois):
        vote_xyz, heading_scores = vote_xyz.contiguous().float(), heading_scores.contiguous().float()
        heading_residuals, rois = heading_residuals.contiguous().float(), rois.contiguous().float()
        B, K, NH = heading_scores.shape
        dev = vote_xyz.device
        heading = torch.empty((B, K), dtype=torch.float32, device=dev)
        size = torch.empty((B, K, 3), dtype=torch.float32, device=dev)
        centre = torch.empty((B, K, 3), dtype=torch.float32, device=dev)
        corners = torch.empty((B, K, 8, 3), dtype=torch.float32, device=dev)
        cls = torch.empty((B, K), dtype=torch.int32, device=dev)
        _ext.call("vlp3d_box_decode_fwd", vote_xyz, heading_scores, heading_residuals, rois, B * K, NH, heading, size,
                  centre, corners, cls)
        ctx.save_for_backward(rois, heading, cls)
        ctx.nh = NH
        ctx.set_materialize_grads(False)  # unused outputs arrive as None (the kernels take NULL), not as zero fills
        ctx.mark_non_differentiable(corners)
        return heading, size, centre, corners

    @staticmethod
    def backward(ctx, d_heading, d_size, d_centre, _d_corners):
        rois, heading, cls = ctx.saved_tensors
        B, K = heading.shape
        d_rois = torch.empty_like(rois)
        nh = ctx.nh
        d_res = torch.empty((B, K, nh), dtype=torch.float32, device=rois.device)
        d_xyz = torch.empty((B, K, 3), dtype=torch.float32, device=rois.device)
        opt = lambda g: None if g is None else g.contiguous().float()
        _ext.call("vlp3d_box_decode_bwd", rois, heading, cls, opt(d_heading), opt(d_size), opt(d_centre), B * K, nh, d_rois,
                  d_res, d_xyz)
        return d_xyz, None, d_res, d_rois


class ProposalModule(nn.Module):
    """Vote clustering (an SA layer on the votes) + ROI heads + box decode."""

    def __init__(self, num_class, num_heading_bin, num_size_cluster, mean_size_arr, num_proposal, sampling,
                 seed_feat_dim=256, mask_box=False, use_kl_loss=False, use_vote_weight=False):
        super().__init__()
        self.num_class, self.num_heading_bin, self.num_size_cluster = num_class, num_heading_bin, num_size_cluster
        self.mean_size_arr = mean_size_arr
        self.num_proposal, self.sampling, self.seed_feat_dim = num_proposal, sampling, seed_feat_dim
        self.mask_box, self.use_kl_loss, self.use_vote_weight = mask_box, use_kl_loss, use_vote_weight
        self.fused_decode = True  # csrc/box_decode.hip; False = the op-by-op restatement below (host tests)
        self.vote_aggregation = PointnetSAModuleVotes(npoint=self.num_proposal, radius=0.3, nsample=16,
                                                      mlp=[self.seed_feat_dim, 128, 128, 128], use_xyz=True,
                                                      normalize_xyz=True)
        self.proposal = StandardROIHeads(num_heading_bin=num_heading_bin, num_class=num_class, seed_feat_dim=256,
                                         use_kl_loss=self.use_kl_loss)
        if self.use_vote_weight:
            self.votes_weight_predictor = nn.Sequential(nn.Conv1d(256, 128, kernel_size=1), nn.BatchNorm1d(128),
                                                        nn.PReLU(), nn.Conv1d(128, 1, kernel_size=1), nn.Sigmoid())

    def forward(self, xyz, features, data_dict):
        if self.use_vote_weight:
            data_dict["vote_weights"] = self.votes_weight_predictor(features)
            features = features * data_dict["vote_weights"]
        xyz, features, fps_inds = self.vote_aggregation(xyz, features)
        data_dict["aggregated_vote_xyz"] = xyz
        data_dict["aggregated_vote_features"] = features.permute(0, 2, 1).contiguous()
        data_dict["aggregated_vote_inds"] = fps_inds
        data_dict = self.proposal(features, data_dict)
        return self.decode_scores(data_dict)

    def decode_pred_box(self, data_dict):
        agg_xyz = data_dict["aggregated_vote_xyz"]
        if self.fused_decode and agg_xyz.is_cuda:  # one kernel each way instead of ~30 element-wise launches
            pred_heading, pred_box_size, pred_center, corners = _BoxDecode.apply(
                agg_xyz, data_dict["heading_scores"], data_dict["heading_residuals"], data_dict["rois"])
            data_dict["pred_heading"] = pred_heading
            if self.mask_box and self.training:
                pred_center, pred_box_size = self.mask(pred_center, pred_box_size)
                corners = box_corners(pred_box_size.detach(), pred_heading.detach(), pred_center.detach())
            data_dict["pred_size"] = pred_box_size
            data_dict["pred_center"] = pred_center
            data_dict["pred_bbox_corner"] = corners
            return data_dict
        heading_class = torch.argmax(data_dict["heading_scores"], -1)
        heading_residual = torch.gather(data_dict["heading_residuals"], 2, heading_class.unsqueeze(-1))
        rois = data_dict["rois"]
        pred_heading = heading_class.float() * (2.0 * np.pi / self.num_heading_bin) + heading_residual[..., 0]
        data_dict["pred_heading"] = pred_heading
        pred_box_size = rois[:, :, 0:3] + rois[:, :, 3:6]
        # centre = vote - rotz(heading)-rotated half-difference of the face distances (:113-120):
        # row-vector v @ R with R = [[c,-s,0],[s,c,0],[0,0,1]]
        half = (rois[:, :, 0:3] - rois[:, :, 3:6]) / 2
        c, s = torch.cos(pred_heading), torch.sin(pred_heading)
        off = torch.stack([half[..., 0] * c + half[..., 1] * s, -half[..., 0] * s + half[..., 1] * c, half[..., 2]], -1)
        pred_center = agg_xyz - off
        if self.mask_box and self.training:
            pred_center, pred_box_size = self.mask(pred_center, pred_box_size)
        data_dict["pred_size"] = pred_box_size
        data_dict["pred_center"] = pred_center
        data_dict["pred_bbox_corner"] = box_corners(pred_box_size.detach(), pred_heading.detach(),
                                                    pred_center.detach())
        return data_dict

    def decode_scores(self, data_dict):
        data_dict = self.decode_pred_box(data_dict)
        data_dict["pred_bbox_feature"] = data_dict["aggregated_vote_features"]
        data_dict["pred_bbox_mask"] = data_dict["bbox_mask"] if "bbox_mask" in data_dict else \
            data_dict["objectness_scores"].argmax(-1)
        if "pred_bbox_sems" not in data_dict:  # the fused ROI split already produced both arg-max masks
            data_dict["pred_bbox_sems"] = data_dict["sem_cls_scores"].argmax(-1)
        return data_dict

    def mask(self, pred_center, pred_box_size):
        """Randomly replace 30 % of the boxes (:146-165); train-time augmentation, off by default."""
        B, K, _ = pred_center.shape
        dev = pred_center.device
        m = torch.bernoulli(torch.full([B, K], 0.3, device=dev)).bool()[:, :, None]
        rc = torch.randn([B, K, 3], device=dev) / 2
        rs = 1 + torch.randn([B, K, 3], device=dev)
        return torch.where(m, rc, pred_center), torch.where(m, rs, pred_box_size)


class _RelationBias(torch.autograd.Function):
    """Fused pairwise-geometry bias MLP (csrc/relation_bias.hip): centre (B,K,3), packed params -> (B,4,K,K)."""
    SLAB_BLOCKS = int(os.environ.get("VLP3D_RELBIAS_BLOCKS", 256))  # workgroups of the backward kernel (one partial-gradient slab each)

    @staticmethod
    def forward(ctx, centre, params):
        centre = centre.contiguous().float()
        params = params.contiguous().float()
        B, K, _ = centre.shape
        out = torch.empty((B, 4, K, K), dtype=torch.float32, device=centre.device)
        _ext.call("vlp3d_relation_bias_fwd", centre, params, B, K, out)
        ctx.save_for_backward(centre, params)
        ctx.bf16_mma = int(mfma_linear.BF16_MMA)  # the step's bf16 timing configuration (read at forward: backward runs outside its context)
        return out

    @staticmethod
    def backward(ctx, dout):
        centre, params = ctx.saved_tensors
        B, K, _ = centre.shape
        n = params.numel()
        dparams = torch.empty_like(params)
        slabs = torch.empty((_RelationBias.SLAB_BLOCKS, n), dtype=torch.float32, device=centre.device)
        dout = dout.contiguous().float()
        bf = ctx.bf16_mma
        run = lambda: _ext.call("vlp3d_relation_bias_bwd", centre, params, dout, B, K, dparams, slabs,
                                _RelationBias.SLAB_BLOCKS, bf)
        q = _ext.slab_queue()
        if q is not None:  # only parameter gradients come out of this kernel: it runs with the other optimiser-only launches
            q.defer(run, (centre, params, dout, dparams, slabs))
            return None, dparams.view(-1)  # a view: contents arrive at the flush (see _lib.SlabReduceQueue)
        run()
        return None, dparams


def relation_bias(centre, fc):
    """fc = the reference's self_attn_fc[i] Sequential; returns its output on all pairs as (B,4,K,K)."""
    params = merge_adjacent([p.reshape(-1) for p in fc.parameters()])  # a view when the step driver's FlatParams packed them
    return _RelationBias.apply(centre.detach(), params)


class RelationModule(nn.Module):
    """2 layers of proposal self-attention with an additive pairwise-geometry bias."""

    def __init__(self, num_proposals=256, hidden_size=128, lang_num_size=300, det_channel=128, head=4, depth=2):
        super().__init__()
        self.use_box_embedding = True
        self.use_dist_weight_matrix = True
        self.use_obj_embedding = True
        self.fused_bias = True  # pairwise MLP in one HIP kernel; False = the reference's op sequence (host tests)
        self.num_proposals, self.hidden_size, self.depth = num_proposals, hidden_size, depth
        self.features_concat = nn.Sequential(nn.Conv1d(det_channel, hidden_size, 1), nn.BatchNorm1d(hidden_size),
                                             nn.PReLU(hidden_size), nn.Conv1d(hidden_size, hidden_size, 1))
        self.self_attn_fc = nn.ModuleList(
            nn.Sequential(nn.Linear(4, 32), nn.ReLU(), nn.LayerNorm(32), nn.Linear(32, 32), nn.ReLU(),
                          nn.LayerNorm(32), nn.Linear(32, 4)) for _ in range(depth))
        self.self_attn = nn.ModuleList(
            MultiHeadAttention(d_model=hidden_size, d_k=hidden_size // head, d_v=hidden_size // head, h=head)
            for _ in range(depth))
        self.bbox_embedding = nn.ModuleList(nn.Linear(27, hidden_size) for _ in range(depth))
        self.obj_embedding = nn.ModuleList(nn.Linear(128, hidden_size) for _ in range(depth))

    def forward(self, data_dict):
        pf = data_dict["pred_bbox_feature"]  # (B, K, det_channel) point-major
        fc = self.features_concat
        layers = [(fc[0].weight, fc[0].bias, fc[1])]
        if (self.fused_bias and pf.is_cuda and pf.dtype == torch.float32 and fc[2].weight.numel() == fc[1].num_features
                and fc[1].num_features % 64 == 0 and row_mlp.supported(pf.reshape(-1, pf.shape[-1]), layers)):
            # Conv1d -> BatchNorm1d -> PReLU on the rows kernels (the slope rides on the final activation), then the second
            # 1x1 convolution as a linear layer: no NCHW round trips, no MIOpen / BLAS launches
            h = row_mlp.row_stack(pf.reshape(-1, pf.shape[-1]), layers, final_slope=fc[2].weight)
            features = _linear(h, fc[3].weight.reshape(fc[3].weight.shape[0], -1), fc[3].bias).view(pf.shape[0], pf.shape[1], -1)
        else:
            features = fc(pf.permute(0, 2, 1)).permute(0, 2, 1)
        B, K = features.shape[:2]
        corners = data_dict["pred_bbox_corner"]

        # the multiview channels: columns 6.. of the raw cloud, or 3.. of the loader's point-major feature split
        if "k/feat_bf" in data_dict:
            src_pc, col0 = data_dict["k/feat_bf"], 3
        else:
            src_pc, col0 = (data_dict["k/feat_pm"], 3) if "k/feat_pm" in data_dict else (data_dict.get("point_clouds"), 6)
        fused_inputs = self.fused_bias and corners.is_cuda and src_pc is not None and src_pc.shape[-1] >= col0 + 128
        if fused_inputs:  # obj_feat, manual_bbox_feat and the corner mean in ONE launch (csrc/glue.hip), no gradient
            obj_feat, manual_bbox_feat, centre = glue.relation_inputs(
                src_pc, data_dict["seed_inds"], data_dict["aggregated_vote_inds"], corners, col0)
        # pairwise geometry (layer-independent): delta[b,i,j] = centre_j - centre_i, plus its norm
        centre = centre if fused_inputs else corners.mean(dim=-2)
        pair = None
        if not (self.fused_bias and centre.is_cuda):
            delta = centre[:, None, :, :] - centre[:, :, None, :]
            pair = torch.cat([delta, delta.pow(2).sum(-1, keepdim=True).sqrt()], dim=-1).detach()  # (B,K,K,4)

        # "multiview feature of the proposal's source point" (:98-113).  REFERENCE QUIRK, reproduced
        # exactly because trained checkpoints depend on it: the reference offsets the per-batch point
        # ids by b*128 (obj_feat.shape[1] AFTER its permute = the channel count, not N) and takes ROWS OF
        # 128 CONSECUTIVE ELEMENTS of the channel-major (B,128,N) copy, so "row id" is the flat element
        # range [id*128, id*128+128) of that layout.  Same values here, without the 164 MB copy.
        if not fused_inputs:
            pc = data_dict["point_clouds"]
            N = pc.shape[1]
            seed_inds = data_dict["seed_inds"].long()
            src = torch.gather(seed_inds, 1, data_dict["aggregated_vote_inds"].long())  # (B,K)
            row_id = src + torch.arange(B, device=src.device)[:, None] * 128
            flat = row_id.unsqueeze(-1) * 128 + torch.arange(128, device=src.device)  # (B,K,128) into (B,128,N)
            fb, rem = flat // (128 * N), flat % (128 * N)
            obj_feat = pc[fb, rem % N, 6 + rem // N]

            cmin, cmax = corners.min(dim=2)[0], corners.max(dim=2)[0]
            box_centre = (cmin + cmax) / 2
            manual_bbox_feat = torch.cat([box_centre, (corners - box_centre[:, :, None, :]).reshape(B, K, -1)], -1).to(features.dtype)

        dist_weights = None
        for i in range(self.depth):
            if self.fused_bias and centre.is_cuda:
                dist_weights = relation_bias(centre, self.self_attn_fc[i])  # (B,4,K,K) additive bias
            else:
                dist_weights = self.self_attn_fc[i](pair).permute(0, 3, 1, 2)
            # features + 0.1 * embedding as ONE element-wise launch (relation_module.py:117: two)
            features = torch.add(features, _linear(obj_feat, self.obj_embedding[i].weight, self.obj_embedding[i].bias), alpha=0.1)
            be = self.bbox_embedding[i]
            if glue.smallk_supported(manual_bbox_feat, be.weight) and not torch.is_autocast_enabled("cuda"):
                features = glue.small_linear(manual_bbox_feat, be.weight, be.bias, base=features)  # add folded in
            else:
                features = features + be(manual_bbox_feat)
            features = self.self_attn[i](features, features, features, attention_weights=dist_weights, way="add")

        data_dict["dist_weights"] = dist_weights
        data_dict["attention_matrix_way"] = "add"
        data_dict["bbox_feature"] = features
        return data_dict
