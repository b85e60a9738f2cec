"""Set-abstraction and feature-propagation modules with the reference's interface.

Mirrors lib/pointnet2/pointnet2_modules.py: PointnetSAModuleVotes :164-272 (the only SA variant the
grounding path instantiates: backbone_module.py:29-63, proposal_module_fcos.py:36-43) and
PointnetFPModule :356-416.  The MSG / LFP variants have no caller in jointnet/refnet and are out of
scope.
"""
import os
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import mfma_linear

from . import pointnet2_utils
from . import row_mlp
from . import sa_fused
from . import _lib as sa_fused_ext
from . import pytorch_utils as pt_utils


class PointnetSAModuleVotes(nn.Module):
    """FPS -> gather -> ball query -> group (+xyz normalise) -> SharedMLP -> pool over nsample.

    forward(xyz (B,N,3), features (B,C,N), inds=None) -> (new_xyz (B,npoint,3),
    new_features (B,mlp[-1],npoint), inds (B,npoint) i32 [, unique_cnt]).
    """

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None, nsample: int = None,
                 bn: bool = True, use_xyz: bool = True, pooling: str = "max", sigma: float = None,
                 normalize_xyz: bool = False, sample_uniformly: bool = False, ret_unique_cnt: bool = False):
        super().__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.pooling = pooling
        self.use_xyz = use_xyz
        self.sigma = self.radius / 2 if sigma is None else sigma
        self.normalize_xyz = normalize_xyz
        self.ret_unique_cnt = ret_unique_cnt
        if npoint is not None:
            self.grouper = pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz, ret_grouped_xyz=True,
                                                         normalize_xyz=normalize_xyz,
                                                         sample_uniformly=sample_uniformly,
                                                         ret_unique_cnt=ret_unique_cnt)
        else:
            self.grouper = pointnet2_utils.GroupAll(use_xyz, ret_grouped_xyz=True)
        mlp_spec = mlp
        if use_xyz and len(mlp_spec) > 0:
            mlp_spec[0] += 3  # the reference mutates the caller's list the same way (:207-208)
        self.mlp_module = pt_utils.SharedMLP(mlp_spec, bn=bn)
        # row-major fused path (group_rows kernel + GEMMs on (rows, C) matrices); `fused=False` restores the
        # literal op-by-op sequence of the reference (NCHW grouped tensor)
        # "mfma": hand-written matrix-core kernels (csrc/sa_mlp.hip); "rows": torch GEMM/BN on the gathered rows;
        # False: the reference's literal NCHW sequence
        ok = bool(bn and use_xyz and pooling == "max" and npoint is not None and not sample_uniformly
                  and not ret_unique_cnt)
        self.fused = "mfma" if ok else False
        # storage / MFMA type of the fused grouped MLP: None = follow autocast (bf16 inside an autocast region,
        # else fp32); torch.bfloat16 selects the bf16 kernels explicitly while the rest of the model stays fp32
        self.mlp_dtype = None
        self.compact = os.environ.get("VLP3D_SA_COMPACT", "1") != "0"
        # True (set by the backbone for sa2..sa4): also build the point -> rows map, and sum the gather layer's input
        # gradient per point through it instead of scattering it with float atomics (csrc/sa_gather_sum.hip)
        self.csr_backward = False
        # bf16 configuration: also emit the pooled output as bf16 rows (an attribute of the returned feature tensor) and read
        # such rows when the incoming feature tensor carries them (vlp3d_sa_pool_rows / bf16_io bit 1): the next level's gather
        # layer and its weight gradient then read 16-byte chunks of rows that are already what they would round them to
        self.pass_rows_bf16 = os.environ.get("VLP3D_SA_ROWS_BF16", "1") != "0"

    @torch.no_grad()
    def compute_geometry(self, xyz, fps_ordered=False):
        """The weight-independent part of the layer: (inds, new_xyz, ball-query idx).  Depends only on the
        coordinates, so a step driver may compute it ahead of time on a side stream (grounding_step.py).
        fps_ordered: xyz is the previous level's new_xyz (FPS samples in sampling order) — a hint, see _lib."""
        N = xyz.shape[1]
        if (xyz.is_cuda and xyz.dtype == torch.float32 and not fps_ordered and sa_fused_ext.FPS_PRUNED_MIN_N <= N <=
                sa_fused_ext.FPS_PRUNED_MAX_N and N >= sa_fused_ext.BALL_QUERY_GRID_MIN_N and self.npoint is not None
                and os.environ.get("VLP3D_BALL_QUERY") in (None, "", "sorted")):
            # a large first level (SA1: 40 000 points): ONE spatial sort serves both — the pruned FPS builds it, the ball
            # query reads it again (csrc/ball_query_sorted.hip: one launch instead of the grid form's six); same outputs
            xyz = xyz.contiguous()
            inds, ws = sa_fused_ext.furthest_point_sampling(xyz, self.npoint, "pruned", return_workspace=True)
            new_xyz = sa_fused_ext.gather_xyz(xyz, inds)
            idx = sa_fused_ext.ball_query_sorted(new_xyz, xyz, self.radius, self.nsample, ws)
        else:
            inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint, fps_ordered)
            new_xyz = sa_fused_ext.gather_xyz(xyz.contiguous(), inds)  # == gather_operation on the transposed cloud, one launch
            idx = pointnet2_utils.ball_query(self.radius, self.nsample, xyz, new_xyz)
        if self._use_compact(xyz):
            cmap = tuple(sa_fused_ext.sa_compact(idx, xyz.shape[1]))                       # + (rowptr, crow)
            if self.csr_backward and os.environ.get("VLP3D_SA_CSR", "1") != "0":
                return (inds, new_xyz, idx) + cmap + tuple(sa_fused_ext.sa_inverse(idx, xyz.shape[1], cmap))  # + (inv_start, inv_rows)
            return (inds, new_xyz, idx) + cmap
        return inds, new_xyz, idx

    def _use_compact(self, xyz):
        """Distinct-row evaluation of the grouped MLP (csrc/sa_compact.hip): bf16 configuration (39 % / 18 % / 72 % / 80 % of
        the padded rows are distinct at SA1 .. SA4 of the bench scenes)."""
        return (self.compact and self.fused == "mfma" and xyz.is_cuda and self.mlp_dtype == torch.bfloat16
                and self.nsample >= int(os.environ.get("VLP3D_SA_COMPACT_MIN_S", 16)))

    def _forward_rows(self, xyz, features, inds, geometry=None, feat_rows_bf16=None):
        """Same math as the reference sequence, on GEMM-ready rows: group_rows -> (linear, BN, ReLU) x L ->
        max over nsample.  BatchNorm over the (B*npoint*nsample) rows of a channel is exactly BatchNorm2d over
        (B, npoint, nsample); the first layer's weight columns are permuted to [features | xyz | 0]."""
        B, N, _ = xyz.shape
        M, S = self.npoint, self.nsample
        cmap = inv = None
        if geometry is not None:
            inds, new_xyz, idx = geometry[:3]
            cmap = tuple(geometry[3:5]) if len(geometry) >= 5 else None
            inv = tuple(geometry[5:7]) if len(geometry) >= 7 else None
        else:
            new_xyz = pointnet2_utils.gather_xyz(xyz, inds)  # gather_operation on the transposed cloud, without the transposes
            idx = pointnet2_utils.ball_query(self.radius, S, xyz, new_xyz)
        dtype = self.mlp_dtype or (torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda")
                                   else torch.float32)
        mlp_out = [layer.conv.weight.shape[0] for layer in self.mlp_module]
        feat_c = feat_rows = None

        def rows_readable(c):
            """bf16 rows of c channels can be read directly: the bf16 kernels' gather fast path (K <= 160) or its wide form
            (128 output channels, K <= 288, c % 8 == 0) — include/vlp3d.h, bf16_io bit 1."""
            K1 = (c + 4 + 15) // 16 * 16
            return (self.fused == "mfma" and dtype == torch.bfloat16 and sa_fused.supported(c, mlp_out, S, B * M * S, M)
                    and (K1 <= 160 or (mlp_out[0] == 128 and K1 <= 288 and c % 8 == 0)))

        if feat_rows_bf16 is not None and features is None:
            # the loader's bf16 copy of the input channels (B, N, round_up(C, 8)), no fp32 form: read as it is by the bf16
            # kernels; every other configuration widens it first
            rows, feat_c = feat_rows_bf16
            if rows_readable(feat_c):
                feat_pm = rows
            else:
                feat_pm, feat_c = rows[..., :feat_c].float().contiguous(), None
        else:
            feat_pm = features.transpose(1, 2).contiguous()  # (B,N,C): no copy when features is a point-major view
            if feat_rows_bf16 is not None and rows_readable(feat_rows_bf16[1]):
                # the previous level's pooled output as bf16 rows beside the fp32 tensor that carries the gradient
                feat_rows, feat_c = feat_rows_bf16
        if self.fused == "mfma" and sa_fused.supported(feat_c or feat_pm.shape[2], mlp_out, S, B * M * S, M):
            if cmap is None and geometry is None and self._use_compact(xyz):
                cmap = sa_fused_ext.sa_compact(idx, N)
                if self.csr_backward and os.environ.get("VLP3D_SA_CSR", "1") != "0" and torch.is_grad_enabled():
                    inv = sa_fused_ext.sa_inverse(idx, N, cmap)
            bf = dtype == torch.bfloat16
            want = bf and self.pass_rows_bf16 and mlp_out[2] % 8 == 0
            pooled = sa_fused.sa_mlp_pool(xyz, new_xyz, idx, feat_pm if feat_pm.dtype == torch.bfloat16 else feat_pm.float(),
                                          self.radius if self.normalize_xyz else 1.0, self.mlp_module, bf, cmap if bf else None,
                                          inv if bf else None, feat_c, feat_rows, want)
            rows_out = None
            if want:
                pooled, rows_out = pooled
            feats = pooled.transpose(1, 2)
            if rows_out is not None:
                # travels with the tensor OBJECT to the next level (Pointnet2Backbone hands `features` on unchanged): the same
                # values as bf16 rows (B, npoint, C) for its gather layer
                feats._vlp3d_rows_bf16 = rows_out
            return new_xyz, feats, inds
        x = pointnet2_utils.group_rows(xyz, new_xyz, idx, feat_pm, self.radius if self.normalize_xyz else 1.0, dtype)
        for i, layer in enumerate(self.mlp_module):
            w = layer.conv.weight[:, :, 0, 0]
            if i == 0:
                w = torch.cat([w[:, 3:], w[:, :3], w.new_zeros(w.shape[0], 1)], dim=1)
            x = F.linear(x, w)
            if x.is_cuda:
                mfma_linear.note_fallback(x.shape[0], w.shape[1], w.shape[0], "grouped MLP rows mode")
            bn = layer.bn.bn
            if bn.training and bn.track_running_stats:
                if bn.num_batches_tracked is not None:  # None: the step driver increments all counters at once
                    bn.num_batches_tracked.add_(1)
                factor = 1.0 / float(bn.num_batches_tracked) if bn.momentum is None else bn.momentum
            else:
                factor = 0.0
            x = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias,
                             bn.training or not bn.track_running_stats, factor, bn.eps)
            x = F.relu(x, inplace=True)
        pooled = x.view(B * M, S, x.shape[-1]).max(dim=1)[0]
        return new_xyz, pooled.view(B, M, -1).transpose(1, 2), inds  # (B,C,npoint) view of point-major data

    def forward(self, xyz, features=None, inds=None, geometry=None, feat_rows_bf16=None):
        # geometry and the gather kernels are fp32-only (like the reference's CHECK_IS_FLOAT); under
        # autocast the previous layer hands over bf16 activations
        xyz = xyz.float()
        if feat_rows_bf16 is not None:   # (bf16 rows (B, N, round_up(C, 8)), C) instead of `features`
            rows, c = feat_rows_bf16
            if self.fused and c % 4 == 0 and xyz.is_cuda:
                if geometry is None and inds is None:
                    inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
                return self._forward_rows(xyz, None, inds, geometry, feat_rows_bf16)
            features = rows[..., :c].float().transpose(1, 2)
        carried = getattr(features, "_vlp3d_rows_bf16", None) if features is not None else None
        features = features.float() if features is not None else None
        use_rows = self.fused and features is not None and features.shape[1] % 4 == 0 and xyz.is_cuda
        if (carried is not None and self.pass_rows_bf16 and use_rows and carried.is_contiguous()
                and tuple(carried.shape) == (features.shape[0], features.shape[2], features.shape[1]) and carried.shape[2] % 8 == 0):
            feat_rows_bf16 = (carried, carried.shape[2])
        if geometry is not None:
            assert use_rows, "precomputed geometry needs the fused path"
            return self._forward_rows(xyz, features, None, geometry, feat_rows_bf16)
        if inds is None:
            inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        else:
            assert inds.shape[1] == self.npoint
        if use_rows:
            return self._forward_rows(xyz, features, inds, None, feat_rows_bf16)
        features = features.contiguous() if features is not None else None
        xyz_flipped = xyz.transpose(1, 2).contiguous()
        new_xyz = (pointnet2_utils.gather_operation(xyz_flipped, inds).transpose(1, 2).contiguous()
                   if self.npoint is not None else None)

        unique_cnt = None
        if self.ret_unique_cnt:
            grouped_features, grouped_xyz, unique_cnt = self.grouper(xyz, new_xyz, features)
        else:
            grouped_features, grouped_xyz = self.grouper(xyz, new_xyz, features)

        new_features = self.mlp_module(grouped_features)  # (B, mlp[-1], npoint, nsample)
        if self.pooling == "max":
            new_features = F.max_pool2d(new_features, kernel_size=[1, new_features.size(3)])
        elif self.pooling == "avg":
            new_features = F.avg_pool2d(new_features, kernel_size=[1, new_features.size(3)])
        elif self.pooling == "rbf":
            rbf = torch.exp(-1 * grouped_xyz.pow(2).sum(1, keepdim=False) / (self.sigma ** 2) / 2)
            new_features = torch.sum(new_features * rbf.unsqueeze(1), -1, keepdim=True) / float(self.nsample)
        new_features = new_features.squeeze(-1)

        if self.ret_unique_cnt:
            return new_xyz, new_features, inds, unique_cnt
        return new_xyz, new_features, inds


class PointnetFPModule(nn.Module):
    """three_nn inverse-distance interpolation of `known_feats` onto `unknown`, concat, SharedMLP."""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn)
        self.fused = bool(bn)  # False: the reference's literal op sequence (three_interpolate, cat, conv / BN / ReLU)

    @staticmethod
    @torch.no_grad()
    def compute_geometry(unknown, known):
        """Weight-independent part: three_nn indices + inverse-distance weights (:393-397)."""
        if unknown.is_cuda and unknown.dtype == torch.float32:
            dist2, idx = sa_fused_ext.three_nn(unknown.contiguous(), known.contiguous())
            weight = sa_fused_ext.three_nn_weights(dist2)   # sqrt, reciprocal, sum, divide: one launch
            if os.environ.get("VLP3D_FP_CSR", "1") != "0":
                # known point -> references map for the atomic-free adjoint of the interpolation (row_mlp._FPRows.backward)
                return (idx, weight) + tuple(sa_fused_ext.sa_inverse(idx, known.shape[1]))
            return idx, weight
        dist, idx = pointnet2_utils.three_nn(unknown, known)
        dist_recip = 1.0 / (dist + 1e-8)
        return idx, dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)

    def forward(self, unknown, known, unknow_feats, known_feats, geometry=None):
        if self.fused and known is not None and unknow_feats is not None and known_feats.is_cuda:
            # point-major rows end to end (csrc/rows_mlp.hip): interpolate + concat in one launch, the SharedMLP as MFMA
            # products with BatchNorm / ReLU folded in; (B,C,n) in and out are VIEWS of point-major data (no transposes)
            known_pm = known_feats.float().transpose(1, 2).contiguous()      # free when the producer was point-major
            unknown_pm = unknow_feats.float().transpose(1, 2).contiguous()
            layers = [(layer.conv.weight, None, layer.bn.bn) for layer in self.mlp]
            B, n = unknown_pm.shape[:2]
            if row_mlp.fp_rows_supported(known_pm, unknown_pm) and (B * n) % 32 == 0 and all(
                    hasattr(layer, "bn") and layer.conv.bias is None for layer in self.mlp):
                geo = geometry if geometry is not None else self.compute_geometry(unknown, known)
                idx, weight = geo[:2]
                X = row_mlp.fp_rows(known_pm, unknown_pm, idx, weight, tuple(geo[2:4]) if len(geo) >= 4 else None)
                if row_mlp.supported(X, layers):
                    out = row_mlp.row_stack(X, layers)
                    return out.view(B, n, -1).transpose(1, 2)
        known_feats = known_feats.float().contiguous()
        if known is not None:
            if geometry is not None:
                idx, weight = geometry[:2]
            else:
                dist, idx = pointnet2_utils.three_nn(unknown, known)
                dist_recip = 1.0 / (dist + 1e-8)
                norm = torch.sum(dist_recip, dim=2, keepdim=True)
                weight = dist_recip / norm
            interpolated_feats = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:
            interpolated_feats = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        new_features = (torch.cat([interpolated_feats, unknow_feats], dim=1)
                        if unknow_feats is not None else interpolated_feats)
        return self.mlp(new_features.unsqueeze(-1)).squeeze(-1)
