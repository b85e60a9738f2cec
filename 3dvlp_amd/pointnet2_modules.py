"""Set-abstraction and feature-propagation modules with the reference's interface.

Mirrors lib/pointnet2/pointnet2_modules.py: PointnetSAModuleVotes :164-272 (the only SA variant the
grounding path instantiates: backbone_module.py:29-63, proposal_module_fcos.py:36-43) and
PointnetFPModule :356-416.  The MSG / LFP variants have no caller in jointnet/refnet and are out of
scope.
"""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils
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

    def forward(self, xyz, features=None, inds=None):
        # geometry and the gather kernels are fp32-only (like the reference's CHECK_IS_FLOAT); under
        # autocast the previous layer hands over bf16 activations
        xyz = xyz.float()
        features = features.float() if features is not None else None
        xyz_flipped = xyz.transpose(1, 2).contiguous()
        if inds is None:
            inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        else:
            assert inds.shape[1] == self.npoint
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

    def forward(self, unknown, known, unknow_feats, known_feats):
        known_feats = known_feats.float()
        if known is not None:
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
