"""Drop-in for the reference's ``lib/pointnet2/pointnet2_utils.py`` — same public names, argument
order and autograd contract, running on the gfx950 kernels of libvlp3d_hip.so.

Reference symbols mirrored (file:line in /root/reference/lib/pointnet2/pointnet2_utils.py):
FurthestPointSampling :51, GatherOperation :81, ThreeNN :118, ThreeInterpolate :150,
GroupingOperation :207, BallQuery :258, QueryAndGroup :290, GroupAll :375.
Gradients flow only to the ``features`` operand of gather / group / interpolate; FPS, ball query
and three_nn return ``None`` grads (:74-75, :143-144, :283-284).

Known deviation: ThreeInterpolate.backward is the true adjoint; the reference calls its
forward kernel by mistake (interpolate.cpp:95).
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib as _ext

_ext.load()  # fail at import time, loudly, when the HIP library is missing


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint, prefix_hint=False):
        """xyz (B,N,3) f32 -> (B,npoint) i32 indices, first index 0.  prefix_hint (not in the reference): the caller expects
        xyz to be an earlier FPS's samples in sampling order — same result, usually without the sequential kernel."""
        out = _ext.furthest_point_sampling(xyz, npoint, prefix_hint=prefix_hint)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint) i32 -> (B,C,npoint)."""
        ctx.for_backwards = (idx, features.size(1), features.size(2))
        return _ext.gather_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        return _ext.gather_points_grad(grad_out.contiguous(), idx, N), None


gather_operation = GatherOperation.apply


class GatherXYZ(Function):
    """xyz (B,N,3), idx (B,M) i32 -> (B,M,3) = gather_operation(xyz^T, idx)^T on the point-major layout (no transposes);
    gradient to xyz (the vote coordinates of the proposal module's aggregation: proposal_module_fcos.py:76)."""

    @staticmethod
    def forward(ctx, xyz, idx):
        ctx.save_for_backward(idx)
        ctx.N = xyz.shape[1]
        return _ext.gather_xyz(xyz.contiguous(), idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return _ext.gather_xyz_grad(g.contiguous(), idx, ctx.N), None


gather_xyz = GatherXYZ.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        """unknown (B,n,3), known (B,m,3) -> (dist (B,n,3) [sqrt applied], idx (B,n,3) i32)."""
        dist2, idx = _ext.three_nn(unknown, known)
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


# False (default): the backward of three_interpolate is its true adjoint.  True: reproduce what the reference executes
# (the host bug of interpolate.cpp:95, see _lib.three_interpolate_grad_asshipped) for like-for-like training comparisons.
THREE_INTERPOLATE_GRAD_AS_SHIPPED = False


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        """features (B,c,m), idx (B,n,3) i32, weight (B,n,3) -> (B,c,n)."""
        ctx.three_interpolate_for_backward = (idx, weight, features.size(2))
        return _ext.three_interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, m = ctx.three_interpolate_for_backward
        if THREE_INTERPOLATE_GRAD_AS_SHIPPED:
            return _ext.three_interpolate_grad_asshipped(grad_out.contiguous(), idx, weight, m), None, None
        return _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight, m), None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint,nsample) i32 -> (B,C,npoint,nsample)."""
        ctx.for_backwards = (idx, features.size(2))
        return _ext.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, N = ctx.for_backwards
        return _ext.group_points_grad(grad_out.contiguous(), idx, N), None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        """radius, nsample, xyz (B,N,3), new_xyz (B,npoint,3) -> (B,npoint,nsample) i32."""
        out = _ext.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class GroupRows(Function):
    """Fused gather of QueryAndGroup in GEMM-ready row layout (csrc/group_rows.hip):
    rows (B*npoint*nsample, C+4) = [features[idx] | (xyz[idx] - new_xyz)/radius | 0]."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, idx, feat_pm, radius, out_dtype):
        ctx.save_for_backward(idx)
        ctx.dims = (xyz.shape[0], xyz.shape[1], feat_pm.shape[2], radius)
        return _ext.group_rows(xyz, new_xyz, idx, feat_pm, radius, out_dtype)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        B, N, C, radius = ctx.dims
        need = ctx.needs_input_grad
        dfeat, dxyz, dnew = _ext.group_rows_grad(grad_out.contiguous(), idx, B, N, C, radius, need[3], need[0],
                                                 need[1])
        return dxyz, dnew, None, dfeat, None, None


group_rows = GroupRows.apply


class QueryAndGroup(nn.Module):
    """Ball query + grouping + centre subtraction (+ /radius) + channel concat.

    Mirrors pointnet2_utils.py:290-372, including the return convention: a single tensor, or a tuple
    (new_features, grouped_xyz[, unique_cnt]) when ret_grouped_xyz / ret_unique_cnt are set.
    """

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False,
                 sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz
        self.sample_uniformly = sample_uniformly
        self.ret_unique_cnt = ret_unique_cnt
        if self.ret_unique_cnt:
            assert self.sample_uniformly

    def _resample_uniformly(self, idx):
        # :332-341 — host loop over every ball; not on the grounding hot path (never enabled there)
        unique_cnt = torch.zeros((idx.shape[0], idx.shape[1]))
        for b in range(idx.shape[0]):
            for r in range(idx.shape[1]):
                uniq = torch.unique(idx[b, r, :])
                n = uniq.shape[0]
                unique_cnt[b, r] = n
                pick = torch.randint(0, n, (self.nsample - n,), dtype=torch.long, device=idx.device)
                idx[b, r, :] = torch.cat((uniq, uniq[pick]))
        return unique_cnt

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        unique_cnt = self._resample_uniformly(idx) if self.sample_uniformly else None

        xyz_trans = xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)  # (B,3,npoint,nsample)
        grouped_xyz -= new_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_xyz /= self.radius

        if features is not None:
            grouped_features = grouping_operation(features, idx)
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz

        ret = [new_features]
        if self.ret_grouped_xyz:
            ret.append(grouped_xyz)
        if self.ret_unique_cnt:
            ret.append(unique_cnt)
        return ret[0] if len(ret) == 1 else tuple(ret)


class GroupAll(nn.Module):
    """Groups all points into one ball (pointnet2_utils.py:375-421)."""

    def __init__(self, use_xyz=True, ret_grouped_xyz=False):
        super().__init__()
        self.use_xyz = use_xyz
        # the reference never stores ret_grouped_xyz (:386-388) and then reads it (:418): it raises
        # AttributeError when reached; kept as an attribute here so GroupAll is usable.
        self.ret_grouped_xyz = ret_grouped_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is not None:
            grouped_features = features.unsqueeze(2)
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
        else:
            new_features = grouped_xyz
        if self.ret_grouped_xyz:
            return new_features, grouped_xyz
        return new_features
