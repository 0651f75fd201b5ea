"""Autograd ops and grouping modules of the batch PointNet++ surface, backed by libspsnet_sa.

Public names, argument order, shapes, dtypes and return values follow the reference's
pcdet/ops/pointnet2/pointnet2_batch/pointnet2_utils.py (cited per op below) so that
callers such as pointnet2_modules.py, surface_feature.py:55, PAGNet_backbone.py:157 and
stability_generate/model.py:20 run unchanged.  The implementation is new: ops are declared
through one helper, outputs are allocated on the input's device with torch.empty / zeros,
index ops are marked non-differentiable, and QueryAndGroup uses the fused
`sps_query_and_group` kernel whenever no gradient is required.
"""
import os
from typing import Optional, Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

from . import pointnet2_batch_cuda as _ext

__all__ = [
    "FarthestPointSampling", "farthest_point_sample", "furthest_point_sample",
    "FurthestPointSamplingWithDist", "furthest_point_sample_with_dist",
    "GatherOperation", "gather_operation", "ThreeNN", "three_nn",
    "ThreeInterpolate", "three_interpolate", "GroupingOperation", "grouping_operation",
    "BallQuery", "ball_query", "BallQueryDilated", "ball_query_dilated",
    "QueryAndGroup", "QueryDilatedAndGroup", "GroupAll",
]

_FPS_TEMP_INIT = 1e10  # reference pointnet2_utils.py:26


# Gradients of the gather / group ops: the reference scatters with atomicAdd (sum order unspecified, low bits vary from
# run to run).  True, or torch.use_deterministic_algorithms(True), selects the fixed-order kernels instead
# (csrc/group_gather.hip: sps_index_add_deterministic; bit-identical to a sequential CPU loop).
DETERMINISTIC_BACKWARD = False
# group_with_index when gradients are wanted: one launch each way (_GroupConcat) instead of grouping_operation x 2, subtract,
# cat and their backward nodes (A/B switch for tools/train_step_time.py; same values, gradients within summation order)
GROUP_CONCAT_TRAINING = os.environ.get("SPS_GROUP_CONCAT_TRAINING", "1") != "0"


def _deterministic():
    return DETERMINISTIC_BACKWARD or torch.are_deterministic_algorithms_enabled()


def _new(like: torch.Tensor, shape, dtype, zero=False):
    return (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=like.device)


class FarthestPointSampling(Function):
    """xyz (B,N,3) f32, npoint -> (B,npoint) i32.  Reference :10-36."""

    @staticmethod
    def forward(ctx, xyz: torch.Tensor, npoint: int) -> torch.Tensor:
        assert xyz.is_contiguous()
        batch, n = xyz.shape[0], xyz.shape[1]
        picked = _new(xyz, (batch, npoint), torch.int32)
        running = torch.full((batch, n), _FPS_TEMP_INIT, dtype=torch.float32, device=xyz.device)
        _ext.farthest_point_sampling_wrapper(batch, n, npoint, xyz, running, picked)
        ctx.mark_non_differentiable(picked)
        return picked

    @staticmethod
    def backward(ctx, grad=None):
        return None, None


farthest_point_sample = furthest_point_sample = FarthestPointSampling.apply


class FurthestPointSamplingWithDist(Function):
    """dist (B,N,N) f32 pairwise distances, npoint -> (B,npoint) i32.  Reference :39-65."""

    @staticmethod
    def forward(ctx, dist: torch.Tensor, npoint: int) -> torch.Tensor:
        assert dist.is_contiguous()
        batch, n = dist.shape[0], dist.shape[1]
        picked = _new(dist, (batch, npoint), torch.int32)
        running = torch.full((batch, n), _FPS_TEMP_INIT, dtype=torch.float32, device=dist.device)
        _ext.furthest_point_sampling_with_dist_wrapper(batch, n, npoint, dist, running, picked)
        ctx.mark_non_differentiable(picked)
        return picked

    @staticmethod
    def backward(ctx, grad=None):
        return None, None


furthest_point_sample_with_dist = FurthestPointSamplingWithDist.apply


class GatherOperation(Function):
    """features (B,C,N), idx (B,M) i32 -> (B,C,M).  Reference :67-101."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert idx.is_contiguous()
        batch, chans, n = features.shape
        m = idx.shape[1]
        out = _new(features, (batch, chans, m), torch.float32)
        _ext.gather_points_wrapper(batch, chans, n, m, features, idx, out)
        ctx.save_for_backward(idx)
        ctx.src_shape = (batch, chans, n)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        batch, chans, n = ctx.src_shape
        grad_out = grad_out.contiguous()
        grad_features = _new(grad_out, (batch, chans, n), torch.float32, zero=True)
        if _deterministic():
            _ext.index_add_deterministic(grad_out, idx, grad_features)
        else:
            _ext.gather_points_grad_wrapper(batch, chans, n, idx.shape[1], grad_out, idx, grad_features)
        return grad_features, None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    """unknown (B,n,3), known (B,m,3) -> (dist (B,n,3) = sqrt of squared distance, idx (B,n,3) i32).
    Reference :104-133."""

    @staticmethod
    def forward(ctx, unknown: torch.Tensor, known: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        assert unknown.is_contiguous()
        assert known.is_contiguous()
        batch, n = unknown.shape[0], unknown.shape[1]
        m = known.shape[1]
        dist2 = _new(unknown, (batch, n, 3), torch.float32)
        idx = _new(unknown, (batch, n, 3), torch.int32)
        _ext.three_nn_wrapper(batch, n, m, unknown, known, dist2, idx)
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """features (B,C,m), idx (B,n,3) i32, weight (B,n,3) -> (B,C,n).  Reference :136-181."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert idx.is_contiguous()
        assert weight.is_contiguous()
        batch, chans, m = features.shape
        n = idx.shape[1]
        out = _new(features, (batch, chans, n), torch.float32)
        _ext.three_interpolate_wrapper(batch, chans, m, n, features, idx, weight, out)
        ctx.save_for_backward(idx, weight)
        ctx.m = m
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        batch, chans, n = grad_out.shape
        grad_out = grad_out.contiguous()
        grad_features = _new(grad_out, (batch, chans, ctx.m), torch.float32, zero=True)
        _ext.three_interpolate_grad_wrapper(batch, chans, n, ctx.m, grad_out, idx, weight, grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    """features (B,C,N), idx (B,M,ns) i32 -> (B,C,M,ns).  Reference :184-225."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert idx.is_contiguous()
        batch, chans, n = features.shape
        m, ns = idx.shape[1], idx.shape[2]
        out = _new(features, (batch, chans, m, ns), torch.float32)
        _ext.group_points_wrapper(batch, chans, n, m, ns, features, idx, out)
        ctx.save_for_backward(idx)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        batch, chans, m, ns = grad_out.shape
        grad_out = grad_out.contiguous()
        grad_features = _new(grad_out, (batch, chans, ctx.n), torch.float32, zero=True)
        if _deterministic():
            _ext.index_add_deterministic(grad_out, idx, grad_features)
        else:
            _ext.group_points_grad_wrapper(batch, chans, ctx.n, m, ns, grad_out, idx, grad_features)
        return grad_features, None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    """radius, nsample, xyz (B,N,3), new_xyz (B,M,3) -> idx (B,M,nsample) i32; rows of empty
    balls stay zero.  Reference :228-256."""

    @staticmethod
    def forward(ctx, radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
        assert new_xyz.is_contiguous()
        assert xyz.is_contiguous()
        batch, n = xyz.shape[0], xyz.shape[1]
        m = new_xyz.shape[1]
        idx = _new(xyz, (batch, m, nsample), torch.int32, zero=True)
        _ext.ball_query_wrapper(batch, n, m, radius, nsample, new_xyz, xyz, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, grad=None):
        return None, None, None, None


ball_query = BallQuery.apply


class BallQueryDilated(Function):
    """max_radius, min_radius, nsample, xyz, new_xyz -> idx (B,M,nsample) i32.  Reference :258-287."""

    @staticmethod
    def forward(ctx, max_radius: float, min_radius: float, nsample: int, xyz: torch.Tensor,
                new_xyz: torch.Tensor) -> torch.Tensor:
        assert new_xyz.is_contiguous()
        assert xyz.is_contiguous()
        batch, n = xyz.shape[0], xyz.shape[1]
        m = new_xyz.shape[1]
        idx = _new(xyz, (batch, m, nsample), torch.int32, zero=True)
        _ext.ball_query_dilated_wrapper(batch, n, m, max_radius, min_radius, nsample, new_xyz, xyz, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, grad=None):
        return None, None, None, None, None


ball_query_dilated = BallQueryDilated.apply


def _needs_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _group_with_index(xyz, new_xyz, features, idx, use_xyz):
    """Differentiable grouping: relative xyz (B,3,M,ns) stacked over grouped features (B,C,M,ns)."""
    rel = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
    rel = rel - new_xyz.transpose(1, 2).unsqueeze(-1)
    if features is None:
        assert use_xyz, "Cannot have not features and not use xyz as a feature!"
        return rel
    grouped = grouping_operation(features, idx)
    return torch.cat([rel, grouped], dim=1) if use_xyz else grouped


class _GroupConcat(Function):
    """_group_with_index as ONE launch forward (sps_group_concat) and one backward: the gradient's feature rows are scattered
    back to the points straight from the (B, 3+C, M, ns) tensor (no slice copy), new_xyz receives minus the sum of a
    centroid's relative-xyz rows.  xyz gets no gradient here (group_with_index keeps the op sequence when it wants one)."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, features, idx, use_xyz):
        out = _ext.group_concat(xyz.contiguous(), new_xyz.contiguous(), features.contiguous(), idx, use_xyz)
        ctx.save_for_backward(idx)
        ctx.n, ctx.use_xyz = xyz.shape[1], bool(use_xyz)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        off = 3 if ctx.use_xyz else 0
        d_new = d_feat = None
        if ctx.needs_input_grad[2]:
            d_feat = _ext.group_concat_grad_features(grad_out, idx, ctx.n, off)
        if ctx.needs_input_grad[1] and ctx.use_xyz:
            d_new = -grad_out[:, :3].sum(dim=3).transpose(1, 2)
        return None, d_new, d_feat, None, None


def group_with_index(xyz, new_xyz, features, idx, use_xyz=True):
    """QueryAndGroup.forward behind its ball query (reference :312-320) for neighbour indices that exist already -- both
    radii of a layer from one scan, or queries that ran while the layer's FPS was still sampling: one launch when no
    gradient is wanted, the differentiable op sequence otherwise."""
    if features is None:
        assert use_xyz, "Cannot have not features and not use xyz as a feature!"
    if xyz.is_cuda and not _needs_grad(xyz, new_xyz, features):
        feats = features.contiguous() if features is not None else None
        return _ext.group_concat(xyz.contiguous(), new_xyz.contiguous(), feats, idx, use_xyz)
    if (GROUP_CONCAT_TRAINING and xyz.is_cuda and features is not None and features.dtype == torch.float32
            and not xyz.requires_grad and not _deterministic()):
        return _GroupConcat.apply(xyz, new_xyz, features, idx, use_xyz)
    return _group_with_index(xyz, new_xyz, features, idx, use_xyz)


class QueryAndGroup(nn.Module):
    """Ball query + grouping: (xyz (B,N,3), new_xyz (B,M,3), features (B,C,N)) -> (B,3+C,M,ns)
    (or (B,C,M,ns) without use_xyz).  Reference :289-322."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: Optional[torch.Tensor] = None):
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
        if not _needs_grad(xyz, new_xyz, features):
            # inference: one fused launch pair writes the concatenated tensor directly
            feats = features.contiguous() if features is not None else None
            out, _ = _ext.query_and_group(self.radius, self.nsample, xyz.contiguous(), new_xyz.contiguous(),
                                          feats, self.use_xyz)
            return out
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        return _group_with_index(xyz, new_xyz, features, idx, self.use_xyz)


class QueryDilatedAndGroup(nn.Module):
    """Annulus query + grouping.  The constructor keeps the reference's argument names: it is
    built as QueryDilatedAndGroup(radius, min_radius, ...) (pointnet2_modules.py:189-190) and
    forwards (radius_in, radius_out) as (max_radius, min_radius).  Reference :324-359."""

    def __init__(self, radius_in: float, radius_out: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius_in, self.radius_out, self.nsample, self.use_xyz = radius_in, radius_out, nsample, use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: Optional[torch.Tensor] = None):
        idx = ball_query_dilated(self.radius_in, self.radius_out, self.nsample, xyz, new_xyz)
        return _group_with_index(xyz, new_xyz, features, idx, self.use_xyz)


class GroupAll(nn.Module):
    """Treat the whole cloud as one group: -> (B,3+C,1,N).  Reference :361-384."""

    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: Optional[torch.Tensor] = None):
        coords = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return coords
        feats = features.unsqueeze(2)
        return torch.cat([coords, feats], dim=1) if self.use_xyz else feats
