"""Autograd surface of the stack ops under the reference's names.

Callers of pcdet/ops/pointnet2/pointnet2_stack/pointnet2_utils.py (voxel_set_abstraction.py:115,254,
pointnet2_backbone.py, the heads) use: ball_query, grouping_operation, QueryAndGroup, farthest_point_sample /
furthest_point_sample, stack_farthest_point_sample, three_nn, three_interpolate, three_nn_for_vector_pool_by_two_step,
vector_pool_with_voxel_query_op and the Function classes behind them -- same positional arguments, same returned tuples.
The work itself lives in `_ops` (one plain function per op); each class here only states which inputs carry gradients.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _ops


def _no_grads(n):
    return (None,) * n


class BallQuery(Function):
    """(radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt) -> (idx (M, nsample) scene-local int32 with
    zero rows for empty balls, empty_ball_mask (M,))."""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt):
        idx, empty = _ops.ball_query(radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt)
        ctx.mark_non_differentiable(idx, empty)
        return idx, empty

    @staticmethod
    def backward(ctx, *unused):
        return _no_grads(6)


ball_query = BallQuery.apply


class GroupingOperation(Function):
    """(features (N, C), features_batch_cnt, idx (M, nsample), idx_batch_cnt) -> (M, C, nsample); differentiable in
    `features`."""

    @staticmethod
    def forward(ctx, features, features_batch_cnt, idx, idx_batch_cnt):
        out = _ops.group(features, features_batch_cnt, idx, idx_batch_cnt)
        ctx.save_for_backward(idx, features_batch_cnt, idx_batch_cnt)
        ctx.n_rows = features.shape[0]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, feature_cnt, idx_cnt = ctx.saved_tensors
        return _ops.group_grad(grad_out, idx, idx_cnt, feature_cnt, ctx.n_rows), None, None, None


grouping_operation = GroupingOperation.apply


class QueryAndGroup(nn.Module):
    """Ball query, then the centred coordinates and the features of the neighbours:
    forward(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None) -> (grouped (M, [3 +] C, nsample), idx).
    Rows of empty balls are all zero (reference :111-158)."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None):
        _ops.require_rows(xyz, xyz_batch_cnt, "xyz")
        _ops.require_rows(new_xyz, new_xyz_batch_cnt, "new_xyz")
        if features is None and not self.use_xyz:
            raise AssertionError("QueryAndGroup without features needs use_xyz=True")
        idx, empty = ball_query(self.radius, self.nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt)
        hole = empty.view(-1, 1, 1)
        parts = []
        if self.use_xyz or features is None:
            offsets = grouping_operation(xyz, xyz_batch_cnt, idx, new_xyz_batch_cnt) - new_xyz.unsqueeze(-1)
            parts.append(offsets.masked_fill(hole, 0))
        if features is not None:
            parts.append(grouping_operation(features, xyz_batch_cnt, idx, new_xyz_batch_cnt).masked_fill(hole, 0))
        return (parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)), idx


class FarthestPointSampling(Function):
    """(xyz (B, N, 3), npoint) -> (B, npoint) int32."""

    @staticmethod
    def forward(ctx, xyz, npoint):
        picks = _ops.fps_batch(xyz, npoint)
        ctx.mark_non_differentiable(picks)
        return picks

    @staticmethod
    def backward(ctx, *unused):
        return _no_grads(2)


farthest_point_sample = furthest_point_sample = FarthestPointSampling.apply


class StackFarthestPointSampling(Function):
    """(xyz (N, 3), xyz_batch_cnt, npoint: int | list | tensor) -> (sum npoint,) int32 global rows."""

    @staticmethod
    def forward(ctx, xyz, xyz_batch_cnt, npoint):
        picks = _ops.fps_stack(xyz, xyz_batch_cnt, npoint)
        ctx.mark_non_differentiable(picks)
        return picks

    @staticmethod
    def backward(ctx, *unused):
        return _no_grads(3)


stack_farthest_point_sample = StackFarthestPointSampling.apply


class ThreeNN(Function):
    """(unknown (N, 3), unknown_batch_cnt, known (M, 3), known_batch_cnt) -> (dist (N, 3) l2, idx (N, 3) global rows)."""

    @staticmethod
    def forward(ctx, unknown, unknown_batch_cnt, known, known_batch_cnt):
        dist, idx = _ops.three_nn(unknown, unknown_batch_cnt, known, known_batch_cnt)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, *unused):
        return _no_grads(4)


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """(features (M, C), idx (N, 3), weight (N, 3)) -> (N, C); differentiable in `features`."""

    @staticmethod
    def forward(ctx, features, idx, weight):
        ctx.save_for_backward(idx, weight)
        ctx.m_rows = features.shape[0]
        return _ops.interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        return _ops.interpolate_grad(grad_out, idx, weight, ctx.m_rows), None, None


three_interpolate = ThreeInterpolate.apply


class ThreeNNForVectorPoolByTwoStep(Function):
    """(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_grid_centers (M, G, 3), new_xyz_batch_cnt, max_neighbour_distance,
    nsample, neighbor_type, avg_length_of_neighbor_idxs, num_total_grids, neighbor_distance_multiplier)
    -> (dist (M, G, 3) l2, idx (M, G, 3) global rows or -1, tensor(neighbours per centre needed))   [reference :301-358]"""

    @staticmethod
    def forward(ctx, support_xyz, xyz_batch_cnt, new_xyz, new_xyz_grid_centers, new_xyz_batch_cnt, max_neighbour_distance,
                nsample, neighbor_type, avg_length_of_neighbor_idxs, num_total_grids, neighbor_distance_multiplier):
        dist, idx, per_centre = _ops.local_three_nn(
            support_xyz, xyz_batch_cnt, new_xyz, new_xyz_grid_centers, new_xyz_batch_cnt,
            max_neighbour_distance * neighbor_distance_multiplier, nsample, neighbor_type, avg_length_of_neighbor_idxs,
            num_total_grids)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx, torch.tensor(per_centre)

    @staticmethod
    def backward(ctx, *unused):
        return _no_grads(11)


three_nn_for_vector_pool_by_two_step = ThreeNNForVectorPoolByTwoStep.apply


class VectorPoolWithVoxelQuery(Function):
    """(support_xyz, xyz_batch_cnt, support_features (N, C), new_xyz, new_xyz_batch_cnt, num_grid_x, num_grid_y,
    num_grid_z, max_neighbour_distance, num_c_out_each_grid, use_xyz, num_mean_points_per_grid=100, nsample=-1,
    neighbor_type=0, pooling_type=0)
    -> (cell means (M, G * c_each), mean local xyz (M, 3 G), tensor([points per centre needed]), points per cell (M, G));
    differentiable in `support_features`   [reference :361-446]"""

    @staticmethod
    def forward(ctx, support_xyz, xyz_batch_cnt, support_features, new_xyz, new_xyz_batch_cnt, num_grid_x, num_grid_y,
                num_grid_z, max_neighbour_distance, num_c_out_each_grid, use_xyz, num_mean_points_per_grid=100, nsample=-1,
                neighbor_type=0, pooling_type=0):
        cells = num_grid_x * num_grid_y * num_grid_z
        sums, xyz_sums, per_cell, triples, mean_needed = _ops.vector_pool(
            support_xyz, xyz_batch_cnt, support_features, new_xyz, new_xyz_batch_cnt, (num_grid_x, num_grid_y, num_grid_z),
            max_neighbour_distance, num_c_out_each_grid, use_xyz, num_mean_points_per_grid, nsample, neighbor_type,
            pooling_type)
        denom = torch.clamp_min(per_cell.unsqueeze(-1).float(), min=1e-6)
        pooled = (sums.view(-1, cells, num_c_out_each_grid) / denom).view(-1, cells * num_c_out_each_grid)
        local_xyz = (xyz_sums.view(-1, cells, 3) / denom).view(-1, cells * 3) if use_xyz else xyz_sums
        mean_needed = torch.Tensor([mean_needed]).int()
        ctx.save_for_backward(per_cell, triples)
        ctx.in_shape = tuple(support_features.shape)
        ctx.mark_non_differentiable(local_xyz, mean_needed, per_cell)
        return pooled, local_xyz, mean_needed, per_cell

    @staticmethod
    def backward(ctx, grad_pooled, *unused):
        per_cell, triples = ctx.saved_tensors
        n_rows, c_in = ctx.in_shape
        return (None, None, _ops.vector_pool_grad(grad_pooled, per_cell, triples, n_rows, c_in)) + _no_grads(12)


vector_pool_with_voxel_query_op = VectorPoolWithVoxelQuery.apply
