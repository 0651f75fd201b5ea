"""Autograd layer over the stack extension: mirror of pcdet/ops/pointnet2/pointnet2_stack/pointnet2_utils.py
(BallQuery :9-48, GroupingOperation :51-108, QueryAndGroup :111-158, FarthestPointSampling :161-185,
StackFarthestPointSampling :188-222, ThreeNN :225-256, ThreeInterpolate :259-298, ThreeNNForVectorPoolByTwoStep :301-358,
VectorPoolWithVoxelQuery :361-446) -- same names, argument order and return values.

Stacked layout: rows of all scenes concatenated, `*_batch_cnt` (batch_size,) int32 gives the rows per scene.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import pointnet2_stack_cuda as pointnet2


def _ints(shape, like, zero=False):
    return (torch.zeros if zero else torch.empty)(shape, dtype=torch.int32, device=like.device)


def _floats(shape, like, zero=False):
    return (torch.zeros if zero else torch.empty)(shape, dtype=torch.float32, device=like.device)


class BallQuery(Function):
    """radius, nsample, xyz (N,3), xyz_batch_cnt, new_xyz (M,3), new_xyz_batch_cnt -> (idx (M,nsample) int32 local to
    the scene, empty_ball_mask (M,) bool); rows of empty balls are all zero."""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt):
        assert new_xyz.is_contiguous() and new_xyz_batch_cnt.is_contiguous()
        assert xyz.is_contiguous() and xyz_batch_cnt.is_contiguous()
        B, M = xyz_batch_cnt.shape[0], new_xyz.shape[0]
        idx = _ints((M, nsample), xyz, zero=True)
        pointnet2.ball_query_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx)
        empty_ball_mask = idx[:, 0] == -1
        idx[empty_ball_mask] = 0
        ctx.mark_non_differentiable(idx, empty_ball_mask)
        return idx, empty_ball_mask

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None, None, None


ball_query = BallQuery.apply


class GroupingOperation(Function):
    """features (N,C), features_batch_cnt, idx (M,nsample), idx_batch_cnt -> (M,C,nsample)."""

    @staticmethod
    def forward(ctx, features, features_batch_cnt, idx, idx_batch_cnt):
        assert features.is_contiguous() and features_batch_cnt.is_contiguous()
        assert idx.is_contiguous() and idx_batch_cnt.is_contiguous()
        assert features.shape[0] == features_batch_cnt.sum(), \
            'features: %s, features_batch_cnt: %s' % (str(features.shape), str(features_batch_cnt))
        assert idx.shape[0] == idx_batch_cnt.sum(), 'idx: %s, idx_batch_cnt: %s' % (str(idx.shape), str(idx_batch_cnt))
        M, nsample = idx.size()
        N, C = features.size()
        B = idx_batch_cnt.shape[0]
        output = _floats((M, C, nsample), features)
        pointnet2.group_points_wrapper(B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, output)
        ctx.for_backwards = (B, N, idx, features_batch_cnt, idx_batch_cnt)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        B, N, idx, features_batch_cnt, idx_batch_cnt = ctx.for_backwards
        M, C, nsample = grad_out.size()
        grad_features = _floats((N, C), grad_out, zero=True)
        pointnet2.group_points_grad_wrapper(B, M, C, N, nsample, grad_out.contiguous(), idx, idx_batch_cnt,
                                            features_batch_cnt, grad_features)
        return grad_features, None, None, None


grouping_operation = GroupingOperation.apply


class QueryAndGroup(nn.Module):
    """Ball query + grouping on stacked scenes -> (new_features (M, 3+C | C, nsample), idx)."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None):
        assert xyz.shape[0] == xyz_batch_cnt.sum(), 'xyz: %s, xyz_batch_cnt: %s' % (str(xyz.shape), str(new_xyz_batch_cnt))
        assert new_xyz.shape[0] == new_xyz_batch_cnt.sum(), \
            'new_xyz: %s, new_xyz_batch_cnt: %s' % (str(new_xyz.shape), str(new_xyz_batch_cnt))
        idx, empty_ball_mask = ball_query(self.radius, self.nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt)
        grouped_xyz = grouping_operation(xyz, xyz_batch_cnt, idx, new_xyz_batch_cnt)  # (M, 3, nsample)
        grouped_xyz = grouped_xyz - new_xyz.unsqueeze(-1)
        grouped_xyz[empty_ball_mask] = 0
        if features is not None:
            grouped_features = grouping_operation(features, xyz_batch_cnt, idx, new_xyz_batch_cnt)
            grouped_features[empty_ball_mask] = 0
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz
        return new_features, idx


class FarthestPointSampling(Function):
    """xyz (B,N,3), npoint -> (B,npoint) int32 (the batch kernel; reference :161-185)."""

    @staticmethod
    def forward(ctx, xyz, npoint):
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        output = _ints((B, npoint), xyz)
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
        pointnet2.farthest_point_sampling_wrapper(B, N, npoint, xyz, temp, output)
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(ctx, a=None):
        return None, None


farthest_point_sample = furthest_point_sample = FarthestPointSampling.apply


class StackFarthestPointSampling(Function):
    """xyz (N,3), xyz_batch_cnt, npoint (int | list | tensor) -> (sum npoint,) int32 global row indices."""

    @staticmethod
    def forward(ctx, xyz, xyz_batch_cnt, npoint):
        assert xyz.is_contiguous() and xyz.shape[1] == 3
        batch_size = len(xyz_batch_cnt)
        if not isinstance(npoint, torch.Tensor):
            if not isinstance(npoint, list):
                npoint = [npoint for _ in range(batch_size)]
            npoint = torch.tensor(npoint, device=xyz.device).int()
        N = xyz.shape[0]
        temp = torch.full((N,), 1e10, dtype=torch.float32, device=xyz.device)
        output = _ints((int(npoint.sum().item()),), xyz)
        pointnet2.stack_farthest_point_sampling_wrapper(xyz, temp, xyz_batch_cnt, output, npoint.contiguous())
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None


stack_farthest_point_sample = StackFarthestPointSampling.apply


class ThreeNN(Function):
    """unknown (N,3), unknown_batch_cnt, known (M,3), known_batch_cnt -> (dist (N,3) l2, idx (N,3) global rows of known)."""

    @staticmethod
    def forward(ctx, unknown, unknown_batch_cnt, known, known_batch_cnt):
        assert unknown.dim() == 2 and unknown.shape[1] == 3
        assert known.dim() == 2 and known.shape[1] == 3
        assert len(unknown_batch_cnt) == len(known_batch_cnt)
        dist2 = unknown.new_zeros(unknown.shape)
        idx = unknown_batch_cnt.new_zeros(unknown.shape).int()
        pointnet2.three_nn_wrapper(unknown.contiguous(), unknown_batch_cnt.contiguous(), known.contiguous(),
                                   known_batch_cnt.contiguous(), dist2, idx)
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """features (M,C), idx (N,3), weight (N,3) -> (N,C)."""

    @staticmethod
    def forward(ctx, features, idx, weight):
        assert idx.shape[0] == weight.shape[0] and idx.shape[1] == weight.shape[1] == 3
        ctx.three_interpolate_for_backward = (idx, weight, features.shape[0])
        output = features.new_zeros((idx.shape[0], features.shape[1]))
        pointnet2.three_interpolate_wrapper(features.contiguous(), idx.contiguous(), weight.contiguous(), output)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, M = ctx.three_interpolate_for_backward
        grad_features = grad_out.new_zeros((M, grad_out.shape[1]))
        pointnet2.three_interpolate_grad_wrapper(grad_out.contiguous(), idx.contiguous(), weight.contiguous(), grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class ThreeNNForVectorPoolByTwoStep(Function):
    """Three nearest support points of every local grid centre, in two steps (reference :301-358): (1) per new_xyz, the
    support points inside max_neighbour_distance * multiplier (cube, or ball when neighbor_type == 1), packed into one
    stacked list -- retried with a larger buffer while it overflows; (2) three-NN of each grid centre inside that list.
    -> (dist (M, G, 3) l2, idx (M, G, 3) global rows or -1, avg_length_of_neighbor_idxs tensor)"""

    @staticmethod
    def forward(ctx, support_xyz, xyz_batch_cnt, new_xyz, new_xyz_grid_centers, new_xyz_batch_cnt, max_neighbour_distance,
                nsample, neighbor_type, avg_length_of_neighbor_idxs, num_total_grids, neighbor_distance_multiplier):
        num_new_xyz = new_xyz.shape[0]
        new_xyz_grid_dist2 = new_xyz_grid_centers.new_zeros(new_xyz_grid_centers.shape)
        new_xyz_grid_idxs = new_xyz_grid_centers.new_zeros(new_xyz_grid_centers.shape).int().fill_(-1)
        while True:
            num_max_sum_points = avg_length_of_neighbor_idxs * num_new_xyz
            stack_neighbor_idxs = new_xyz_grid_idxs.new_zeros(num_max_sum_points)
            start_len = new_xyz_grid_idxs.new_zeros(num_new_xyz, 2).int()
            cumsum = new_xyz_grid_idxs.new_zeros(1)
            pointnet2.query_stacked_local_neighbor_idxs_wrapper_stack(
                support_xyz.contiguous(), xyz_batch_cnt.contiguous(), new_xyz.contiguous(), new_xyz_batch_cnt.contiguous(),
                stack_neighbor_idxs.contiguous(), start_len.contiguous(), cumsum, avg_length_of_neighbor_idxs,
                max_neighbour_distance * neighbor_distance_multiplier, nsample, neighbor_type)
            found = int(cumsum[0].item())
            avg_length_of_neighbor_idxs = found // num_new_xyz + int(found % num_new_xyz > 0)
            if found <= num_max_sum_points:
                break
        stack_neighbor_idxs = stack_neighbor_idxs[:found]
        pointnet2.query_three_nn_by_stacked_local_idxs_wrapper_stack(
            support_xyz.contiguous(), new_xyz.contiguous(), new_xyz_grid_centers.contiguous(), new_xyz_grid_idxs,
            new_xyz_grid_dist2, stack_neighbor_idxs.contiguous(), start_len, num_new_xyz, num_total_grids)
        return torch.sqrt(new_xyz_grid_dist2), new_xyz_grid_idxs, torch.tensor(avg_length_of_neighbor_idxs)

    @staticmethod
    def backward(ctx, *grads):
        return (None,) * 11


three_nn_for_vector_pool_by_two_step = ThreeNNForVectorPoolByTwoStep.apply


class VectorPoolWithVoxelQuery(Function):
    """Per-grid-cell pooling of the support features around every new_xyz (reference :361-446): sums divided by the cell's
    point count (pooling_type 0) or the first point of a cell (1).
    -> (new_features (M, G * c_each), new_local_xyz (M, 3 G), num_mean_points_per_grid, point_cnt_of_grid (M, G))"""

    @staticmethod
    def forward(ctx, support_xyz, xyz_batch_cnt, support_features, new_xyz, new_xyz_batch_cnt, num_grid_x, num_grid_y,
                num_grid_z, max_neighbour_distance, num_c_out_each_grid, use_xyz, num_mean_points_per_grid=100, nsample=-1,
                neighbor_type=0, pooling_type=0):
        assert support_xyz.is_contiguous() and support_features.is_contiguous() and xyz_batch_cnt.is_contiguous()
        assert new_xyz.is_contiguous() and new_xyz_batch_cnt.is_contiguous()
        num_total_grids = num_grid_x * num_grid_y * num_grid_z
        num_c_out = num_c_out_each_grid * num_total_grids
        N, num_c_in = support_features.shape
        M = new_xyz.shape[0]
        assert num_c_in % num_c_out_each_grid == 0, \
            f'the input channels ({num_c_in}) should be an integral multiple of num_c_out_each_grid({num_c_out_each_grid})'
        while True:
            new_features = support_features.new_zeros((M, num_c_out))
            new_local_xyz = support_features.new_zeros((M, 3 * num_total_grids))
            point_cnt_of_grid = xyz_batch_cnt.new_zeros((M, num_total_grids))
            num_max_sum_points = num_mean_points_per_grid * M
            grouped_idxs = xyz_batch_cnt.new_zeros((num_max_sum_points, 3))
            num_cum_sum = pointnet2.vector_pool_wrapper(
                support_xyz, xyz_batch_cnt, support_features, new_xyz, new_xyz_batch_cnt, new_features, new_local_xyz,
                point_cnt_of_grid, grouped_idxs, num_grid_x, num_grid_y, num_grid_z, max_neighbour_distance, use_xyz,
                num_max_sum_points, nsample, neighbor_type, pooling_type)
            num_mean_points_per_grid = num_cum_sum // M + int(num_cum_sum % M > 0)
            if num_cum_sum <= num_max_sum_points:
                break
        grouped_idxs = grouped_idxs[:num_cum_sum]
        normalizer = torch.clamp_min(point_cnt_of_grid[:, :, None].float(), min=1e-6)
        new_features = (new_features.view(-1, num_total_grids, num_c_out_each_grid) / normalizer).view(-1, num_c_out)
        if use_xyz:
            new_local_xyz = (new_local_xyz.view(-1, num_total_grids, 3) / normalizer).view(-1, num_total_grids * 3)
        num_mean_points_per_grid = torch.Tensor([num_mean_points_per_grid]).int()
        nsample = torch.Tensor([nsample]).int()
        ctx.vector_pool_for_backward = (point_cnt_of_grid, grouped_idxs, N, num_c_in)
        ctx.mark_non_differentiable(new_local_xyz, num_mean_points_per_grid, nsample, point_cnt_of_grid)
        return new_features, new_local_xyz, num_mean_points_per_grid, point_cnt_of_grid

    @staticmethod
    def backward(ctx, grad_new_features, grad_local_xyz, grad_num_cum_sum, grad_point_cnt_of_grid):
        point_cnt_of_grid, grouped_idxs, N, num_c_in = ctx.vector_pool_for_backward
        grad_support_features = grad_new_features.new_zeros((N, num_c_in))
        if grouped_idxs.shape[0] > 0:
            pointnet2.vector_pool_grad_wrapper(grad_new_features.contiguous(), point_cnt_of_grid, grouped_idxs.contiguous(),
                                               grad_support_features)
        return (None, None, grad_support_features) + (None,) * 12


vector_pool_with_voxel_query_op = VectorPoolWithVoxelQuery.apply
