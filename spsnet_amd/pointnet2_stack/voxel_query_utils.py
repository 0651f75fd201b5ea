"""Voxel-indexed neighbour query on stacked scenes under the reference's names
(pcdet/ops/pointnet2/pointnet2_stack/voxel_query_utils.py: VoxelQuery / voxel_query, VoxelQueryAndGrouping; used by
voxel_pool_modules.NeighborVoxelSAModuleMSG, Voxel-RCNN's head)."""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _ops
from .pointnet2_utils import grouping_operation


class VoxelQuery(Function):
    """(max_range (z, y, x), radius, nsample, xyz (N, 3), new_xyz (M, 3), new_coords (M, 4) = [b, z, y, x],
    point_indices (B, Z, Y, X)) -> (idx (M, nsample) global rows with zero rows where nothing was found, mask (M,))."""

    @staticmethod
    def forward(ctx, max_range, radius, nsample, xyz, new_xyz, new_coords, point_indices):
        idx, empty = _ops.voxel_query(max_range, radius, nsample, xyz, new_xyz, new_coords, point_indices)
        ctx.mark_non_differentiable(idx, empty)
        return idx, empty

    @staticmethod
    def backward(ctx, *unused):
        return (None,) * 7


voxel_query = VoxelQuery.apply


class VoxelQueryAndGrouping(nn.Module):
    """forward(new_coords, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, voxel2point_indices)
    -> (grouped_features (M, C, nsample), grouped_xyz (M, 3, nsample) NOT centred, empty_ball_mask (M,))."""

    def __init__(self, max_range, radius: float, nsample: int):
        super().__init__()
        self.max_range, self.radius, self.nsample = max_range, radius, nsample

    def forward(self, new_coords, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, voxel2point_indices):
        _ops.require_rows(xyz, xyz_batch_cnt, "xyz")
        _ops.require_rows(new_coords, new_xyz_batch_cnt, "new_coords")
        scenes = xyz_batch_cnt.shape[0]
        rows, empty = voxel_query(self.max_range, self.radius, self.nsample, xyz, new_xyz, new_coords, voxel2point_indices)
        # The voxel table holds GLOBAL point rows, grouping wants scene-local ones.  Like the reference (:84-90) the queries
        # are taken to be split evenly over the scenes (view(batch, -1, nsample)); scene b's first global row is subtracted.
        first_row = torch.cumsum(xyz_batch_cnt, dim=0) - xyz_batch_cnt
        local = rows.view(scenes, -1, self.nsample) - first_row.view(scenes, 1, 1).to(rows.dtype)
        local = local.view(-1, self.nsample).masked_fill(empty.unsqueeze(1), 0).contiguous()
        grouped_xyz = grouping_operation(xyz, xyz_batch_cnt, local, new_xyz_batch_cnt)
        grouped_features = grouping_operation(features, xyz_batch_cnt, local, new_xyz_batch_cnt)
        return grouped_features, grouped_xyz, empty
