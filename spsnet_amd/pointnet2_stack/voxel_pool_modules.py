"""Voxel-RCNN's neighbour-voxel set abstraction on stacked scenes, under the reference's name
(pcdet/ops/pointnet2/pointnet2_stack/voxel_pool_modules.py: NeighborVoxelSAModuleMSG; caller voxelrcnn_head.py).

Per scale: point features go through `mlps_in` (Conv1d + BN) BEFORE grouping, the centred neighbour offsets through
`mlps_pos` (Conv2d + BN), the two are added, ReLU, pooled over the samples and finished by `mlps_out`
(Conv1d + BN + ReLU).  Parameter names (`mlps_in.{k}.*`, `mlps_pos.{k}.*`, `mlps_out.{k}.*`) are the checkpoint's.
"""
from typing import List

import torch
import torch.nn as nn

from . import voxel_query_utils
from .pointnet2_modules import _as_image, _as_rows, _pool_samples, _reset_parameters


class NeighborVoxelSAModuleMSG(nn.Module):
    """forward(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, new_coords (M, 4) = [b, x, y, z], features (N, C),
    voxel2point_indices (B, Z, Y, X)) -> (M, sum mlps[k][2])"""

    def __init__(self, *, query_ranges: List[List[int]], radii: List[float], nsamples: List[int], mlps: List[List[int]],
                 use_xyz: bool = True, pool_method='max_pool'):
        super().__init__()
        if not (len(query_ranges) == len(nsamples) == len(mlps)):
            raise AssertionError("query_ranges, nsamples and mlps must have one entry per scale")
        self.groupers = nn.ModuleList()
        self.mlps_in, self.mlps_pos, self.mlps_out = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for reach, radius, nsample, (c_in, c_mid, c_out) in zip(query_ranges, radii, nsamples, (m[:3] for m in mlps)):
            self.groupers.append(voxel_query_utils.VoxelQueryAndGrouping(reach, radius, nsample))
            self.mlps_in.append(nn.Sequential(nn.Conv1d(c_in, c_mid, kernel_size=1, bias=False), nn.BatchNorm1d(c_mid)))
            self.mlps_pos.append(nn.Sequential(nn.Conv2d(3, c_mid, kernel_size=1, bias=False), nn.BatchNorm2d(c_mid)))
            self.mlps_out.append(nn.Sequential(nn.Conv1d(c_mid, c_out, kernel_size=1, bias=False), nn.BatchNorm1d(c_out),
                                               nn.ReLU()))
        self.relu = nn.ReLU()
        self.pool_method = pool_method
        self.init_weights()

    def init_weights(self):
        _reset_parameters(self)

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, new_coords, features, voxel2point_indices):
        coords_bzyx = new_coords[:, [0, 3, 2, 1]].contiguous()      # callers pass [b, x, y, z]; the voxel table is (B, Z, Y, X)
        scales = []
        for grouper, pre, pos, post in zip(self.groupers, self.mlps_in, self.mlps_pos, self.mlps_out):
            embedded = _as_rows(pre(_as_image(features))).contiguous()                          # (N, c_mid)
            grouped, grouped_xyz, empty = grouper(coords_bzyx, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, embedded,
                                                  voxel2point_indices)
            hole = empty.view(-1, 1, 1)
            offsets = (grouped_xyz - new_xyz.unsqueeze(-1)).masked_fill(hole, 0)
            mixed = self.relu(_as_image(grouped.masked_fill(hole, 0)) + pos(_as_image(offsets)))   # (1, c_mid, M, ns)
            scales.append(_as_rows(post(_pool_samples(mixed, self.pool_method))))
        return torch.cat(scales, dim=1)
