"""Set-abstraction / feature-propagation modules on stacked scenes: mirror of
pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py (build_local_aggregation_module :9-27, StackSAModuleMSG :30-112,
StackPointnetFPModule :115-157, VectorPoolLocalInterpolateModule :160-244, VectorPoolAggregationModule :247-420,
VectorPoolAggregationModuleMSG :423-470) -- same constructor keywords, forward signatures and state_dict keys."""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils


def _shared_mlp(spec: List[int]) -> nn.Sequential:
    layers = []
    for cin, cout in zip(spec[:-1], spec[1:]):
        layers += [nn.Conv2d(cin, cout, kernel_size=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU()]
    return nn.Sequential(*layers)


def build_local_aggregation_module(input_channels, config):
    name = config.get('NAME', 'StackSAModuleMSG')
    if name == 'StackSAModuleMSG':
        mlps = config.MLPS
        for k in range(len(mlps)):
            mlps[k] = [input_channels] + mlps[k]   # in place, like the reference (:13-15)
        layer = StackSAModuleMSG(radii=config.POOL_RADIUS, nsamples=config.NSAMPLE, mlps=mlps, use_xyz=True,
                                 pool_method='max_pool')
        return layer, sum(x[-1] for x in mlps)
    if name == 'VectorPoolAggregationModuleMSG':
        return VectorPoolAggregationModuleMSG(input_channels=input_channels, config=config), config.MSG_POST_MLPS[-1]
    raise NotImplementedError


class StackSAModuleMSG(nn.Module):
    """Multi-scale grouping around given centres; forward(xyz (N,3), xyz_batch_cnt, new_xyz (M,3), new_xyz_batch_cnt,
    features (N,C)) -> (new_xyz, new_features (M, sum C_out))."""

    def __init__(self, *, radii: List[float], nsamples: List[int], mlps: List[List[int]], use_xyz: bool = True,
                 pool_method='max_pool'):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz))
            if use_xyz:
                spec[0] += 3   # in place on the caller's list (:54-55)
            self.mlps.append(_shared_mlp(spec))
        self.pool_method = pool_method
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            if isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0)

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None, empty_voxel_set_zeros=True):
        pooled = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            grouped, _ = grouper(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features)   # (M, C, nsample)
            y = mlp(grouped.permute(1, 0, 2).unsqueeze(dim=0))                                 # (1, C', M, nsample)
            if self.pool_method == 'max_pool':
                y = F.max_pool2d(y, kernel_size=[1, y.size(3)]).squeeze(dim=-1)
            elif self.pool_method == 'avg_pool':
                y = F.avg_pool2d(y, kernel_size=[1, y.size(3)]).squeeze(dim=-1)
            else:
                raise NotImplementedError
            pooled.append(y.squeeze(dim=0).permute(1, 0))                                      # (M, C')
        return new_xyz, torch.cat(pooled, dim=1)


class StackPointnetFPModule(nn.Module):
    """Three-NN inverse-distance interpolation + shared MLP; -> (N, C_out)."""

    def __init__(self, *, mlp: List[int]):
        super().__init__()
        self.mlp = _shared_mlp(mlp)

    def forward(self, unknown, unknown_batch_cnt, known, known_batch_cnt, unknown_feats=None, known_feats=None):
        dist, idx = pointnet2_utils.three_nn(unknown, unknown_batch_cnt, known, known_batch_cnt)
        dist_recip = 1.0 / (dist + 1e-8)
        weight = dist_recip / torch.sum(dist_recip, dim=-1, keepdim=True)
        feats = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        if unknown_feats is not None:
            feats = torch.cat([feats, unknown_feats], dim=1)
        y = self.mlp(feats.permute(1, 0)[None, :, :, None])                                    # (1, C, N, 1)
        return y.squeeze(dim=0).squeeze(dim=-1).permute(1, 0)


class VectorPoolLocalInterpolateModule(nn.Module):
    """Features at the centres of a local voxel grid around every new_xyz, by three-NN inverse-distance interpolation of
    the support features (+ the 9 offsets to the three neighbours) (reference :160-244)."""

    def __init__(self, mlp, num_voxels, max_neighbour_distance, nsample, neighbor_type, use_xyz=True,
                 neighbour_distance_multiplier=1.0, xyz_encoding_type='concat'):
        super().__init__()
        self.num_voxels = num_voxels
        self.num_total_grids = self.num_voxels[0] * self.num_voxels[1] * self.num_voxels[2]
        self.max_neighbour_distance = max_neighbour_distance
        self.neighbor_distance_multiplier = neighbour_distance_multiplier
        self.nsample = nsample
        self.neighbor_type = neighbor_type
        self.use_xyz = use_xyz
        self.xyz_encoding_type = xyz_encoding_type
        if mlp is not None:
            if self.use_xyz:
                mlp[0] += 9 if self.xyz_encoding_type == 'concat' else 0
            self.mlp = _shared_mlp(mlp)
        else:
            self.mlp = None
        self.num_avg_length_of_neighbor_idxs = 1000

    def forward(self, support_xyz, support_features, xyz_batch_cnt, new_xyz, new_xyz_grid_centers, new_xyz_batch_cnt):
        """-> ((M * num_total_grids), C_out)"""
        with torch.no_grad():
            dist, idx, avg_len = pointnet2_utils.three_nn_for_vector_pool_by_two_step(
                support_xyz, xyz_batch_cnt, new_xyz, new_xyz_grid_centers, new_xyz_batch_cnt, self.max_neighbour_distance,
                self.nsample, self.neighbor_type, self.num_avg_length_of_neighbor_idxs, self.num_total_grids,
                self.neighbor_distance_multiplier)
        self.num_avg_length_of_neighbor_idxs = max(self.num_avg_length_of_neighbor_idxs, avg_len.item())
        dist_recip = 1.0 / (dist + 1e-8)
        weight = dist_recip / torch.clamp_min(torch.sum(dist_recip, dim=-1, keepdim=True), min=1e-8)
        empty_mask = idx.view(-1, 3)[:, 0] == -1
        idx.view(-1, 3)[empty_mask] = 0
        feats = pointnet2_utils.three_interpolate(support_features, idx.view(-1, 3), weight.view(-1, 3))
        feats = feats.view(idx.shape[0], idx.shape[1], -1)                                   # (M, G, C)
        if self.use_xyz:
            near = support_xyz[idx.view(-1, 3).long()].view(-1, 3, 3)
            local_xyz = (new_xyz_grid_centers.view(-1, 1, 3) - near).view(-1, idx.shape[1], 9)
            if self.xyz_encoding_type != 'concat':
                raise NotImplementedError
            feats = torch.cat((feats, local_xyz), dim=-1)                                      # (M, G, C + 9)
        new_features = feats.view(-1, feats.shape[-1])
        new_features[empty_mask, :] = 0
        if self.mlp is not None:
            new_features = self.mlp(new_features.permute(1, 0)[None, :, :, None]).squeeze(dim=0).squeeze(dim=-1).permute(1, 0)
        return new_features


class VectorPoolAggregationModule(nn.Module):
    """PV-RCNN++'s vector-pool aggregation: local voxel features (interpolated, averaged or first-point), a grouped
    convolution per voxel, post MLPs (reference :247-420)."""

    def __init__(self, input_channels, num_local_voxel=(3, 3, 3), local_aggregation_type='local_interpolation',
                 num_reduced_channels=30, num_channels_of_local_aggregation=32, post_mlps=(128,),
                 max_neighbor_distance=None, neighbor_nsample=-1, neighbor_type=0, neighbor_distance_multiplier=2.0):
        super().__init__()
        self.num_local_voxel = num_local_voxel
        self.total_voxels = self.num_local_voxel[0] * self.num_local_voxel[1] * self.num_local_voxel[2]
        self.local_aggregation_type = local_aggregation_type
        assert self.local_aggregation_type in ['local_interpolation', 'voxel_avg_pool', 'voxel_random_choice']
        self.input_channels = input_channels
        self.num_reduced_channels = input_channels if num_reduced_channels is None else num_reduced_channels
        self.num_channels_of_local_aggregation = num_channels_of_local_aggregation
        self.max_neighbour_distance = max_neighbor_distance
        self.neighbor_nsample = neighbor_nsample
        self.neighbor_type = neighbor_type
        if self.local_aggregation_type == 'local_interpolation':
            self.local_interpolate_module = VectorPoolLocalInterpolateModule(
                mlp=None, num_voxels=self.num_local_voxel, max_neighbour_distance=self.max_neighbour_distance,
                nsample=self.neighbor_nsample, neighbor_type=self.neighbor_type,
                neighbour_distance_multiplier=neighbor_distance_multiplier)
            num_c_in = (self.num_reduced_channels + 9) * self.total_voxels
        else:
            self.local_interpolate_module = None
            num_c_in = (self.num_reduced_channels + 3) * self.total_voxels
        num_c_out = self.total_voxels * self.num_channels_of_local_aggregation
        self.separate_local_aggregation_layer = nn.Sequential(
            nn.Conv1d(num_c_in, num_c_out, kernel_size=1, groups=self.total_voxels, bias=False),
            nn.BatchNorm1d(num_c_out), nn.ReLU())
        post, c_in = [], num_c_out
        for width in post_mlps:
            post += [nn.Conv1d(c_in, width, kernel_size=1, bias=False), nn.BatchNorm1d(width), nn.ReLU()]
            c_in = width
        self.post_mlps = nn.Sequential(*post)
        self.num_mean_points_per_grid = 20
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv1d)):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0)

    def extra_repr(self) -> str:
        return (f'radius={self.max_neighbour_distance}, local_voxels=({self.num_local_voxel}, '
                f'local_aggregation_type={self.local_aggregation_type}, '
                f'num_c_reduction={self.input_channels}->{self.num_reduced_channels}, '
                f'num_c_local_aggregation={self.num_channels_of_local_aggregation}')

    def vector_pool_with_voxel_query(self, xyz, xyz_batch_cnt, features, new_xyz, new_xyz_batch_cnt):
        pooling_type = 0 if self.local_aggregation_type == 'voxel_avg_pool' else 1
        new_features, new_local_xyz, mean_pts, point_cnt_of_grid = pointnet2_utils.vector_pool_with_voxel_query_op(
            xyz, xyz_batch_cnt, features, new_xyz, new_xyz_batch_cnt, self.num_local_voxel[0], self.num_local_voxel[1],
            self.num_local_voxel[2], self.max_neighbour_distance, self.num_reduced_channels, 1,
            self.num_mean_points_per_grid, self.neighbor_nsample, self.neighbor_type, pooling_type)
        self.num_mean_points_per_grid = max(self.num_mean_points_per_grid, mean_pts.item())
        m = new_features.shape[0]
        new_local_xyz = new_local_xyz.view(m, -1, 3)
        new_features = new_features.view(m, -1, self.num_reduced_channels)
        return torch.cat((new_local_xyz, new_features), dim=-1).view(m, -1), point_cnt_of_grid

    @staticmethod
    def get_dense_voxels_by_center(point_centers, max_neighbour_distance, num_voxels):
        """(N, 3) -> (N, total_voxels, 3): centres of the local voxels, x slowest (reference :336-359)."""
        R, dev = max_neighbour_distance, point_centers.device
        axes = [torch.arange(-R + R / n, R - R / n + 1e-5, 2 * R / n, device=dev) for n in num_voxels]
        gx, gy, gz = torch.meshgrid(axes[0], axes[1], axes[2], indexing='ij')
        offsets = torch.cat((gx.contiguous().view(-1, 1), gy.contiguous().view(-1, 1), gz.contiguous().view(-1, 1)), dim=-1)
        return point_centers[:, None, :] + offsets[None, :, :]

    def vector_pool_with_local_interpolate(self, xyz, xyz_batch_cnt, features, new_xyz, new_xyz_batch_cnt):
        centres = self.get_dense_voxels_by_center(new_xyz, self.max_neighbour_distance, self.num_local_voxel)
        voxel_features = self.local_interpolate_module.forward(
            support_xyz=xyz, support_features=features, xyz_batch_cnt=xyz_batch_cnt, new_xyz=new_xyz,
            new_xyz_grid_centers=centres, new_xyz_batch_cnt=new_xyz_batch_cnt)
        return voxel_features.contiguous().view(-1, self.total_voxels * voxel_features.shape[-1])

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, **kwargs):
        """-> (new_xyz, new_features (M, post_mlps[-1]))"""
        N, C = features.shape
        assert C % self.num_reduced_channels == 0, \
            f'the input channels ({C}) should be an integral multiple of num_reduced_channels({self.num_reduced_channels})'
        features = features.view(N, -1, self.num_reduced_channels).sum(dim=1)
        if self.local_aggregation_type in ['voxel_avg_pool', 'voxel_random_choice']:
            vector_features, _ = self.vector_pool_with_voxel_query(xyz, xyz_batch_cnt, features, new_xyz, new_xyz_batch_cnt)
        elif self.local_aggregation_type == 'local_interpolation':
            vector_features = self.vector_pool_with_local_interpolate(xyz, xyz_batch_cnt, features, new_xyz, new_xyz_batch_cnt)
        else:
            raise NotImplementedError
        y = self.separate_local_aggregation_layer(vector_features.permute(1, 0)[None, :, :])
        return new_xyz, self.post_mlps(y).squeeze(dim=0).permute(1, 0)


class VectorPoolAggregationModuleMSG(nn.Module):
    """Several VectorPoolAggregationModule groups + shared post MLPs (reference :423-470)."""

    def __init__(self, input_channels, config):
        super().__init__()
        self.model_cfg = config
        self.num_groups = self.model_cfg.NUM_GROUPS
        self.layers = []
        c_in = 0
        for k in range(self.num_groups):
            cur = self.model_cfg[f'GROUP_CFG_{k}']
            self.__setattr__(f'layer_{k}', VectorPoolAggregationModule(
                input_channels=input_channels, num_local_voxel=cur.NUM_LOCAL_VOXEL, post_mlps=cur.POST_MLPS,
                max_neighbor_distance=cur.MAX_NEIGHBOR_DISTANCE, neighbor_nsample=cur.NEIGHBOR_NSAMPLE,
                local_aggregation_type=self.model_cfg.LOCAL_AGGREGATION_TYPE,
                num_reduced_channels=self.model_cfg.get('NUM_REDUCED_CHANNELS', None),
                num_channels_of_local_aggregation=self.model_cfg.NUM_CHANNELS_OF_LOCAL_AGGREGATION,
                neighbor_distance_multiplier=2.0))
            c_in += cur.POST_MLPS[-1]
        c_in += 3  # use_xyz
        shared = []
        for width in self.model_cfg.MSG_POST_MLPS:
            shared += [nn.Conv1d(c_in, width, kernel_size=1, bias=False), nn.BatchNorm1d(width), nn.ReLU()]
            c_in = width
        self.msg_post_mlps = nn.Sequential(*shared)

    def forward(self, **kwargs):
        feats = []
        for k in range(self.num_groups):
            cur_xyz, cur_features = self.__getattr__(f'layer_{k}')(**kwargs)
            feats.append(cur_features)
        features = torch.cat((cur_xyz, torch.cat(feats, dim=-1)), dim=-1)
        return cur_xyz, self.msg_post_mlps(features.permute(1, 0)[None, :, :]).squeeze(dim=0).permute(1, 0)
