"""Set-abstraction / feature-propagation modules on stacked scenes: mirror of
pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py (build_local_aggregation_module :9-27, StackSAModuleMSG :30-112,
StackPointnetFPModule :115-157) -- same constructor keywords, forward signatures and state_dict keys.
The VectorPool* modules (:160-470) sit on the vector-pool kernels, which are not built: constructing one raises."""
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


class _VectorPoolMissing(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError(f"{type(self).__name__}: the vector-pool kernels (vector_pool_gpu.cu) are not built")


class VectorPoolLocalInterpolateModule(_VectorPoolMissing):
    pass


class VectorPoolAggregationModule(_VectorPoolMissing):
    pass


class VectorPoolAggregationModuleMSG(_VectorPoolMissing):
    pass
