"""Set-abstraction, feature-propagation and vector-pool modules on stacked (ragged-batch) scenes.

Public surface = what callers of pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py rely on
(voxel_set_abstraction.py:149-162, pvrcnn_head.py:13, pointnet2_backbone.py): `build_local_aggregation_module`,
`StackSAModuleMSG`, `StackPointnetFPModule`, `VectorPoolLocalInterpolateModule`, `VectorPoolAggregationModule`,
`VectorPoolAggregationModuleMSG` -- constructor keywords, keyword-callable forwards, returned tuples, and the parameter
names a checkpoint holds (`mlps.{k}.{0,1,3,4,...}`, `mlp.*`, `separate_local_aggregation_layer.*`, `post_mlps.*`,
`layer_{k}.*`, `msg_post_mlps.*`).

Layout convention used throughout: a stacked feature table is (rows, C); the shared 1x1 stacks run on it as a
one-image batch, (1, C, rows[, nsample]) -- `_as_image` / `_as_rows` do the two reshapes.
"""
from typing import List, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils


# ------------------------------------------------------------------------------------------------------- helpers
def _conv_bn_relu(widths: Sequence[int], conv, norm, groups_first: int = 1) -> nn.Sequential:
    """[conv(k=1, no bias), norm, ReLU] for each consecutive pair of `widths` (Sequential slots 0,1,2, 3,4,5, ...)."""
    stack = []
    for k, (cin, cout) in enumerate(zip(widths[:-1], widths[1:])):
        extra = {"groups": groups_first} if (k == 0 and groups_first != 1) else {}
        stack += [conv(cin, cout, kernel_size=1, bias=False, **extra), norm(cout), nn.ReLU()]
    return nn.Sequential(*stack)


def _reset_parameters(module: nn.Module):
    """He-normal convolution weights, unit BatchNorm scale, zero shifts (what the reference's modules start from)."""
    for m in module.modules():
        if isinstance(m, (nn.Conv1d, nn.Conv2d)):
            nn.init.kaiming_normal_(m.weight)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)


def _as_image(table: torch.Tensor, trailing: int = 0) -> torch.Tensor:
    """(rows, C) -> (1, C, rows) [+ `trailing` singleton axes]; (rows, C, ns) -> (1, C, rows, ns)."""
    if table.dim() == 3:
        return table.permute(1, 0, 2).unsqueeze(0)
    img = table.t().unsqueeze(0)
    for _ in range(trailing):
        img = img.unsqueeze(-1)
    return img


def _as_rows(img: torch.Tensor) -> torch.Tensor:
    """(1, C, rows[, 1]) -> (rows, C)"""
    img = img.squeeze(0)
    if img.dim() == 3:
        img = img.squeeze(-1)
    return img.t()


def _pool_samples(img: torch.Tensor, how: str) -> torch.Tensor:
    """(1, C, rows, ns) -> (1, C, rows) over the sample axis."""
    window = [1, img.size(3)]
    if how == 'max_pool':
        return F.max_pool2d(img, kernel_size=window).squeeze(-1)
    if how == 'avg_pool':
        return F.avg_pool2d(img, kernel_size=window).squeeze(-1)
    raise NotImplementedError(how)


def _inverse_distance_weights(dist: torch.Tensor, floor=None) -> torch.Tensor:
    """(…, 3) l2 distances -> weights 1/(d + 1e-8), normalised over the three neighbours."""
    inv = 1.0 / (dist + 1e-8)
    total = inv.sum(dim=-1, keepdim=True)
    if floor is not None:
        total = torch.clamp_min(total, min=floor)
    return inv / total


# ------------------------------------------------------------------------------------------------------- factory
def build_local_aggregation_module(input_channels, config):
    """config.NAME in {'StackSAModuleMSG' (default), 'VectorPoolAggregationModuleMSG'} -> (module, output channels).
    As in the reference (:13-15) the MLP specs inside `config.MLPS` are extended IN PLACE with the input width."""
    kind = config.get('NAME', 'StackSAModuleMSG')
    if kind == 'StackSAModuleMSG':
        specs = config.MLPS
        for k, spec in enumerate(specs):
            specs[k] = [input_channels] + spec
        module = StackSAModuleMSG(radii=config.POOL_RADIUS, nsamples=config.NSAMPLE, mlps=specs, use_xyz=True,
                                  pool_method='max_pool')
        return module, sum(spec[-1] for spec in specs)
    if kind == 'VectorPoolAggregationModuleMSG':
        return VectorPoolAggregationModuleMSG(input_channels=input_channels, config=config), config.MSG_POST_MLPS[-1]
    raise NotImplementedError(kind)


# ------------------------------------------------------------------------------------------------------- SA / FP
class StackSAModuleMSG(nn.Module):
    """Multi-scale grouping around GIVEN centres.
    forward(xyz (N, 3), xyz_batch_cnt, new_xyz (M, 3), new_xyz_batch_cnt, features (N, C)) -> (new_xyz, (M, sum C_out))."""

    def __init__(self, *, radii: List[float], nsamples: List[int], mlps: List[List[int]], use_xyz: bool = True,
                 pool_method='max_pool'):
        super().__init__()
        if not (len(radii) == len(nsamples) == len(mlps)):
            raise AssertionError("radii, nsamples and mlps must have one entry per scale")
        self.groupers = nn.ModuleList(pointnet2_utils.QueryAndGroup(r, ns, use_xyz=use_xyz) for r, ns in zip(radii, nsamples))
        self.mlps = nn.ModuleList()
        for spec in mlps:
            if use_xyz:
                spec[0] += 3      # on the caller's list: the reference's callers read the widened spec back (:54-55)
            self.mlps.append(_conv_bn_relu(spec, nn.Conv2d, nn.BatchNorm2d))
        self.pool_method = pool_method
        self.init_weights()

    def init_weights(self):
        _reset_parameters(self)

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None, empty_voxel_set_zeros=True):
        scales = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            grouped, _ = grouper(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features)   # (M, C, nsample)
            scales.append(_as_rows(_pool_samples(mlp(_as_image(grouped)), self.pool_method)))
        return new_xyz, torch.cat(scales, dim=1)


class StackPointnetFPModule(nn.Module):
    """Feature propagation: three-NN inverse-distance interpolation of `known_feats` onto `unknown`, concatenated with
    `unknown_feats`, then a shared MLP.  forward(...) -> (N, mlp[-1])."""

    def __init__(self, *, mlp: List[int]):
        super().__init__()
        self.mlp = _conv_bn_relu(mlp, nn.Conv2d, nn.BatchNorm2d)

    def forward(self, unknown, unknown_batch_cnt, known, known_batch_cnt, unknown_feats=None, known_feats=None):
        dist, idx = pointnet2_utils.three_nn(unknown, unknown_batch_cnt, known, known_batch_cnt)
        carried = pointnet2_utils.three_interpolate(known_feats, idx, _inverse_distance_weights(dist))
        table = carried if unknown_feats is None else torch.cat([carried, unknown_feats], dim=1)
        return _as_rows(self.mlp(_as_image(table, trailing=1)))


# ------------------------------------------------------------------------------------------------------- vector pool
class VectorPoolLocalInterpolateModule(nn.Module):
    """Features at the cell centres of a small voxel grid around every centre, each interpolated from its three nearest
    support points (+ the 9 offsets to them when use_xyz), optionally through a shared MLP.
    forward(support_xyz, support_features, xyz_batch_cnt, new_xyz, new_xyz_grid_centers (M, G, 3), new_xyz_batch_cnt)
    -> (M * G, C_out)"""

    def __init__(self, mlp, num_voxels, max_neighbour_distance, nsample, neighbor_type, use_xyz=True,
                 neighbour_distance_multiplier=1.0, xyz_encoding_type='concat'):
        super().__init__()
        self.num_voxels = num_voxels
        self.num_total_grids = num_voxels[0] * num_voxels[1] * num_voxels[2]
        self.max_neighbour_distance = max_neighbour_distance
        self.neighbor_distance_multiplier = neighbour_distance_multiplier
        self.nsample = nsample
        self.neighbor_type = neighbor_type       # 1: ball, anything else: cube
        self.use_xyz = use_xyz
        self.xyz_encoding_type = xyz_encoding_type
        self.mlp = None
        if mlp is not None:
            if use_xyz and xyz_encoding_type == 'concat':
                mlp[0] += 9
            self.mlp = _conv_bn_relu(mlp, nn.Conv2d, nn.BatchNorm2d)
        self.num_avg_length_of_neighbor_idxs = 1000   # adaptive: neighbours per centre offered to the query's buffer

    def forward(self, support_xyz, support_features, xyz_batch_cnt, new_xyz, new_xyz_grid_centers, new_xyz_batch_cnt):
        with torch.no_grad():
            dist, idx, needed = pointnet2_utils.three_nn_for_vector_pool_by_two_step(
                support_xyz, xyz_batch_cnt, new_xyz, new_xyz_grid_centers, new_xyz_batch_cnt, self.max_neighbour_distance,
                self.nsample, self.neighbor_type, self.num_avg_length_of_neighbor_idxs, self.num_total_grids,
                self.neighbor_distance_multiplier)
        self.num_avg_length_of_neighbor_idxs = max(self.num_avg_length_of_neighbor_idxs, needed.item())
        cells = idx.shape[1]
        flat_idx = idx.view(-1, 3)
        unfilled = flat_idx[:, 0] == -1                       # a cell whose owner centre had no support point in reach
        flat_idx[unfilled] = 0
        weight = _inverse_distance_weights(dist, floor=1e-8).view(-1, 3)
        table = pointnet2_utils.three_interpolate(support_features, flat_idx, weight)           # (M * G, C)
        if self.use_xyz:
            if self.xyz_encoding_type != 'concat':
                raise NotImplementedError(self.xyz_encoding_type)
            neighbours = support_xyz[flat_idx.long()]                                           # (M * G, 3, 3)
            offsets = (new_xyz_grid_centers.reshape(-1, 1, 3) - neighbours).reshape(-1, 9)
            table = torch.cat((table, offsets), dim=-1)
        table[unfilled, :] = 0
        if self.mlp is not None:
            table = _as_rows(self.mlp(_as_image(table, trailing=1)))
        return table


class VectorPoolAggregationModule(nn.Module):
    """PV-RCNN++ vector-pool aggregation: per-cell local features of a voxel grid around every centre (interpolated,
    cell-averaged or first-point), one grouped 1x1 convolution that keeps the cells apart, then post MLPs.
    forward(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features) -> (new_xyz, (M, post_mlps[-1]))."""

    _KINDS = ('local_interpolation', 'voxel_avg_pool', 'voxel_random_choice')

    def __init__(self, input_channels, num_local_voxel=(3, 3, 3), local_aggregation_type='local_interpolation',
                 num_reduced_channels=30, num_channels_of_local_aggregation=32, post_mlps=(128,),
                 max_neighbor_distance=None, neighbor_nsample=-1, neighbor_type=0, neighbor_distance_multiplier=2.0):
        super().__init__()
        if local_aggregation_type not in self._KINDS:
            raise AssertionError(f"local_aggregation_type must be one of {self._KINDS}")
        self.num_local_voxel = num_local_voxel
        self.total_voxels = num_local_voxel[0] * num_local_voxel[1] * num_local_voxel[2]
        self.local_aggregation_type = local_aggregation_type
        self.input_channels = input_channels
        self.num_reduced_channels = input_channels if num_reduced_channels is None else num_reduced_channels
        self.num_channels_of_local_aggregation = num_channels_of_local_aggregation
        self.max_neighbour_distance = max_neighbor_distance
        self.neighbor_nsample = neighbor_nsample
        self.neighbor_type = neighbor_type
        self.local_interpolate_module = None
        xyz_channels = 3
        if local_aggregation_type == 'local_interpolation':
            self.local_interpolate_module = VectorPoolLocalInterpolateModule(
                mlp=None, num_voxels=num_local_voxel, max_neighbour_distance=max_neighbor_distance, nsample=neighbor_nsample,
                neighbor_type=neighbor_type, neighbour_distance_multiplier=neighbor_distance_multiplier)
            xyz_channels = 9
        per_cell_in = self.num_reduced_channels + xyz_channels
        width = self.total_voxels * num_channels_of_local_aggregation
        self.separate_local_aggregation_layer = _conv_bn_relu([per_cell_in * self.total_voxels, width], nn.Conv1d,
                                                              nn.BatchNorm1d, groups_first=self.total_voxels)
        self.post_mlps = _conv_bn_relu([width] + list(post_mlps), nn.Conv1d, nn.BatchNorm1d)
        self.num_mean_points_per_grid = 20     # adaptive: (point, cell) pairs per centre offered to the pooling buffer
        self.init_weights()

    def init_weights(self):
        _reset_parameters(self)

    def extra_repr(self) -> str:
        return (f'radius={self.max_neighbour_distance}, local_voxels={tuple(self.num_local_voxel)}, '
                f'local_aggregation_type={self.local_aggregation_type}, '
                f'num_c_reduction={self.input_channels}->{self.num_reduced_channels}, '
                f'num_c_local_aggregation={self.num_channels_of_local_aggregation}')

    # -- the two ways of producing the (M, cells * per_cell_in) vector ------------------------------------------
    def vector_pool_with_voxel_query(self, xyz, xyz_batch_cnt, features, new_xyz, new_xyz_batch_cnt):
        first_point_only = 0 if self.local_aggregation_type == 'voxel_avg_pool' else 1
        gx, gy, gz = self.num_local_voxel
        pooled, local_xyz, needed, per_cell = pointnet2_utils.vector_pool_with_voxel_query_op(
            xyz, xyz_batch_cnt, features, new_xyz, new_xyz_batch_cnt, gx, gy, gz, self.max_neighbour_distance,
            self.num_reduced_channels, 1, self.num_mean_points_per_grid, self.neighbor_nsample, self.neighbor_type,
            first_point_only)
        self.num_mean_points_per_grid = max(self.num_mean_points_per_grid, needed.item())
        rows = pooled.shape[0]
        per_cell_vec = torch.cat((local_xyz.view(rows, -1, 3), pooled.view(rows, -1, self.num_reduced_channels)), dim=-1)
        return per_cell_vec.view(rows, -1), per_cell

    @staticmethod
    def get_dense_voxels_by_center(point_centers, max_neighbour_distance, num_voxels):
        """Centres of the num_voxels = (nx, ny, nz) cells that tile the cube of half-edge max_neighbour_distance around each
        point -> (rows, nx * ny * nz, 3), x slowest.  The per-axis coordinates come from the same float32 `arange`
        (start -R + R/n, step 2R/n, end R - R/n + 1e-5) the reference uses, because the three-NN that consumes them is
        sensitive to their last bit."""
        R = max_neighbour_distance
        axes = [torch.arange(-R + R / n, R - R / n + 1e-5, 2 * R / n, device=point_centers.device) for n in num_voxels]
        gx, gy, gz = torch.meshgrid(*axes, indexing='ij')
        offsets = torch.stack((gx.reshape(-1), gy.reshape(-1), gz.reshape(-1)), dim=-1)
        return point_centers.unsqueeze(1) + offsets.unsqueeze(0)

    def vector_pool_with_local_interpolate(self, xyz, xyz_batch_cnt, features, new_xyz, new_xyz_batch_cnt):
        cell_centres = self.get_dense_voxels_by_center(new_xyz, self.max_neighbour_distance, self.num_local_voxel)
        per_cell = self.local_interpolate_module(
            support_xyz=xyz, support_features=features, xyz_batch_cnt=xyz_batch_cnt, new_xyz=new_xyz,
            new_xyz_grid_centers=cell_centres, new_xyz_batch_cnt=new_xyz_batch_cnt)              # (M * cells, C + 9)
        return per_cell.contiguous().view(-1, self.total_voxels * per_cell.shape[-1])

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, **kwargs):
        rows, channels = features.shape
        if channels % self.num_reduced_channels:
            raise AssertionError(f'the input channels ({channels}) should be an integral multiple of '
                                 f'num_reduced_channels({self.num_reduced_channels})')
        reduced = features.view(rows, -1, self.num_reduced_channels).sum(dim=1)   # channel groups folded by summation
        if self.local_aggregation_type == 'local_interpolation':
            vector = self.vector_pool_with_local_interpolate(xyz, xyz_batch_cnt, reduced, new_xyz, new_xyz_batch_cnt)
        else:
            vector, _ = self.vector_pool_with_voxel_query(xyz, xyz_batch_cnt, reduced, new_xyz, new_xyz_batch_cnt)
        out = self.post_mlps(self.separate_local_aggregation_layer(_as_image(vector)))
        return new_xyz, _as_rows(out)


class VectorPoolAggregationModuleMSG(nn.Module):
    """Several VectorPoolAggregationModule groups (config.GROUP_CFG_{k}, registered as `layer_{k}`) side by side, their
    outputs concatenated behind the centre coordinates and mixed by `msg_post_mlps`.
    forward(**kwargs of VectorPoolAggregationModule.forward) -> (new_xyz, (M, MSG_POST_MLPS[-1]))."""

    def __init__(self, input_channels, config):
        super().__init__()
        self.model_cfg = config
        self.num_groups = config.NUM_GROUPS
        width = 3                                                     # the centre coordinates lead the concatenation
        for k in range(self.num_groups):
            group = config[f'GROUP_CFG_{k}']
            self.add_module(f'layer_{k}', VectorPoolAggregationModule(
                input_channels=input_channels, num_local_voxel=group.NUM_LOCAL_VOXEL, post_mlps=group.POST_MLPS,
                max_neighbor_distance=group.MAX_NEIGHBOR_DISTANCE, neighbor_nsample=group.NEIGHBOR_NSAMPLE,
                local_aggregation_type=config.LOCAL_AGGREGATION_TYPE,
                num_reduced_channels=config.get('NUM_REDUCED_CHANNELS', None),
                num_channels_of_local_aggregation=config.NUM_CHANNELS_OF_LOCAL_AGGREGATION,
                neighbor_distance_multiplier=2.0))
            width += group.POST_MLPS[-1]
        self.msg_post_mlps = _conv_bn_relu([width] + list(config.MSG_POST_MLPS), nn.Conv1d, nn.BatchNorm1d)

    def forward(self, **kwargs):
        centres, per_group = None, []
        for k in range(self.num_groups):
            centres, feats = getattr(self, f'layer_{k}')(**kwargs)
            per_group.append(feats)
        table = torch.cat([centres] + per_group, dim=-1)
        return centres, _as_rows(self.msg_post_mlps(_as_image(table)))
