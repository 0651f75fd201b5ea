"""Tier-2 oracle for the stacked (ragged-batch) ops: a CPU stand-in for the 14-function extension
`pointnet2_stack_cuda` (pcdet/ops/pointnet2/pointnet2_stack/src/pointnet2_api.cpp:12-31) over the C oracle.

TEST INFRASTRUCTURE ONLY.  Two uses:
  * build container only: `load_reference()` imports the REFERENCE's own pointnet2_stack Python (pointnet2_utils,
    pointnet2_modules, voxel_query_utils, voxel_pool_modules) over the stand-in, which is how tests/golden/stackmod_*.npz
    were generated (oracle/gen_golden.py) -- the reference never travels;
  * anywhere: `standin_module()` can be patched over spsnet_amd.pointnet2_stack.pointnet2_stack_cuda by a CPU test, so
    that the build's own host logic above the boundary is exercised without a GPU (tests/test_stack_modules_cpu.py).
The stand-in follows the extension's argument order; every function writes into the caller's buffers like the kernels.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

from . import oracle as O

REFERENCE_ROOT = "/root/reference"
_PKG = "pcdet.ops.pointnet2.pointnet2_stack"


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "pcdet", "ops", "pointnet2", "pointnet2_stack"))


def _n(t):
    assert t.device.type == "cpu" and t.is_contiguous(), "the stand-in works on contiguous CPU tensors"
    return t.detach().numpy()


def _put(dst, src):
    dst.detach().copy_(torch.from_numpy(np.ascontiguousarray(src)).view_as(dst))


def standin_module(name=_PKG + ".pointnet2_stack_cuda"):
    m = types.ModuleType(name)

    def ball_query_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx):
        _put(idx, O.stack_ball_query(radius, nsample, _n(xyz), _n(xyz_batch_cnt), _n(new_xyz), _n(new_xyz_batch_cnt)))
        return 1

    def voxel_query_wrapper(M, R1, R2, R3, nsample, radius, z_range, y_range, x_range, new_xyz, xyz, new_coords,
                            point_indices, idx):
        _put(idx, O.stack_voxel_query((z_range, y_range, x_range), radius, nsample, _n(xyz), _n(new_xyz), _n(new_coords),
                                      _n(point_indices)))
        return 1

    def farthest_point_sampling_wrapper(b, n, m, points, temp, idx):
        picks, running = O.fps(_n(points), m, temp=_n(temp), return_temp=True)
        _put(idx, picks)
        _put(temp, running)
        return 1

    def stack_farthest_point_sampling_wrapper(points, temp, xyz_batch_cnt, idx, num_sampled_points):
        _put(idx, O.stack_fps(_n(points), _n(xyz_batch_cnt), _n(num_sampled_points)))
        return 1

    def group_points_wrapper(B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, out):
        _put(out, O.stack_group_points(_n(features), _n(features_batch_cnt), _n(idx), _n(idx_batch_cnt)))
        return 1

    def group_points_grad_wrapper(B, M, C, N, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features):
        _put(grad_features, O.stack_group_points_grad(_n(grad_out), _n(idx), _n(idx_batch_cnt), _n(features_batch_cnt), N))
        return 1

    def three_nn_wrapper(unknown, unknown_batch_cnt, known, known_batch_cnt, dist2, idx):
        d2, i = O.stack_three_nn(_n(unknown), _n(unknown_batch_cnt), _n(known), _n(known_batch_cnt))
        _put(dist2, d2)
        _put(idx, i)
        return 1

    def three_interpolate_wrapper(features, idx, weight, out):
        _put(out, O.stack_three_interpolate(_n(features), _n(idx), _n(weight)))
        return 1

    def three_interpolate_grad_wrapper(grad_out, idx, weight, grad_features):
        _put(grad_features, O.stack_three_interpolate_grad(_n(grad_out), _n(idx), _n(weight), grad_features.shape[0]))
        return 1

    def query_stacked_local_neighbor_idxs_wrapper_stack(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt,
                                                        stack_neighbor_idxs, start_len, cumsum, avg_length_of_neighbor_idxs,
                                                        max_neighbour_distance, nsample, neighbor_type):
        lists, lens = O.vp_local_neighbors(_n(support_xyz), _n(xyz_batch_cnt), _n(new_xyz), _n(new_xyz_batch_cnt),
                                           max_neighbour_distance, nsample, neighbor_type)
        total = int(lens.sum())
        cumsum[0] = total
        if total > stack_neighbor_idxs.numel():
            return 0            # overflow: the caller retries with the reported size
        starts = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
        packed = np.zeros((stack_neighbor_idxs.numel(),), np.int32)
        for j, (s, ln) in enumerate(zip(starts, lens)):
            packed[s:s + ln] = lists[j, :ln]
        _put(stack_neighbor_idxs, packed)
        _put(start_len, np.stack([starts, lens.astype(np.int32)], axis=1))
        return 0

    def query_three_nn_by_stacked_local_idxs_wrapper_stack(support_xyz, new_xyz, new_xyz_grid_centers, new_xyz_grid_idxs,
                                                           new_xyz_grid_dist2, stack_neighbor_idxs, start_len, M,
                                                           num_total_grids):
        sl = _n(start_len)
        packed = _n(stack_neighbor_idxs)
        cap = max(1, int(sl[:, 1].max()) if sl.shape[0] else 1)
        lists = np.zeros((sl.shape[0], cap), np.int32)
        for j, (s, ln) in enumerate(sl):
            lists[j, :ln] = packed[s:s + ln]
        d2, idx = O.vp_three_nn_local(_n(support_xyz), _n(new_xyz_grid_centers), lists, sl[:, 1].copy())
        _put(new_xyz_grid_dist2, d2)
        _put(new_xyz_grid_idxs, idx)
        return 0

    def vector_pool_wrapper(support_xyz, xyz_batch_cnt, support_features, new_xyz, new_xyz_batch_cnt, new_features,
                            new_local_xyz, point_cnt_of_grid, grouped_idxs, num_grid_x, num_grid_y, num_grid_z,
                            max_neighbour_distance, use_xyz, num_max_sum_points, nsample, neighbor_type, pooling_type):
        total = num_grid_x * num_grid_y * num_grid_z
        c_each = new_features.shape[1] // total
        cum, nf, nl, cg, grouped = O.vp_pool(_n(support_xyz), _n(xyz_batch_cnt), _n(support_features), _n(new_xyz),
                                             _n(new_xyz_batch_cnt), (num_grid_x, num_grid_y, num_grid_z),
                                             max_neighbour_distance, c_each, use_xyz, num_max_sum_points, nsample,
                                             neighbor_type, pooling_type)
        _put(new_features, nf)
        _put(new_local_xyz, nl)
        _put(point_cnt_of_grid, cg)
        if grouped.shape[0]:
            grouped_idxs[:grouped.shape[0]] = torch.from_numpy(np.ascontiguousarray(grouped))
        return int(cum)

    def vector_pool_grad_wrapper(grad_new_features, point_cnt_of_grid, grouped_idxs, grad_support_features):
        n_rows, c_in = grad_support_features.shape
        _put(grad_support_features, O.vp_pool_grad(_n(grad_new_features), _n(point_cnt_of_grid), _n(grouped_idxs), n_rows, c_in))
        return 0

    for f in (ball_query_wrapper, voxel_query_wrapper, farthest_point_sampling_wrapper,
              stack_farthest_point_sampling_wrapper, group_points_wrapper, group_points_grad_wrapper, three_nn_wrapper,
              three_interpolate_wrapper, three_interpolate_grad_wrapper, query_stacked_local_neighbor_idxs_wrapper_stack,
              query_three_nn_by_stacked_local_idxs_wrapper_stack, vector_pool_wrapper, vector_pool_grad_wrapper):
        setattr(m, f.__name__, f)
    return m


class patched_build_package:
    """Context manager: route spsnet_amd.pointnet2_stack's extension calls to the CPU stand-in (host-logic tests)."""

    def __enter__(self):
        from spsnet_amd.pointnet2_stack import pointnet2_stack_cuda as ext
        self.ext, self.saved = ext, {}
        standin = standin_module("standin")
        for name in dir(standin):
            if name.endswith("_wrapper") or name.endswith("_wrapper_stack"):
                self.saved[name] = getattr(ext, name)
                setattr(ext, name, getattr(standin, name))
        return self

    def __exit__(self, *exc):
        for name, fn in self.saved.items():
            setattr(self.ext, name, fn)


_loaded = None


def load_reference():
    """-> (pointnet2_utils, pointnet2_modules, voxel_query_utils, voxel_pool_modules) of the REFERENCE's stack package,
    running on CPU over the stand-in (build container only)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present (Tier-2 oracle is build-container only)")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    torch.cuda.IntTensor = torch.IntTensor       # the wrappers allocate with the legacy torch.cuda.*Tensor constructors
    torch.cuda.FloatTensor = torch.FloatTensor
    pkg = importlib.import_module(_PKG)
    standin = standin_module()
    sys.modules[_PKG + ".pointnet2_stack_cuda"] = standin
    setattr(pkg, "pointnet2_stack_cuda", standin)
    _loaded = tuple(importlib.import_module(f"{_PKG}.{name}")
                    for name in ("pointnet2_utils", "pointnet2_modules", "voxel_query_utils", "voxel_pool_modules"))
    return _loaded
