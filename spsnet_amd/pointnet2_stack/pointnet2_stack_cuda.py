"""Drop-in for the reference's pybind extension `pointnet2_stack_cuda`
(pcdet/ops/pointnet2/pointnet2_stack/src/pointnet2_api.cpp:12-31): same function names, positional signatures and
return values for all fourteen functions, forwarded to libspsnet_sa.so (csrc/pointnet2_stack.hip).
Like the batch module: failed checks raise instead of exit(-1); kernels run on torch's current stream."""
import torch

from .. import _lib
from ..pointnet2_batch_cuda import _ptr, _need, _stream, _on, F32, I32, farthest_point_sampling_wrapper  # noqa: F401

_L = _lib.load()


def ball_query_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx):
    """ball_query.cpp (stack): idx (M, nsample) pre-zeroed; local indices, idx[row, 0] = -1 for an empty ball."""
    q, qc = _ptr(new_xyz, F32, "new_xyz"), _ptr(new_xyz_batch_cnt, I32, "new_xyz_batch_cnt")
    p, pc = _ptr(xyz, F32, "xyz"), _ptr(xyz_batch_cnt, I32, "xyz_batch_cnt")
    _need(new_xyz, M * 3, "new_xyz"); _need(idx, M * nsample, "idx"); _need(new_xyz_batch_cnt, B, "new_xyz_batch_cnt")
    _need(xyz_batch_cnt, B, "xyz_batch_cnt")
    with _on(xyz):
        _lib.check(_L.sps_ball_query_kernel_launcher_stack(B, M, radius, nsample, q, qc, p, pc, _ptr(idx, I32, "idx"),
                                                           _stream(xyz)), "ball_query_stack")
    return 1


def voxel_query_wrapper(M, R1, R2, R3, nsample, radius, z_range, y_range, x_range, new_xyz, xyz, new_coords,
                        point_indices, idx):
    _need(new_xyz, M * 3, "new_xyz"); _need(new_coords, M * 4, "new_coords"); _need(idx, M * nsample, "idx")
    with _on(xyz):
        _lib.check(_L.sps_voxel_query_kernel_launcher_stack(
            M, R1, R2, R3, nsample, radius, z_range, y_range, x_range, _ptr(new_xyz, F32, "new_xyz"), _ptr(xyz, F32, "xyz"),
            _ptr(new_coords, I32, "new_coords"), _ptr(point_indices, I32, "point_indices"), _ptr(idx, I32, "idx"),
            _stream(xyz)), "voxel_query_stack")
    return 1


def stack_farthest_point_sampling_wrapper(points, temp, xyz_batch_cnt, idx, num_sampled_points):
    """sampling.cpp:50-66 (stack): points (N,3), temp (N) = 1e10, idx (sum npoint) -> global indices."""
    N, B = points.shape[0], xyz_batch_cnt.shape[0]
    _need(temp, N, "temp"); _need(num_sampled_points, B, "num_sampled_points")
    with _on(points):
        _lib.check(_L.sps_stack_farthest_point_sampling_kernel_launcher(
            N, B, _ptr(points, F32, "points"), _ptr(temp, F32, "temp"), _ptr(xyz_batch_cnt, I32, "xyz_batch_cnt"),
            _ptr(idx, I32, "idx"), _ptr(num_sampled_points, I32, "num_sampled_points"), _stream(points)), "stack_fps")
    return 1


def group_points_wrapper(B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, out):
    _need(idx, M * nsample, "idx"); _need(out, M * C * nsample, "out")
    with _on(features):
        _lib.check(_L.sps_group_points_kernel_launcher_stack(
            B, M, C, nsample, _ptr(features, F32, "features"), _ptr(features_batch_cnt, I32, "features_batch_cnt"),
            _ptr(idx, I32, "idx"), _ptr(idx_batch_cnt, I32, "idx_batch_cnt"), _ptr(out, F32, "out"), _stream(features)),
            "group_points_stack")
    return 1


def group_points_grad_wrapper(B, M, C, N, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features):
    _need(grad_out, M * C * nsample, "grad_out"); _need(grad_features, N * C, "grad_features")
    with _on(grad_out):
        _lib.check(_L.sps_group_points_grad_kernel_launcher_stack(
            B, M, C, N, nsample, _ptr(grad_out, F32, "grad_out"), _ptr(idx, I32, "idx"),
            _ptr(idx_batch_cnt, I32, "idx_batch_cnt"), _ptr(features_batch_cnt, I32, "features_batch_cnt"),
            _ptr(grad_features, F32, "grad_features"), _stream(grad_out)), "group_points_grad_stack")
    return 1


def three_nn_wrapper(unknown, unknown_batch_cnt, known, known_batch_cnt, dist2, idx):
    """interpolate.cpp (stack): dist2 / idx (N,3); idx are global rows of `known`."""
    B, N, M = unknown_batch_cnt.shape[0], unknown.shape[0], known.shape[0]
    _need(dist2, N * 3, "dist2"); _need(idx, N * 3, "idx")
    with _on(unknown):
        _lib.check(_L.sps_three_nn_kernel_launcher_stack(
            B, N, M, _ptr(unknown, F32, "unknown"), _ptr(unknown_batch_cnt, I32, "unknown_batch_cnt"),
            _ptr(known, F32, "known"), _ptr(known_batch_cnt, I32, "known_batch_cnt"), _ptr(dist2, F32, "dist2"),
            _ptr(idx, I32, "idx"), _stream(unknown)), "three_nn_stack")
    return 1


def three_interpolate_wrapper(features, idx, weight, out):
    N, C = idx.shape[0], features.shape[1]
    _need(weight, N * 3, "weight"); _need(out, N * C, "out")
    with _on(features):
        _lib.check(_L.sps_three_interpolate_kernel_launcher_stack(
            N, C, _ptr(features, F32, "features"), _ptr(idx, I32, "idx"), _ptr(weight, F32, "weight"), _ptr(out, F32, "out"),
            _stream(features)), "three_interpolate_stack")
    return 1


def three_interpolate_grad_wrapper(grad_out, idx, weight, grad_features):
    N, C = grad_out.shape[0], grad_out.shape[1]
    _need(idx, N * 3, "idx"); _need(weight, N * 3, "weight")
    with _on(grad_out):
        _lib.check(_L.sps_three_interpolate_grad_kernel_launcher_stack(
            N, C, _ptr(grad_out, F32, "grad_out"), _ptr(idx, I32, "idx"), _ptr(weight, F32, "weight"),
            _ptr(grad_features, F32, "grad_features"), _stream(grad_out)), "three_interpolate_grad_stack")
    return 1


def query_stacked_local_neighbor_idxs_wrapper_stack(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, stack_neighbor_idxs,
                                                    start_len, cumsum, avg_length_of_neighbor_idxs, max_neighbour_distance,
                                                    nsample, neighbor_type):
    """vector_pool.cpp: per-centre local neighbour lists, packed; cumsum (1,) int32 zeroed by the caller."""
    B, M = xyz_batch_cnt.shape[0], new_xyz.shape[0]
    with _on(support_xyz):
        _lib.check(_L.sps_query_stacked_local_neighbor_idxs_kernel_launcher_stack(
            _ptr(support_xyz, F32, "support_xyz"), _ptr(xyz_batch_cnt, I32, "xyz_batch_cnt"), _ptr(new_xyz, F32, "new_xyz"),
            _ptr(new_xyz_batch_cnt, I32, "new_xyz_batch_cnt"), _ptr(stack_neighbor_idxs, I32, "stack_neighbor_idxs"),
            _ptr(start_len, I32, "start_len"), _ptr(cumsum, I32, "cumsum"), int(avg_length_of_neighbor_idxs),
            float(max_neighbour_distance), B, M, int(nsample), int(neighbor_type), _stream(support_xyz)),
            "query_stacked_local_neighbor_idxs")
    return 0


def query_three_nn_by_stacked_local_idxs_wrapper_stack(support_xyz, new_xyz, new_xyz_grid_centers, new_xyz_grid_idxs,
                                                       new_xyz_grid_dist2, stack_neighbor_idxs, start_len, M, num_total_grids):
    with _on(support_xyz):
        _lib.check(_L.sps_query_three_nn_by_stacked_local_idxs_kernel_launcher_stack(
            _ptr(support_xyz, F32, "support_xyz"), _ptr(new_xyz, F32, "new_xyz"),
            _ptr(new_xyz_grid_centers, F32, "new_xyz_grid_centers"), _ptr(new_xyz_grid_idxs, I32, "new_xyz_grid_idxs"),
            _ptr(new_xyz_grid_dist2, F32, "new_xyz_grid_dist2"), _ptr(stack_neighbor_idxs, I32, "stack_neighbor_idxs"),
            _ptr(start_len, I32, "start_len"), int(M), int(num_total_grids), _stream(support_xyz)),
            "query_three_nn_by_stacked_local_idxs")
    return 0


def vector_pool_wrapper(support_xyz, xyz_batch_cnt, support_features, new_xyz, new_xyz_batch_cnt, new_features, new_local_xyz,
                        point_cnt_of_grid, grouped_idxs, num_grid_x, num_grid_y, num_grid_z, max_neighbour_distance, use_xyz,
                        num_max_sum_points, nsample, neighbor_type, pooling_type):
    """vector_pool.cpp:vector_pool_wrapper_stack -> the number of (point, centre, grid) triples found (may exceed
    num_max_sum_points: the caller then retries with a larger buffer).  Reads the device counter: synchronises, like the
    reference's cudaMemcpy."""
    B, N, M = xyz_batch_cnt.shape[0], support_xyz.shape[0], new_xyz.shape[0]
    c_in, c_out = support_features.shape[1], new_features.shape[1]
    total = num_grid_x * num_grid_y * num_grid_z
    counter = torch.zeros((1,), dtype=I32, device=support_xyz.device)
    with _on(support_xyz):
        _lib.check(_L.sps_vector_pool_kernel_launcher_stack(
            _ptr(support_xyz, F32, "support_xyz"), _ptr(support_features, F32, "support_features"),
            _ptr(xyz_batch_cnt, I32, "xyz_batch_cnt"), _ptr(new_xyz, F32, "new_xyz"), _ptr(new_features, F32, "new_features"),
            _ptr(new_local_xyz, F32, "new_local_xyz"), _ptr(new_xyz_batch_cnt, I32, "new_xyz_batch_cnt"),
            _ptr(point_cnt_of_grid, I32, "point_cnt_of_grid"), _ptr(grouped_idxs, I32, "grouped_idxs"), int(num_grid_x),
            int(num_grid_y), int(num_grid_z), float(max_neighbour_distance), B, N, M, c_in, c_out, total, int(use_xyz),
            int(num_max_sum_points), int(nsample), int(neighbor_type), int(pooling_type), counter.data_ptr(),
            _stream(support_xyz)), "vector_pool")
    return int(counter.item())


def vector_pool_grad_wrapper(grad_new_features, point_cnt_of_grid, grouped_idxs, grad_support_features):
    M, c_out = grad_new_features.shape
    N, c_in = grad_support_features.shape
    total = point_cnt_of_grid.shape[1]
    rows = grouped_idxs.shape[0]
    with _on(grad_new_features):
        _lib.check(_L.sps_vector_pool_grad_kernel_launcher_stack(
            _ptr(grad_new_features, F32, "grad_new_features"), _ptr(point_cnt_of_grid, I32, "point_cnt_of_grid"),
            _ptr(grouped_idxs, I32, "grouped_idxs"), _ptr(grad_support_features, F32, "grad_support_features"), N, M, c_out,
            c_in, total, rows, _stream(grad_new_features)), "vector_pool_grad")
    return 0
