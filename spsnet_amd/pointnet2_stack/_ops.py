"""Functional layer of the ragged-batch ("stack") ops: plain functions over the fourteen entry points of
`pointnet2_stack_cuda` (csrc/pointnet2_stack.hip through the C ABI).

Conventions used by everything in this package:
  * a ragged batch is a row-concatenated tensor plus its `counts` (batch_size,) int32 -- rows of scene b are
    [offset_b, offset_b + counts[b]);
  * functions allocate their outputs on the inputs' device, never on "the current CUDA device";
  * neighbour indices are int32; an index row whose first entry came back as -1 marks "no neighbour" and is returned
    as a zero row together with a boolean mask (what the reference's wrappers hand to their callers,
    pointnet2_stack/pointnet2_utils.py:38-40, voxel_query_utils.py:38-40).

The autograd classes in pointnet2_utils.py / voxel_query_utils.py are thin adapters around these functions.
"""
import torch

from . import pointnet2_stack_cuda as _ext

I32, F32 = torch.int32, torch.float32


def rows_of(counts) -> int:
    return int(counts.sum().item()) if isinstance(counts, torch.Tensor) else int(sum(counts))


def require_rows(t, counts, what):
    """A stacked tensor must hold exactly sum(counts) rows."""
    want = rows_of(counts)
    if t.shape[0] != want:
        raise AssertionError(f"{what}: {t.shape[0]} rows, but the batch counts {counts.tolist()} add up to {want}")


def require_contiguous(**named):
    for name, t in named.items():
        if t is not None and not t.is_contiguous():
            raise AssertionError(f"{name} must be contiguous")


def _clear_empty_rows(idx):
    """-1 in column 0 marks a row without neighbours -> (zeroed rows, mask)."""
    empty = idx[:, 0] == -1
    idx.masked_fill_(empty.unsqueeze(1), 0)
    return idx, empty


# ---------------------------------------------------------------------------------------------- neighbour queries
def ball_query(radius, nsample, xyz, xyz_cnt, centres, centre_cnt):
    """First `nsample` points (scene-local index order) within `radius` of each centre -> (idx (M, nsample), empty (M,))."""
    require_contiguous(xyz=xyz, xyz_batch_cnt=xyz_cnt, new_xyz=centres, new_xyz_batch_cnt=centre_cnt)
    M = centres.shape[0]
    idx = torch.zeros((M, nsample), dtype=I32, device=xyz.device)
    _ext.ball_query_wrapper(xyz_cnt.shape[0], M, radius, nsample, centres, centre_cnt, xyz, xyz_cnt, idx)
    return _clear_empty_rows(idx)


def voxel_query(max_range, radius, nsample, xyz, centres, centre_coords, voxel_to_point):
    """Neighbours looked up through a dense voxel -> point table (B, Z, Y, X) around each centre's voxel
    [b, z, y, x] within `max_range` = (z, y, x) voxels -> (idx (M, nsample) GLOBAL rows, empty (M,))."""
    require_contiguous(new_xyz=centres, xyz=xyz, new_coords=centre_coords, point_indices=voxel_to_point)
    M = centre_coords.shape[0]
    _, Z, Y, X = voxel_to_point.shape
    rz, ry, rx = max_range
    idx = torch.zeros((M, nsample), dtype=I32, device=xyz.device)
    _ext.voxel_query_wrapper(M, Z, Y, X, nsample, radius, rz, ry, rx, centres, xyz, centre_coords, voxel_to_point, idx)
    return _clear_empty_rows(idx)


# ---------------------------------------------------------------------------------------------- grouping
def group(features, feature_cnt, idx, idx_cnt):
    """features (N, C), idx (M, nsample) scene-local -> (M, C, nsample)."""
    require_contiguous(features=features, features_batch_cnt=feature_cnt, idx=idx, idx_batch_cnt=idx_cnt)
    require_rows(features, feature_cnt, "features")
    require_rows(idx, idx_cnt, "idx")
    M, nsample = idx.shape
    C = features.shape[1]
    out = torch.empty((M, C, nsample), dtype=F32, device=features.device)
    _ext.group_points_wrapper(idx_cnt.shape[0], M, C, nsample, features, feature_cnt, idx, idx_cnt, out)
    return out


def group_grad(grad_out, idx, idx_cnt, feature_cnt, n_rows):
    """Scatter-add of d(group): grad_out (M, C, nsample) -> (n_rows, C)."""
    M, C, nsample = grad_out.shape
    grad = torch.zeros((n_rows, C), dtype=F32, device=grad_out.device)
    _ext.group_points_grad_wrapper(idx_cnt.shape[0], M, C, n_rows, nsample, grad_out.contiguous(), idx, idx_cnt,
                                   feature_cnt, grad)
    return grad


# ---------------------------------------------------------------------------------------------- sampling
def fps_batch(xyz, npoint):
    """Equal-size scenes (B, N, 3) -> (B, npoint) int32 (the batch kernel the stack extension re-exports)."""
    require_contiguous(xyz=xyz)
    B, N, _ = xyz.shape
    picks = torch.empty((B, npoint), dtype=I32, device=xyz.device)
    running = torch.full((B, N), 1e10, dtype=F32, device=xyz.device)
    _ext.farthest_point_sampling_wrapper(B, N, npoint, xyz, running, picks)
    return picks


def fps_stack(xyz, xyz_cnt, npoint):
    """Ragged scenes (N, 3): npoint = int, list or (B,) tensor of picks per scene -> (sum npoint,) GLOBAL rows."""
    if not (xyz.is_contiguous() and xyz.dim() == 2 and xyz.shape[1] == 3):
        raise AssertionError("xyz must be a contiguous (N, 3) tensor")
    scenes = len(xyz_cnt)
    if not isinstance(npoint, torch.Tensor):
        per_scene = list(npoint) if isinstance(npoint, (list, tuple)) else [npoint] * scenes
        npoint = torch.tensor(per_scene, device=xyz.device).int()
    npoint = npoint.contiguous()
    running = torch.full((xyz.shape[0],), 1e10, dtype=F32, device=xyz.device)
    picks = torch.empty((int(npoint.sum().item()),), dtype=I32, device=xyz.device)
    _ext.stack_farthest_point_sampling_wrapper(xyz, running, xyz_cnt, picks, npoint)
    return picks


# ---------------------------------------------------------------------------------------------- interpolation
def three_nn(unknown, unknown_cnt, known, known_cnt):
    """-> (l2 distances (N, 3), GLOBAL rows of `known` (N, 3)) of the three nearest known points of each unknown one."""
    for name, t in (("unknown", unknown), ("known", known)):
        if t.dim() != 2 or t.shape[1] != 3:
            raise AssertionError(f"{name} must be (rows, 3)")
    if len(unknown_cnt) != len(known_cnt):
        raise AssertionError("unknown and known must describe the same number of scenes")
    sq = torch.zeros(unknown.shape, dtype=F32, device=unknown.device)
    idx = torch.zeros(unknown.shape, dtype=I32, device=unknown.device)
    _ext.three_nn_wrapper(unknown.contiguous(), unknown_cnt.contiguous(), known.contiguous(), known_cnt.contiguous(), sq, idx)
    return torch.sqrt(sq), idx


def interpolate(features, idx, weight):
    """features (M, C), idx / weight (N, 3) -> (N, C) = sum_k weight[:, k] * features[idx[:, k]]."""
    if not (idx.shape == weight.shape and idx.dim() == 2 and idx.shape[1] == 3):
        raise AssertionError("idx and weight must both be (rows, 3)")
    out = torch.zeros((idx.shape[0], features.shape[1]), dtype=features.dtype, device=features.device)
    _ext.three_interpolate_wrapper(features.contiguous(), idx.contiguous(), weight.contiguous(), out)
    return out


def interpolate_grad(grad_out, idx, weight, m_rows):
    grad = torch.zeros((m_rows, grad_out.shape[1]), dtype=grad_out.dtype, device=grad_out.device)
    _ext.three_interpolate_grad_wrapper(grad_out.contiguous(), idx.contiguous(), weight.contiguous(), grad)
    return grad


# ---------------------------------------------------------------------------------------------- vector pool (PV-RCNN++)
def run_with_growing_buffer(attempt, rows, per_row):
    """The extension's variable-length outputs use an overflow-and-retry protocol: the caller offers a buffer of
    per_row * rows entries, the kernel reports how many it needed, and the call is repeated with the reported average if it
    did not fit (reference :327-343, :401-418).  attempt(per_row) -> (needed, payload), offering per_row * rows entries.
    -> (ceil(needed / rows), needed, payload of the attempt that fitted)"""
    while True:
        offered = per_row * rows
        needed, payload = attempt(per_row)
        per_row = -(-needed // rows) if rows > 0 else 0
        if needed <= offered:
            return per_row, needed, payload


def local_three_nn(support_xyz, xyz_cnt, centres, grid_centres, centre_cnt, reach, nsample, neighbor_type, per_row, grids):
    """Three nearest support points of every local grid centre (M, grids, 3), searched only among the support points
    within `reach` of the grid's owner centre (cube, or ball when neighbor_type == 1; at most nsample of them when
    nsample > 0).  -> (l2 dist (M, grids, 3), GLOBAL rows or -1 (M, grids, 3), neighbours per centre actually needed)"""
    M = centres.shape[0]
    support_xyz, xyz_cnt = support_xyz.contiguous(), xyz_cnt.contiguous()
    centres, centre_cnt, grid_centres = centres.contiguous(), centre_cnt.contiguous(), grid_centres.contiguous()
    dev = support_xyz.device

    def attempt(per_centre):
        lists = torch.zeros((per_centre * M,), dtype=I32, device=dev)
        spans = torch.zeros((M, 2), dtype=I32, device=dev)       # (start, length) of each centre's list
        total = torch.zeros((1,), dtype=I32, device=dev)
        _ext.query_stacked_local_neighbor_idxs_wrapper_stack(support_xyz, xyz_cnt, centres, centre_cnt, lists, spans, total,
                                                             per_centre, reach, nsample, neighbor_type)
        return int(total[0].item()), (lists, spans)

    per_row, needed, (lists, spans) = run_with_growing_buffer(attempt, M, per_row)
    sq = torch.zeros(grid_centres.shape, dtype=F32, device=dev)
    idx = torch.full(grid_centres.shape, -1, dtype=I32, device=dev)
    _ext.query_three_nn_by_stacked_local_idxs_wrapper_stack(support_xyz, centres, grid_centres, idx, sq,
                                                            lists[:needed].contiguous(), spans, M, grids)
    return torch.sqrt(sq), idx, per_row


def vector_pool(support_xyz, xyz_cnt, support_features, centres, centre_cnt, grid, reach, c_each, use_xyz, per_cell,
                nsample, neighbor_type, pooling_type):
    """Sum (pooling_type 0) or first point (1) of the support features falling into each cell of the local `grid` =
    (gx, gy, gz) around every centre; channel group k of a point goes to output channel group k % c_each.
    -> (sums (M, G * c_each), xyz sums (M, 3 G), points per cell (M, G), (row, cell, point) triples (T, 3), ceil(T / M))"""
    require_contiguous(support_xyz=support_xyz, support_features=support_features, xyz_batch_cnt=xyz_cnt, new_xyz=centres,
                       new_xyz_batch_cnt=centre_cnt)
    gx, gy, gz = grid
    G = gx * gy * gz
    M = centres.shape[0]
    c_in = support_features.shape[1]
    if c_in % c_each:
        raise AssertionError(f"the input channels ({c_in}) should be an integral multiple of num_c_out_each_grid({c_each})")
    dev = support_features.device

    def attempt(mean_per_centre):
        capacity = mean_per_centre * M
        sums = torch.zeros((M, G * c_each), dtype=support_features.dtype, device=dev)
        xyz_sums = torch.zeros((M, 3 * G), dtype=support_features.dtype, device=dev)
        per_cell_cnt = torch.zeros((M, G), dtype=I32, device=dev)
        triples = torch.zeros((capacity, 3), dtype=I32, device=dev)
        needed = _ext.vector_pool_wrapper(support_xyz, xyz_cnt, support_features, centres, centre_cnt, sums, xyz_sums,
                                          per_cell_cnt, triples, gx, gy, gz, reach, use_xyz, capacity, nsample,
                                          neighbor_type, pooling_type)
        return int(needed), (sums, xyz_sums, per_cell_cnt, triples)

    per_cell, needed, (sums, xyz_sums, per_cell_cnt, triples) = run_with_growing_buffer(attempt, M, per_cell)
    return sums, xyz_sums, per_cell_cnt, triples[:needed], per_cell


def vector_pool_grad(grad_pooled, per_cell_cnt, triples, n_rows, c_in):
    grad = torch.zeros((n_rows, c_in), dtype=grad_pooled.dtype, device=grad_pooled.device)
    if triples.shape[0] > 0:
        _ext.vector_pool_grad_wrapper(grad_pooled.contiguous(), per_cell_cnt, triples.contiguous(), grad)
    return grad
