"""numpy/ctypes front-end of the C oracle (oracle/sa_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() as the checker -- never by the product package.

Each function mirrors one reference kernel (file:line in sa_oracle.c) and, like the
reference's pybind functions (src/pointnet2_api.cpp:10-26), writes into caller-style
buffers that it allocates the way pointnet2_utils.py does (temp = 1e10, idx = 0, grads = 0).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libsa_oracle.so")
_lib = None

_F = ctypes.POINTER(ctypes.c_float)
_I = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    """Compile sa_oracle.c with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "sa_oracle.c")
    stale = (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "clean", "all"], check=True,
                       stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_opt_n_threads.restype = ctypes.c_int
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_F)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_I)


def set_threads(n):
    """OpenMP threads used by the following oracle calls (bench.py: all-cores and single-thread CPU rows)."""
    lib().orc_set_threads(int(n))


def opt_n_threads(n):
    return int(lib().orc_opt_n_threads(int(n)))


def fps(xyz, npoint, temp=None, return_temp=False):
    """xyz (B,N,3) f32 -> idx (B,npoint) i32 [sampling_gpu.cu:93-208]."""
    xyz, px = _f(xyz)
    B, N, _ = xyz.shape
    temp = np.full((B, N), 1e10, np.float32) if temp is None else np.array(temp, np.float32)
    idx = np.zeros((B, npoint), np.int32)
    lib().orc_fps(B, N, npoint, px, temp.ctypes.data_as(_F), idx.ctypes.data_as(_I))
    return (idx, temp) if return_temp else idx


def fps_with_dist(dist, npoint):
    """dist (B,N,N) f32 -> idx (B,npoint) i32 [sampling_gpu.cu:256-371]."""
    dist, pd = _f(dist)
    B, N, _ = dist.shape
    temp = np.full((B, N), 1e10, np.float32)
    idx = np.zeros((B, npoint), np.int32)
    lib().orc_fps_with_dist(B, N, npoint, pd, temp.ctypes.data_as(_F), idx.ctypes.data_as(_I))
    return idx


def gather_points(points, idx):
    """points (B,C,N), idx (B,M) -> (B,C,M) [sampling_gpu.cu:8-24]."""
    points, pp = _f(points)
    idx, pi = _i(idx)
    B, C, N = points.shape
    M = idx.shape[1]
    out = np.empty((B, C, M), np.float32)
    lib().orc_gather_points(B, C, N, M, pp, pi, out.ctypes.data_as(_F))
    return out


def gather_points_grad(grad_out, idx, N):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    B, C, M = grad_out.shape
    gp = np.zeros((B, C, N), np.float32)
    lib().orc_gather_points_grad(B, C, N, M, pg, pi, gp.ctypes.data_as(_F))
    return gp


def ball_query(radius, nsample, xyz, new_xyz):
    """-> idx (B,M,nsample) i32 [ball_query_gpu.cu:9-45]; argument order of BallQuery.forward."""
    xyz, px = _f(xyz)
    new_xyz, pn = _f(new_xyz)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = np.zeros((B, M, nsample), np.int32)
    lib().orc_ball_query(B, N, M, ctypes.c_float(radius), nsample, pn, px, idx.ctypes.data_as(_I))
    return idx


def ball_query_dilated(max_radius, min_radius, nsample, xyz, new_xyz):
    xyz, px = _f(xyz)
    new_xyz, pn = _f(new_xyz)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = np.zeros((B, M, nsample), np.int32)
    lib().orc_ball_query_dilated(B, N, M, ctypes.c_float(max_radius), ctypes.c_float(min_radius),
                                 nsample, pn, px, idx.ctypes.data_as(_I))
    return idx


def group_points(points, idx):
    """points (B,C,N), idx (B,M,ns) -> (B,C,M,ns) [group_points_gpu.cu:53-71]."""
    points, pp = _f(points)
    idx, pi = _i(idx)
    B, C, N = points.shape
    _, M, ns = idx.shape
    out = np.empty((B, C, M, ns), np.float32)
    lib().orc_group_points(B, C, N, M, ns, pp, pi, out.ctypes.data_as(_F))
    return out


def group_points_grad(grad_out, idx, N):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    B, C, M, ns = grad_out.shape
    gp = np.zeros((B, C, N), np.float32)
    lib().orc_group_points_grad(B, C, N, M, ns, pg, pi, gp.ctypes.data_as(_F))
    return gp


def three_nn(unknown, known):
    """-> (dist2 (B,n,3) f32, idx (B,n,3) i32) [interpolate_gpu.cu:16-58] (dist2, not sqrt)."""
    unknown, pu = _f(unknown)
    known, pk = _f(known)
    B, n, _ = unknown.shape
    m = known.shape[1]
    d2 = np.empty((B, n, 3), np.float32)
    idx = np.empty((B, n, 3), np.int32)
    lib().orc_three_nn(B, n, m, pu, pk, d2.ctypes.data_as(_F), idx.ctypes.data_as(_I))
    return d2, idx


def three_interpolate(points, idx, weight):
    points, pp = _f(points)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    B, C, m = points.shape
    n = idx.shape[1]
    out = np.empty((B, C, n), np.float32)
    lib().orc_three_interpolate(B, C, m, n, pp, pi, pw, out.ctypes.data_as(_F))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    B, C, n = grad_out.shape
    gp = np.zeros((B, C, m), np.float32)
    lib().orc_three_interpolate_grad(B, C, n, m, pg, pi, pw, gp.ctypes.data_as(_F))
    return gp


def score_ctr(cls):
    """cls (B,N,C) -> sigmoid(max_c) (B,N) [pointnet2_modules.py:288-289]."""
    cls, pc = _f(cls)
    B, N, C = cls.shape
    s = np.empty((B, N), np.float32)
    lib().orc_score_ctr(B, N, C, pc, s.ctypes.data_as(_F))
    return s


def score_stability(cls, stds):
    """-> sigmoid(max_c cls) * (1 - sigmoid(stds/8-3)) (B,N) [pointnet2_modules.py:296-301]."""
    cls, pc = _f(cls)
    stds, ps = _f(stds)
    B, N, C = cls.shape
    s = np.empty((B, N), np.float32)
    lib().orc_score_stability(B, N, C, pc, ps, s.ctypes.data_as(_F))
    return s


def topk_desc(score, k):
    """(B,N) -> (B,k) i32: score descending, index ascending on ties."""
    score, ps = _f(score)
    B, N = score.shape
    idx = np.empty((B, k), np.int32)
    lib().orc_topk_desc(B, N, k, ps, idx.ctypes.data_as(_I))
    return idx


# ---- stacked (ragged-batch) variants: pcdet/ops/pointnet2/pointnet2_stack/src (parity unpinned, see sa_oracle.c) ----
def stack_ball_query(radius, nsample, xyz, xyz_cnt, new_xyz, new_cnt):
    """-> raw idx (M, nsample) as the kernel leaves it on a zeroed buffer: local indices, [row, 0] = -1 if empty."""
    xyz, px = _f(xyz); new_xyz, pq = _f(new_xyz)
    xyz_cnt, pxc = _i(xyz_cnt); new_cnt, pqc = _i(new_cnt)
    M = new_xyz.shape[0]
    idx = np.zeros((M, nsample), np.int32)
    lib().orc_stack_ball_query(len(xyz_cnt), M, ctypes.c_float(radius), nsample, pq, pqc, px, pxc, idx.ctypes.data_as(_I))
    return idx


def stack_voxel_query(max_range, radius, nsample, xyz, new_xyz, new_coords, point_indices):
    xyz, px = _f(xyz); new_xyz, pq = _f(new_xyz)
    new_coords, pc = _i(new_coords); point_indices, pp = _i(point_indices)
    M = new_xyz.shape[0]
    B, Z, Y, X = point_indices.shape
    idx = np.zeros((M, nsample), np.int32)
    lib().orc_stack_voxel_query(M, Z, Y, X, nsample, ctypes.c_float(radius), int(max_range[0]), int(max_range[1]),
                                int(max_range[2]), pq, px, pc, pp, idx.ctypes.data_as(_I))
    return idx


def stack_fps(xyz, xyz_cnt, npoint):
    xyz, px = _f(xyz); xyz_cnt, pc = _i(xyz_cnt); npoint, pn = _i(npoint)
    temp = np.full((xyz.shape[0],), 1e10, np.float32)
    out = np.zeros((int(npoint.sum()),), np.int32)
    lib().orc_stack_fps(len(xyz_cnt), px, temp.ctypes.data_as(_F), pc, out.ctypes.data_as(_I), pn)
    return out


def stack_group_points(features, features_cnt, idx, idx_cnt):
    features, pf = _f(features); features_cnt, pfc = _i(features_cnt); idx, pi = _i(idx); idx_cnt, pic = _i(idx_cnt)
    M, ns = idx.shape
    C = features.shape[1]
    out = np.empty((M, C, ns), np.float32)
    lib().orc_stack_group_points(len(idx_cnt), M, C, ns, pf, pfc, pi, pic, out.ctypes.data_as(_F))
    return out


def stack_group_points_grad(grad_out, idx, idx_cnt, features_cnt, N):
    grad_out, pg = _f(grad_out); idx, pi = _i(idx); idx_cnt, pic = _i(idx_cnt); features_cnt, pfc = _i(features_cnt)
    M, C, ns = grad_out.shape
    out = np.zeros((N, C), np.float32)
    lib().orc_stack_group_points_grad(len(idx_cnt), M, C, ns, pg, pi, pic, pfc, out.ctypes.data_as(_F))
    return out


def stack_three_nn(unknown, unknown_cnt, known, known_cnt):
    unknown, pu = _f(unknown); known, pk = _f(known); unknown_cnt, puc = _i(unknown_cnt); known_cnt, pkc = _i(known_cnt)
    N = unknown.shape[0]
    d2 = np.empty((N, 3), np.float32)
    idx = np.empty((N, 3), np.int32)
    lib().orc_stack_three_nn(len(unknown_cnt), N, pu, puc, pk, pkc, d2.ctypes.data_as(_F), idx.ctypes.data_as(_I))
    return d2, idx


def stack_three_interpolate(features, idx, weight):
    features, pf = _f(features); idx, pi = _i(idx); weight, pw = _f(weight)
    N, C = idx.shape[0], features.shape[1]
    out = np.empty((N, C), np.float32)
    lib().orc_stack_three_interpolate(N, C, pf, pi, pw, out.ctypes.data_as(_F))
    return out


def stack_three_interpolate_grad(grad_out, idx, weight, M):
    grad_out, pg = _f(grad_out); idx, pi = _i(idx); weight, pw = _f(weight)
    N, C = grad_out.shape
    out = np.zeros((M, C), np.float32)
    lib().orc_stack_three_interpolate_grad(N, C, pg, pi, pw, out.ctypes.data_as(_F))
    return out


def vp_local_neighbors(support_xyz, xyz_cnt, new_xyz, new_cnt, dist, nsample, neighbor_type):
    """-> list of int32 arrays: per centre, the global rows of its local neighbours in index order."""
    support_xyz, ps = _f(support_xyz); new_xyz, pq = _f(new_xyz); xyz_cnt, pc = _i(xyz_cnt); new_cnt, pqc = _i(new_cnt)
    M = new_xyz.shape[0]
    cap = 1000
    lists = np.zeros((M, cap), np.int32)
    lens = np.zeros((M,), np.int32)
    lib().orc_vp_local_neighbors(len(xyz_cnt), M, ps, pc, pq, pqc, ctypes.c_float(dist), int(nsample), int(neighbor_type), cap,
                                 lists.ctypes.data_as(_I), lens.ctypes.data_as(_I))
    return lists, lens


def vp_three_nn_local(support_xyz, grid_centers, lists, lens):
    support_xyz, ps = _f(support_xyz); grid_centers, pg = _f(grid_centers); lists, pl = _i(lists); lens, pn = _i(lens)
    M, G, _ = grid_centers.shape
    idx = np.empty((M, G, 3), np.int32)
    d2 = np.empty((M, G, 3), np.float32)
    lib().orc_vp_three_nn_local(M, G, lists.shape[1], ps, pg, pl, pn, idx.ctypes.data_as(_I), d2.ctypes.data_as(_F))
    return d2, idx


def vp_pool(support_xyz, xyz_cnt, support_features, new_xyz, new_cnt, grid, dist, c_each, use_xyz, max_sum, nsample,
            neighbor_type, pooling_type):
    support_xyz, ps = _f(support_xyz); support_features, pf = _f(support_features); new_xyz, pq = _f(new_xyz)
    xyz_cnt, pc = _i(xyz_cnt); new_cnt, pqc = _i(new_cnt)
    M, total = new_xyz.shape[0], grid[0] * grid[1] * grid[2]
    c_in, c_out = support_features.shape[1], c_each * total
    nf = np.zeros((M, c_out), np.float32); nl = np.zeros((M, 3 * total), np.float32)
    cg = np.zeros((M, total), np.int32); grouped = np.zeros((max(max_sum, 1), 3), np.int32)
    lib().orc_vp_pool.restype = ctypes.c_int
    cum = lib().orc_vp_pool(len(xyz_cnt), M, ps, pf, pc, pq, pqc, grid[0], grid[1], grid[2], ctypes.c_float(dist), c_in, c_out,
                            int(use_xyz), int(max_sum), int(nsample), int(neighbor_type), int(pooling_type),
                            nf.ctypes.data_as(_F), nl.ctypes.data_as(_F), cg.ctypes.data_as(_I), grouped.ctypes.data_as(_I))
    return cum, nf, nl, cg, grouped[:min(cum, max_sum)]


def vp_pool_grad(grad_new_features, cnt_of_grid, grouped, N, c_in):
    grad_new_features, pg = _f(grad_new_features); cnt_of_grid, pc = _i(cnt_of_grid); grouped, pr = _i(grouped)
    out = np.zeros((N, c_in), np.float32)
    lib().orc_vp_pool_grad(grouped.shape[0], c_in, grad_new_features.shape[1], cnt_of_grid.shape[1], pg, pc, pr,
                           out.ctypes.data_as(_F))
    return out
