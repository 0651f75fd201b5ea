"""numpy <-> HIP helpers for the GPU parity tests: every call goes through the C ABI
(spsnet_amd.pointnet2_batch_cuda -> libspsnet_sa.so), mirroring how pointnet2_utils.py allocates."""
import numpy as np
import torch

DEV = "cuda:0"


def t(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return x if dtype is None else x.to(dtype)


def n(x):
    return x.detach().cpu().numpy()


def fps(ext, xyz, m, temp=None):
    B, N, _ = xyz.shape
    tx = t(xyz)
    tt = torch.full((B, N), 1e10, dtype=torch.float32, device=DEV) if temp is None else t(temp)
    idx = torch.zeros((B, m), dtype=torch.int32, device=DEV)
    assert ext.farthest_point_sampling_wrapper(B, N, m, tx, tt, idx) == 1
    return n(idx), n(tt)


def fps_with_dist(ext, dist, m):
    B, N, _ = dist.shape
    tt = torch.full((B, N), 1e10, dtype=torch.float32, device=DEV)
    idx = torch.zeros((B, m), dtype=torch.int32, device=DEV)
    assert ext.furthest_point_sampling_with_dist_wrapper(B, N, m, t(dist), tt, idx) == 2
    return n(idx)


def ball_query(ext, radius, nsample, xyz, new_xyz, dilated_min=None):
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = torch.zeros((B, M, nsample), dtype=torch.int32, device=DEV)
    if dilated_min is None:
        assert ext.ball_query_wrapper(B, N, M, radius, nsample, t(new_xyz), t(xyz), idx) == 1
    else:
        assert ext.ball_query_dilated_wrapper(B, N, M, radius, dilated_min, nsample, t(new_xyz), t(xyz), idx) == 1
    return n(idx)


def group(ext, pts, idx):
    B, C, N = pts.shape
    _, M, ns = idx.shape
    out = torch.empty((B, C, M, ns), dtype=torch.float32, device=DEV)
    assert ext.group_points_wrapper(B, C, N, M, ns, t(pts), t(idx), out) == 1
    return n(out)


def gather(ext, pts, idx):
    B, C, N = pts.shape
    M = idx.shape[1]
    out = torch.empty((B, C, M), dtype=torch.float32, device=DEV)
    assert ext.gather_points_wrapper(B, C, N, M, t(pts), t(idx), out) == 1
    return n(out)


def compare_matched_rows(got_idx, want_idx, triples):
    """Two score-sampled layers whose picks agree up to near-ties: match rows by sampled index and compare each
    (got, want, rel_tol, axis) tensor on the intersection (`axis` = the axis that runs over the M picks; tol 0 = exact).
    -> mean fraction of picks shared."""
    B = got_idx.shape[0]
    shared = []
    for b in range(B):
        common, gpos, wpos = np.intersect1d(got_idx[b], want_idx[b], return_indices=True)
        shared.append(len(common) / want_idx.shape[1])
        for got, want, tol, axis in triples:
            g = np.take(got[b], gpos, axis=axis - 1)
            w = np.take(want[b], wpos, axis=axis - 1)
            if tol == 0.0:
                np.testing.assert_array_equal(g, w)
            else:
                scale = max(1.0, float(np.abs(want[b]).max()))
                err = float(np.abs(g - w).max()) if g.size else 0.0
                assert err <= tol * scale, f"scene {b}: matched rows differ by {err} (scale {scale})"
    return float(np.mean(shared))


# substrings of kernel names that come out of a vendor library (rocBLAS / hipBLASLt / Tensile GEMMs, MIOpen convolutions,
# BatchNorm and layout kernels): none may run inside a step that claims hand-written kernels
LIBRARY_KERNEL_MARKS = ("Cijk_", "igemm_", "MIOpen", "miopen", "batched_transpose", "SubTensorOp", "gemm", "Gemm", "naive_conv",
                        "im2col", "Im2Col", "rocblas", "hipblas")


def kernel_names(fn):
    """{kernel name: launches} of ONE call of fn() from torch.profiler's device activity records (roctracer): what the GPU
    actually ran, whatever Python path led there.  fn() is called once before the trace (one-time set-up: weight packing,
    stream probes, MIOpen's find pass would otherwise pollute it)."""
    from torch.profiler import ProfilerActivity, profile
    fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    names = {}
    for ev in prof.events():
        if str(getattr(ev, "device_type", "")).endswith("CUDA") and ev.name:
            names[ev.name] = names.get(ev.name, 0) + 1
    return names


def library_kernels(names):
    return sorted(k for k in names if any(mark in k for mark in LIBRARY_KERNEL_MARKS))
