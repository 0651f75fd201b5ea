"""Drop-in for the reference's pybind extension `pointnet2_batch_cuda`.

Same 11 function names, positional signatures and return values as
pcdet/ops/pointnet2/pointnet2_batch/src/pointnet2_api.cpp:10-26 (wrappers in
src/{sampling,ball_query,group_points,interpolate}.cpp), so the reference's
`from . import pointnet2_batch_cuda as pointnet2` (pointnet2_utils.py:7) can bind this
module unchanged.  Each function hands raw device pointers and torch's
current HIP stream to libspsnet_sa.so through its C ABI (include/spsnet_sa.h).

Differences from the reference, both deliberate:
  * a failed check or launch raises (TypeError / ValueError / SpsError) instead of exit(-1);
  * kernels run on torch's *current* stream rather than the legacy default stream.
"""
import os
import threading

import torch

from . import _lib

_L = _lib.load()


def _ptr(t, dtype, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise ValueError(f"{name} must be a CUDA/HIP tensor")  # ball_query.cpp:17-22 CHECK_CUDA
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")  # ball_query.cpp:24-29 CHECK_CONTIGUOUS
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")  # .data<float>() / .data<int>()
    return t.data_ptr()


def _need(t, numel, name):
    if t.numel() < numel:
        raise ValueError(f"{name} has {t.numel()} elements, the given sizes need {numel}")


class _ScopeState(threading.local):
    """(device index, stream handle) while a launch_scope is open -- per THREAD: autograd's per-device backward threads and
    nn.DataParallel replicas run beside the thread that opened a scope and must not inherit its stream or device."""
    scope = None


_TLS = _ScopeState()


def _stream(t):
    sc = _TLS.scope
    if sc is not None and sc[0] == t.device.index:     # a tensor of another device never takes the scoped stream
        return sc[1]
    return _lib.raw_stream(t.device)


def _bump_versions(*tensors):
    """Tell torch that a kernel wrote these tensors through their raw pointers.  fused._version_key keys every folded /
    packed inference cache on (data_ptr, _version) of the BatchNorm running statistics: without the bump an eval() forward
    after train-mode forwards with frozen weights (BN recalibration, swa_utils.update_bn, lr = 0) reuses stale folds."""
    for t in tensors:
        if t is not None:
            torch.autograd.graph.increment_version(t)


class launch_scope:
    """A run of launches on ONE tensor's device and the current stream (the forward or the backward of a fused training
    stack: ~10-20 launches): the stream handle is looked up once and the per-launch device guard is skipped -- these
    lookups are a third of a wrapper call's host time, and the training step of the backbones is host-bound."""

    def __init__(self, t):
        self.t = t
        self.guard = _on(t)

    def __enter__(self):
        self.prev = _TLS.scope
        self.guard.__enter__()
        _TLS.scope = (self.t.device.index, _lib.raw_stream(self.t.device))

    def __exit__(self, *exc):
        _TLS.scope = self.prev
        self.guard.__exit__(*exc)


class _on:
    """Make the tensor's device current for the launch (no-op in the one-process-per-GPU case)."""

    def __init__(self, t):
        self.dev = t.device.index
        self.prev = None

    def __enter__(self):
        sc = _TLS.scope
        if sc is not None and sc[0] == self.dev:
            return                                   # inside this thread's launch_scope on this device: already current
        cur = torch.cuda.current_device()
        if self.dev is not None and self.dev != cur:
            self.prev = cur
            torch.cuda.set_device(self.dev)

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)


F32, I32 = torch.float32, torch.int32


def farthest_point_sampling_wrapper(b, n, m, points_tensor, temp_tensor, idx_tensor):
    """sampling.cpp:34-43.  points (B,N,3), temp (B,N) pre-filled 1e10, idx (B,M) -> 1."""
    p, t, i = _ptr(points_tensor, F32, "points"), _ptr(temp_tensor, F32, "temp"), _ptr(idx_tensor, I32, "idx")
    _need(points_tensor, b * n * 3, "points"); _need(temp_tensor, b * n, "temp"); _need(idx_tensor, b * m, "idx")
    # scenes too large for the register-resident kernel (16 384 < n <= 262 144) take the pruned large-scene kernel,
    # which sorts the points into a workspace (20 B per point) -- torch's caching allocator hands it out
    # (6144 .. 16 384 points: a workspace would select the sorting pre-pass; measured, it pays for the publishing kernel of the
    #  streamed layer only -- 8 x 16 384 -> 4 096 through this wrapper 1.790 ms without, 1.813 ms with)
    work_floats = int(_L.sps_fps_workspace_floats(n)) if m > 1 and n > 16384 else 0
    with _on(points_tensor):
        if work_floats:
            work = torch.empty((b * work_floats,), dtype=F32, device=points_tensor.device)
            _lib.check(_L.sps_fps_with_workspace(b, n, m, p, t, i, work.data_ptr(), _stream(points_tensor)),
                       "farthest_point_sampling")
        else:
            _lib.check(_L.sps_farthest_point_sampling_kernel_launcher(b, n, m, p, t, i, _stream(points_tensor)),
                       "farthest_point_sampling")
    return 1


def furthest_point_sampling_with_dist_wrapper(b, n, m, points_tensor, temp_tensor, idx_tensor):
    """sampling.cpp:46-56.  points (B,N,N) distance matrix -> 2 (sic)."""
    p, t, i = _ptr(points_tensor, F32, "points"), _ptr(temp_tensor, F32, "temp"), _ptr(idx_tensor, I32, "idx")
    _need(points_tensor, b * n * n, "points"); _need(temp_tensor, b * n, "temp"); _need(idx_tensor, b * m, "idx")
    with _on(points_tensor):
        _lib.check(_L.sps_furthest_point_sampling_with_dist_kernel_launcher(b, n, m, p, t, i, _stream(points_tensor)),
                   "furthest_point_sampling_with_dist")
    return 2


def gather_points_wrapper(b, c, n, npoints, points_tensor, idx_tensor, out_tensor):
    """sampling.cpp:11-20.  points (B,C,N), idx (B,npoints) -> out (B,C,npoints); returns 1."""
    p, i, o = _ptr(points_tensor, F32, "points"), _ptr(idx_tensor, I32, "idx"), _ptr(out_tensor, F32, "out")
    _need(points_tensor, b * c * n, "points"); _need(idx_tensor, b * npoints, "idx"); _need(out_tensor, b * c * npoints, "out")
    with _on(points_tensor):
        _lib.check(_L.sps_gather_points_kernel_launcher_fast(b, c, n, npoints, p, i, o, _stream(points_tensor)),
                   "gather_points")
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out_tensor, idx_tensor, grad_points_tensor):
    """sampling.cpp:22-31.  grad_out (B,C,npoints) scattered (+=) into zeroed grad_points (B,C,N)."""
    g, i, o = _ptr(grad_out_tensor, F32, "grad_out"), _ptr(idx_tensor, I32, "idx"), _ptr(grad_points_tensor, F32, "grad_points")
    _need(grad_out_tensor, b * c * npoints, "grad_out"); _need(idx_tensor, b * npoints, "idx"); _need(grad_points_tensor, b * c * n, "grad_points")
    with _on(grad_out_tensor):
        _lib.check(_L.sps_gather_points_grad_kernel_launcher_fast(b, c, n, npoints, g, i, o, _stream(grad_out_tensor)),
                   "gather_points_grad")
    return 1


# Launches with at least this many points AND centroids per scene go through the cell grid (csrc/ball_query_grid.hip:
# same rows, far fewer pair tests); smaller ones are cheaper on the plain scan.  (0, 0) = always, None = never.
BQ_GRID_MIN = (4096, 4096)


def _grid_workspace(b, n, m, like):
    if BQ_GRID_MIN is None or n < BQ_GRID_MIN[0] or m < BQ_GRID_MIN[1]:
        return None
    ints = _L.sps_ball_query_grid_workspace_ints(b, n, m)
    return torch.empty(ints, dtype=I32, device=like.device) if ints > 0 else None


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz_tensor, xyz_tensor, idx_tensor):
    """ball_query.cpp:32-43.  new_xyz (B,M,3), xyz (B,N,3), idx (B,M,nsample) pre-zeroed -> 1."""
    q, p, i = _ptr(new_xyz_tensor, F32, "new_xyz"), _ptr(xyz_tensor, F32, "xyz"), _ptr(idx_tensor, I32, "idx")
    _need(new_xyz_tensor, b * m * 3, "new_xyz"); _need(xyz_tensor, b * n * 3, "xyz"); _need(idx_tensor, b * m * nsample, "idx")
    with _on(xyz_tensor):
        work = _grid_workspace(b, n, m, xyz_tensor)
        if work is not None:
            _lib.check(_L.sps_ball_query_grid(b, n, m, radius, 0.0, 0, nsample, 0, q, p, i, work.data_ptr(),
                                              _stream(xyz_tensor)), "ball_query_grid")
        else:
            _lib.check(_L.sps_ball_query_kernel_launcher_fast(b, n, m, radius, nsample, q, p, i, _stream(xyz_tensor)),
                       "ball_query")
    return 1


def ball_query_dilated_wrapper(b, n, m, max_radius, min_radius, nsample, new_xyz_tensor, xyz_tensor, idx_tensor):
    """ball_query.cpp:45-57."""
    q, p, i = _ptr(new_xyz_tensor, F32, "new_xyz"), _ptr(xyz_tensor, F32, "xyz"), _ptr(idx_tensor, I32, "idx")
    _need(new_xyz_tensor, b * m * 3, "new_xyz"); _need(xyz_tensor, b * n * 3, "xyz"); _need(idx_tensor, b * m * nsample, "idx")
    with _on(xyz_tensor):
        work = _grid_workspace(b, n, m, xyz_tensor)
        if work is not None:
            _lib.check(_L.sps_ball_query_grid(b, n, m, max_radius, min_radius, 1, nsample, 0, q, p, i, work.data_ptr(),
                                              _stream(xyz_tensor)), "ball_query_grid")
        else:
            _lib.check(_L.sps_ball_query_dilated_kernel_launcher_fast(b, n, m, max_radius, min_radius, nsample, q, p, i,
                                                                      _stream(xyz_tensor)), "ball_query_dilated")
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points_tensor, idx_tensor, out_tensor):
    """group_points.cpp:30-39.  points (B,C,N), idx (B,npoints,nsample) -> out (B,C,npoints,nsample)."""
    p, i, o = _ptr(points_tensor, F32, "points"), _ptr(idx_tensor, I32, "idx"), _ptr(out_tensor, F32, "out")
    _need(points_tensor, b * c * n, "points"); _need(idx_tensor, b * npoints * nsample, "idx"); _need(out_tensor, b * c * npoints * nsample, "out")
    with _on(points_tensor):
        _lib.check(_L.sps_group_points_kernel_launcher_fast(b, c, n, npoints, nsample, p, i, o, _stream(points_tensor)),
                   "group_points")
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out_tensor, idx_tensor, grad_points_tensor):
    """group_points.cpp:18-28."""
    g, i, o = _ptr(grad_out_tensor, F32, "grad_out"), _ptr(idx_tensor, I32, "idx"), _ptr(grad_points_tensor, F32, "grad_points")
    _need(grad_out_tensor, b * c * npoints * nsample, "grad_out"); _need(idx_tensor, b * npoints * nsample, "idx"); _need(grad_points_tensor, b * c * n, "grad_points")
    with _on(grad_out_tensor):
        _lib.check(_L.sps_group_points_grad_kernel_launcher_fast(b, c, n, npoints, nsample, g, i, o,
                                                                 _stream(grad_out_tensor)), "group_points_grad")
    return 1


def index_add_deterministic(grad_out, idx, grad_points):
    """grad_points (B,C,N) += grad_out (B,C,cols) scattered by idx (B,cols) with every target summed in ascending column
    order: the run-to-run reproducible form of group_points_grad / gather_points_grad (idx (B,M,ns) or (B,M) flattened)."""
    B, C, N = grad_points.shape
    cols = idx.numel() // max(B, 1)
    g, i, o = _ptr(grad_out, F32, "grad_out"), _ptr(idx, I32, "idx"), _ptr(grad_points, F32, "grad_points")
    _need(grad_out, B * C * cols, "grad_out")
    work = torch.empty((int(_L.sps_index_add_workspace_ints(B, N, cols)),), dtype=I32, device=grad_out.device)
    with _on(grad_out):
        _lib.check(_L.sps_index_add_deterministic(B, C, N, cols, g, i, o, work.data_ptr(), _stream(grad_out)),
                   "index_add_deterministic")
    return grad_points


def bn_relu_train_fwd(x, weight, bias, eps, momentum, running_mean, running_var):
    """x (B,C,...) contiguous -> (y, mean (C), invstd (C)); running_* updated in place (None = not tracked)."""
    B, C = x.shape[0], x.shape[1]
    L = x.numel() // max(B * C, 1)
    y = torch.empty_like(x)
    mean = torch.empty((C,), dtype=F32, device=x.device)
    invstd = torch.empty((C,), dtype=F32, device=x.device)
    work = torch.empty((int(_L.sps_bn_train_workspace_doubles(B, C, L)),), dtype=torch.float64, device=x.device)
    opt = lambda t: 0 if t is None else _ptr(t, F32, "parameter")
    with _on(x):
        _lib.check(_L.sps_bn_relu_train_fwd(B, C, L, _ptr(x, F32, "x"), opt(weight), opt(bias), float(eps), float(momentum),
                                            opt(running_mean), opt(running_var), mean.data_ptr(), invstd.data_ptr(),
                                            y.data_ptr(), work.data_ptr(), _stream(x)), "bn_relu_train_fwd")
    _bump_versions(running_mean, running_var)
    return y, mean, invstd


def bn_relu_train_bwd(x, dy, mean, invstd, weight, bias):
    """-> (dx, dweight (C), dbias (C))"""
    B, C = x.shape[0], x.shape[1]
    L = x.numel() // max(B * C, 1)
    dx = torch.empty_like(x)
    dweight = torch.empty((C,), dtype=F32, device=x.device)
    dbias = torch.empty((C,), dtype=F32, device=x.device)
    scratch = torch.empty((2 * C,), dtype=F32, device=x.device)
    work = torch.empty((int(_L.sps_bn_train_workspace_doubles(B, C, L)),), dtype=torch.float64, device=x.device)
    with _on(x):
        _lib.check(_L.sps_bn_relu_train_bwd(B, C, L, _ptr(x, F32, "x"), _ptr(dy, F32, "dy"), mean.data_ptr(),
                                            invstd.data_ptr(), 0 if weight is None else _ptr(weight, F32, "weight"),
                                            0 if bias is None else _ptr(bias, F32, "bias"), dx.data_ptr(), dweight.data_ptr(), dbias.data_ptr(), scratch.data_ptr(),
                                            work.data_ptr(), _stream(x)), "bn_relu_train_bwd")
    return dx, dweight, dbias


_CONV_MAPS = {}


def _conv_frag_map(co, ci, transposed, device):
    """Gather indices into [0, W.flatten()] that give the fragment order of csrc/conv1x1_train.hip for A = W (co x ci) or
    A = W^T (ci x co); cached per shape."""
    key = (co, ci, transposed, device)
    m = _CONV_MAPS.get(key)
    if m is None:
        rows, cols = (ci, co) if transposed else (co, ci)          # A is rows x cols
        t = torch.arange((rows + 15) // 16, device=device).view(-1, 1, 1, 1)
        ks = torch.arange((cols + 3) // 4, device=device).view(1, -1, 1, 1)
        q = torch.arange(4, device=device).view(1, 1, 4, 1)
        i = torch.arange(16, device=device).view(1, 1, 1, 16)
        r, k = 16 * t + i, 4 * ks + q                               # A[r][k], lane = 16 q + i
        flat = (k * ci + r) if transposed else (r * ci + k)         # W[k][r] or W[r][k]
        m = torch.where((r < rows) & (k < cols), flat + 1, torch.zeros_like(flat)).reshape(-1)
        _CONV_MAPS[key] = m
    return m


def conv1x1_apply(x, weight2d, transposed):
    """x (B, Cin, ...) contiguous -> (B, Cout, ...): W x (transposed=False, W (Cout, Cin)) or W^T x (True, W (Cin_of_W...))."""
    co, ci = weight2d.shape
    B = x.shape[0]
    cin_x = ci if not transposed else co
    cout = co if not transposed else ci
    L = x.numel() // max(B * cin_x, 1)
    flat = torch.cat([weight2d.new_zeros(1), weight2d.reshape(-1)])
    frag = flat[_conv_frag_map(co, ci, transposed, x.device)]
    out = torch.empty((B, cout) + tuple(x.shape[2:]), dtype=F32, device=x.device)
    with _on(x):
        _lib.check(_L.sps_conv1x1_apply(B, cin_x, cout, L, _ptr(x, F32, "x"), frag.data_ptr(), out.data_ptr(), _stream(x)),
                   "conv1x1_apply")
    return out


def conv1x1_wgrad(x, dy):
    """x (B, Cin, ...), dy (B, Cout, ...) -> dW (Cout, Cin)"""
    B, ci, co = x.shape[0], x.shape[1], dy.shape[1]
    L = x.numel() // max(B * ci, 1)
    dw = torch.empty((co, ci), dtype=F32, device=x.device)
    work = torch.empty((int(_L.sps_conv1x1_wgrad_workspace_floats(B, ci, co, L)),), dtype=F32, device=x.device)
    with _on(x):
        _lib.check(_L.sps_conv1x1_wgrad(B, ci, co, L, _ptr(x, F32, "x"), _ptr(dy, F32, "dy"), dw.data_ptr(), work.data_ptr(),
                                        _stream(x)), "conv1x1_wgrad")
    return dw


def pool_max_fwd(x, out, arg):
    """x (..., ns) contiguous -> out (...) row maxima, arg (...) uint8 position of the first maximum."""
    ns = x.shape[-1]
    rows = x.numel() // ns
    _need(out, rows, "out"); _need(arg, rows, "arg")
    if arg.dtype != torch.uint8 or not arg.is_contiguous():
        raise TypeError("arg must be a contiguous uint8 tensor")
    with _on(x):
        _lib.check(_L.sps_pool_max_fwd(rows, ns, _ptr(x, F32, "x"), _ptr(out, F32, "out"), arg.data_ptr(), _stream(x)),
                   "pool_max_fwd")


def pool_max_bwd(grad_out, arg, grad_in):
    ns = grad_in.shape[-1]
    rows = grad_in.numel() // ns
    _need(grad_out, rows, "grad_out"); _need(arg, rows, "arg")
    with _on(grad_out):
        _lib.check(_L.sps_pool_max_bwd(rows, ns, _ptr(grad_out, F32, "grad_out"), arg.data_ptr(),
                                       _ptr(grad_in, F32, "grad_in"), _stream(grad_out)), "pool_max_bwd")


def three_nn_wrapper(b, n, m, unknown_tensor, known_tensor, dist2_tensor, idx_tensor):
    """interpolate.cpp:21-30.  unknown (B,n,3), known (B,m,3) -> dist2, idx (B,n,3); returns None."""
    u, k = _ptr(unknown_tensor, F32, "unknown"), _ptr(known_tensor, F32, "known")
    d, i = _ptr(dist2_tensor, F32, "dist2"), _ptr(idx_tensor, I32, "idx")
    _need(unknown_tensor, b * n * 3, "unknown"); _need(known_tensor, b * m * 3, "known")
    _need(dist2_tensor, b * n * 3, "dist2"); _need(idx_tensor, b * n * 3, "idx")
    with _on(unknown_tensor):
        _lib.check(_L.sps_three_nn_kernel_launcher_fast(b, n, m, u, k, d, i, _stream(unknown_tensor)), "three_nn")


def three_interpolate_wrapper(b, c, m, n, points_tensor, idx_tensor, weight_tensor, out_tensor):
    """interpolate.cpp:32-44.  points (B,C,m), idx/weight (B,n,3) -> out (B,C,n); returns None."""
    p, i = _ptr(points_tensor, F32, "points"), _ptr(idx_tensor, I32, "idx")
    w, o = _ptr(weight_tensor, F32, "weight"), _ptr(out_tensor, F32, "out")
    _need(points_tensor, b * c * m, "points"); _need(idx_tensor, b * n * 3, "idx")
    _need(weight_tensor, b * n * 3, "weight"); _need(out_tensor, b * c * n, "out")
    with _on(points_tensor):
        _lib.check(_L.sps_three_interpolate_kernel_launcher_fast(b, c, m, n, p, i, w, o, _stream(points_tensor)),
                   "three_interpolate")


def three_interpolate_grad_wrapper(b, c, n, m, grad_out_tensor, idx_tensor, weight_tensor, grad_points_tensor):
    """interpolate.cpp:46-58."""
    g, i = _ptr(grad_out_tensor, F32, "grad_out"), _ptr(idx_tensor, I32, "idx")
    w, o = _ptr(weight_tensor, F32, "weight"), _ptr(grad_points_tensor, F32, "grad_points")
    _need(grad_out_tensor, b * c * n, "grad_out"); _need(idx_tensor, b * n * 3, "idx")
    _need(weight_tensor, b * n * 3, "weight"); _need(grad_points_tensor, b * c * m, "grad_points")
    with _on(grad_out_tensor):
        _lib.check(_L.sps_three_interpolate_grad_kernel_launcher_fast(b, c, n, m, g, i, w, o, _stream(grad_out_tensor)),
                   "three_interpolate_grad")


# ---- fused entry points (not part of the reference extension) -----------------------------

def score_topk(cls_features, npoint, stds=None, return_scores=False, xyz=None):
    """Fused sampler of pointnet2_modules.py:287-303: cls (B,N,C) [, stds (B,N)] -> idx (B,npoint) i32.
    xyz (B,N,3): also return the centroid rows xyz[idx] (B,npoint,3) from the same launch (the gather that follows the
    sampler, :423-424) -> (idx, new_xyz)."""
    c = _ptr(cls_features, F32, "cls_features")
    B, N, C = cls_features.shape
    s = 0
    if stds is not None:
        s = _ptr(stds, F32, "stds")
        _need(stds, B * N, "stds")
    idx = torch.empty((B, npoint), dtype=I32, device=cls_features.device)
    scores = torch.empty((B, N), dtype=F32, device=cls_features.device) if return_scores else None
    new_xyz = None
    if xyz is not None:
        _need(xyz, B * N * 3, "xyz")
        new_xyz = torch.empty((B, npoint, 3), dtype=F32, device=cls_features.device)
    with _on(cls_features):
        _lib.check(_L.sps_score_topk_gather(B, N, C, npoint, c, s, 0 if xyz is None else _ptr(xyz, F32, "xyz"), idx.data_ptr(),
                                            0 if new_xyz is None else new_xyz.data_ptr(),
                                            scores.data_ptr() if scores is not None else 0, _stream(cls_features)),
                   "score_topk")
    out = (idx, scores) if return_scores else idx
    return (out, new_xyz) if xyz is not None else out


def fps_ordered_prefix(xyz, npoint, return_flags=False):
    """D-FPS of an FPS-ordered cloud (B,N,3) -> (B,npoint) int32: checks the identity-prefix guess in parallel and
    recomputes only scenes where fp32 distance ties break it.  Bit-identical to farthest_point_sampling_wrapper."""
    p = _ptr(xyz, F32, "xyz")
    B, N, _ = xyz.shape
    dev = xyz.device
    temp = torch.full((B, N), 1e10, dtype=F32, device=dev)
    idx = torch.empty((B, npoint), dtype=I32, device=dev)
    work_t = torch.empty((B, max(npoint, 1)), dtype=F32, device=dev)
    work_temp = torch.empty((B, N), dtype=F32, device=dev)
    flags = torch.empty((B,), dtype=I32, device=dev)
    with _on(xyz):
        _lib.check(_L.sps_fps_ordered_prefix(B, N, npoint, p, temp.data_ptr(), idx.data_ptr(), work_t.data_ptr(),
                                             work_temp.data_ptr(), flags.data_ptr(), _stream(xyz)), "fps_ordered_prefix")
    return (idx, flags, temp) if return_flags else idx


class OrderedPrefix:
    """sps_fps_ordered_prefix in two steps for callers that receive the FPS-ordered cloud piecewise: begin() once the
    first `npoint` points of every scene exist, finish() once all of them do (possibly on another stream, ordered
    after begin by the caller)."""

    def __init__(self, xyz, npoint):
        B, N, _ = xyz.shape
        dev = xyz.device
        self.xyz, self.npoint = xyz, npoint
        self.temp = torch.full((B, N), 1e10, dtype=F32, device=dev)
        self.idx = torch.empty((B, npoint), dtype=I32, device=dev)
        self.work_t = torch.empty((B, max(npoint, 1)), dtype=F32, device=dev)
        self.work_temp = torch.empty((B, N), dtype=F32, device=dev)
        self.flags = torch.empty((B,), dtype=I32, device=dev)
        self.begun, self.checked = False, 0      # checked: points [0, checked) went through the second pass already

    def tensors(self):
        return (self.temp, self.idx, self.work_t, self.work_temp, self.flags)

    def begin(self):
        B, N, _ = self.xyz.shape
        with _on(self.xyz):
            _lib.check(_L.sps_fps_ordered_prefix_begin(B, N, self.npoint, _ptr(self.xyz, F32, "xyz"), self.temp.data_ptr(),
                                                        self.work_t.data_ptr(), self.flags.data_ptr(), _stream(self.xyz)),
                       "fps_ordered_prefix_begin")
        self.begun, self.checked = True, 0

    def check_upto(self, k1):
        """The second pass for the points [checked, k1) of every scene -- they exist already, the rest does not yet (k1 a
        multiple of 64; after begin(), stream-ordered behind it): finish() then has only the remaining points to check."""
        B, N, _ = self.xyz.shape
        k1 = min(int(k1), N) & ~63
        if not self.begun or k1 <= self.checked or k1 >= N:
            return
        with _on(self.xyz):
            _lib.check(_L.sps_fps_ordered_prefix_check_range(B, N, self.npoint, self.checked, k1 - self.checked,
                                                              _ptr(self.xyz, F32, "xyz"), self.temp.data_ptr(), self.idx.data_ptr(),
                                                              self.work_t.data_ptr(), self.work_temp.data_ptr(),
                                                              self.flags.data_ptr(), _stream(self.xyz)),
                       "fps_ordered_prefix_check_range")
        self.checked = k1

    def finish(self, force_redo=None):
        """force_redo: device int32; non-zero = the inputs of begin() were not ready, recompute every scene."""
        B, N, _ = self.xyz.shape
        with _on(self.xyz):
            _lib.check(_L.sps_fps_ordered_prefix_finish_from(B, N, self.npoint, self.checked, _ptr(self.xyz, F32, "xyz"),
                                                              self.temp.data_ptr(), self.idx.data_ptr(), self.work_t.data_ptr(),
                                                              self.work_temp.data_ptr(), self.flags.data_ptr(),
                                                              0 if force_redo is None else _ptr(force_redo, I32, "force_redo"),
                                                              _stream(self.xyz)), "fps_ordered_prefix_finish")
        return self.idx


def current_stream_handle(t):
    return _stream(t)


# sort the scenes of the register-resident PUBLISHING FPS (streamed first layer) in a pre-pass of 8 workgroups per scene
# (fps_presort.hip): 2.219 -> 2.194 ms per pass at the bench shape
PRESORT = os.environ.get("SPS_FPS_PRESORT", "1") != "0"
ORDERED_PREFIX_MAX = 7168   # centres the identity-prefix verification stages in LDS (fps_verify.hip FV_MAX_M)


def fps_publish(xyz, temp, idx, progress, presort=True):
    """Launch the publishing FPS (sps_fps_publish / sps_fps_publish_ws) on the current stream; all tensors are
    caller-allocated.  temp = None: the running distances start at 1e10 and are not handed back (no fill launch in front of
    the producer).  Scenes of more than 16 384 points take the clustered large-scene kernel, whose workspace is returned
    (keep it alive until the producer has finished; record_stream it).  presort=False: no sorting pre-pass for scenes of
    up to 16 384 points (its workgroups spin on each other and must all be resident: not on a CU-masked stream)."""
    B, N, _ = xyz.shape
    tp = 0 if temp is None else _ptr(temp, F32, "temp")
    _lib.ensure_init(xyz.device)      # (the sorting pre-pass's flag pool: once per device, never under stream capture)
    with _on(xyz):
        wf = int(_L.sps_fps_workspace_floats(N))
        if wf > 0 and (N > 16384 or (PRESORT and presort)):   # large scenes: sorted points; 6144 .. 16 384: the sorting pre-pass's output
            work = torch.empty((B * wf,), dtype=F32, device=xyz.device)
            _lib.check(_L.sps_fps_publish_ws(B, N, idx.shape[1], _ptr(xyz, F32, "xyz"), tp, _ptr(idx, I32, "idx"),
                                             _ptr(progress, I32, "progress"), work.data_ptr(), _stream(xyz)), "fps_publish")
            return work
        _lib.check(_L.sps_fps_publish(B, N, idx.shape[1], _ptr(xyz, F32, "xyz"), tp, _ptr(idx, I32, "idx"),
                                      _ptr(progress, I32, "progress"), _stream(xyz)), "fps_publish")
    return None


def fps_can_publish(B, N):
    """Is there a publishing FPS kernel for B scenes of N points?  (6144 .. 16 384 points: the register-resident kernel;
    up to 262 144: the clustered kernel, which needs every scene's workgroups resident at once: B <= 16.)"""
    return 6144 <= N <= 16384 or (16384 < N <= 262144 and B <= 16)


def fps_redo_where(xyz, idx, redo, temp=None):
    """The ordinary D-FPS of xyz (B,N,3) into idx (B,m) for the scenes with redo[scene] != 0 (device int32 (B,)); the others
    keep their idx.  Normally a launch that does nothing (see sa_stack._streamed_first_layer).  temp: (B,N) filled with 1e10."""
    B, N, _ = xyz.shape
    if temp is None:
        temp = torch.full((B, N), 1e10, dtype=F32, device=xyz.device)
    with _on(xyz):
        _lib.check(_L.sps_fps_redo_where(B, N, idx.shape[1], _ptr(xyz, F32, "xyz"), temp.data_ptr(), _ptr(idx, I32, "idx"),
                                         _ptr(redo, I32, "redo"), _stream(xyz)), "fps_redo_where")


def wait_progress(progress, need, timed_out, patient=False):
    """Enqueue a bounded device-side wait on the current stream until all scenes published `need` samples; on giving up
    it sets timed_out (B,) -- every entry."""
    with _on(progress):
        _need(timed_out, progress.numel(), "timed_out")
        _lib.check(_L.sps_wait_progress_ex(_ptr(progress, I32, "progress"), progress.numel(), need,
                                           _ptr(timed_out, I32, "timed_out"), 1 if patient else 0, _stream(progress)),
                   "wait_progress")


def _flag(run_if):
    return 0 if run_if is None else _ptr(run_if, I32, "run_if")


def gather_xyz_range(xyz, idx, out, j0, jcount, run_if=None):
    """run_if (device int32): the launch does nothing while it is zero (sa_stack's redo after a timed-out wait)."""
    B, N, _ = xyz.shape
    with _on(xyz):
        _lib.check(_L.sps_gather_xyz_range(B, N, idx.shape[1], j0, jcount, _ptr(xyz, F32, "xyz"), _ptr(idx, I32, "idx"),
                                           _ptr(out, F32, "out"), _flag(run_if), _stream(xyz)), "gather_xyz_range")


def ball_query_full2_range(radius_a, radius_b, xyz, new_xyz, idx_a, idx_b, j0, jcount, run_if=None, full_range_if=None,
                           gather_idx=None):
    """gather_idx (B, M) int32: the centroids of the range are xyz[gather_idx] and the launch writes them to new_xyz too."""
    B, N, _ = xyz.shape
    with _on(xyz):
        _lib.check(_L.sps_ball_query_full2_range(B, N, new_xyz.shape[1], j0, jcount, radius_a, idx_a.shape[2], radius_b,
                                                 idx_b.shape[2], _ptr(new_xyz, F32, "new_xyz"), _ptr(xyz, F32, "xyz"),
                                                 _ptr(idx_a, I32, "idx_a"), _ptr(idx_b, I32, "idx_b"), 0, _flag(run_if),
                                                 _flag(full_range_if),
                                                 0 if gather_idx is None else _ptr(gather_idx, I32, "gather_idx"),
                                                 _stream(xyz)), "ball_query_full2_range")


def ball_query_full2_points(radius_a, nsample_a, radius_b, nsample_b, xyz, new_xyz, k0, kcount, full_if=None, full_if_any=None,
                            gather_idx=None):
    """Both radii for ALL centroids over the points [k0, k0 + kcount) of every scene only -> (idx_a, idx_b), every row written
    (first nsample hits among those points in index order; zeros when there is none).  full_if (device int32) /
    full_if_any (device int32 array): while either says so the launch scans the whole scene instead (a repair).
    gather_idx (B, M) int32: the centroids are xyz[gather_idx] and the launch writes them into new_xyz (in place)."""
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    ia = torch.empty((B, M, nsample_a), dtype=I32, device=xyz.device)
    ib = torch.empty((B, M, nsample_b), dtype=I32, device=xyz.device)
    with _on(xyz):
        _lib.check(_L.sps_ball_query_full2_points_gather(
            B, N, M, int(k0), int(kcount), radius_a, nsample_a, radius_b, nsample_b, _ptr(new_xyz, F32, "new_xyz"),
            _ptr(xyz, F32, "xyz"), ia.data_ptr(), ib.data_ptr(), _flag(full_if),
            0 if full_if_any is None else _ptr(full_if_any, I32, "flags"), 0 if full_if_any is None else full_if_any.numel(),
            0 if gather_idx is None else _ptr(gather_idx, I32, "gather_idx"), _stream(xyz)), "ball_query_full2_points")
    return ia, ib


def gather_xyz(xyz, idx):
    """xyz (B,N,3), idx (B,M) int32 -> (B,M,3): rows of xyz, without the (B,3,N) detour."""
    p, i = _ptr(xyz, F32, "xyz"), _ptr(idx, I32, "idx")
    B, N, _ = xyz.shape
    M = idx.shape[1]
    out = torch.empty((B, M, 3), dtype=F32, device=xyz.device)
    with _on(xyz):
        _lib.check(_L.sps_gather_xyz(B, N, M, p, i, out.data_ptr(), _stream(xyz)), "gather_xyz")
    return out


def ball_query_full(radius, nsample, xyz, new_xyz):
    """Ball query that writes every row (zeros for empty balls) into a fresh (B,M,nsample) int32 tensor."""
    p, q = _ptr(xyz, F32, "xyz"), _ptr(new_xyz, F32, "new_xyz")
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = torch.empty((B, M, nsample), dtype=I32, device=xyz.device)
    with _on(xyz):
        _lib.check(_L.sps_ball_query_full(B, N, M, radius, nsample, q, p, idx.data_ptr(), _stream(xyz)),
                   "ball_query_full")
    return idx


def ball_query_full2(radius_a, nsample_a, radius_b, nsample_b, xyz, new_xyz, spatial_groups=False, wave_per_centroid=False):
    """Two radii in one scan -> (idx_a (B,M,nsample_a), idx_b (B,M,nsample_b)), every row written.
    spatial_groups: process centroids in spatially sorted groups of 64 (same result).  Measured on MI355X
    (tools/bq_time.py, 8 x 4096 centroids over 16384 points): no gain at the IA-SSD radii (0.2/0.8: 353 us either
    way), 1.5x faster at radius 3.0, slower for tiny balls -- hence off by default."""
    p, q = _ptr(xyz, F32, "xyz"), _ptr(new_xyz, F32, "new_xyz")
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    ia = torch.empty((B, M, nsample_a), dtype=I32, device=xyz.device)
    ib = torch.empty((B, M, nsample_b), dtype=I32, device=xyz.device)
    if wave_per_centroid:
        with _on(xyz):
            _lib.check(_L.sps_ball_query_full2_wave(B, N, M, radius_a, nsample_a, radius_b, nsample_b, q, p, ia.data_ptr(),
                                                    ib.data_ptr(), _stream(xyz)), "ball_query_full2_wave")
        return ia, ib
    grid_work = None if spatial_groups else _grid_workspace(B, N, M, xyz)
    if grid_work is not None:  # large cloud AND many centroids: cell grid + bitmap (csrc/ball_query_grid.hip), same rows
        with _on(xyz):
            _lib.check(_L.sps_ball_query_grid2(B, N, M, radius_a, nsample_a, radius_b, nsample_b, q, p, ia.data_ptr(),
                                               ib.data_ptr(), grid_work.data_ptr(), _stream(xyz)), "ball_query_grid2")
        return ia, ib
    work = torch.empty((B, M), dtype=I32, device=xyz.device) if spatial_groups else None
    with _on(xyz):
        _lib.check(_L.sps_ball_query_full2(B, N, M, radius_a, nsample_a, radius_b, nsample_b, q, p, ia.data_ptr(),
                                           ib.data_ptr(), work.data_ptr() if work is not None else 0, _stream(xyz)),
                   "ball_query_full2")
    return ia, ib


def query_and_group(radius, nsample, xyz, new_xyz, features=None, use_xyz=True):
    """Fused QueryAndGroup.forward (pointnet2_utils.py:299-322) -> (new_features, idx)."""
    p, q = _ptr(xyz, F32, "xyz"), _ptr(new_xyz, F32, "new_xyz")
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    C = 0
    f = 0
    if features is not None:
        f = _ptr(features, F32, "features")
        C = features.shape[1]
        _need(features, B * C * N, "features")
    cout = C + (3 if use_xyz else 0)
    idx = torch.empty((B, M, nsample), dtype=I32, device=xyz.device)
    out = torch.empty((B, cout, M, nsample), dtype=F32, device=xyz.device)
    with _on(xyz):
        _lib.check(_L.sps_query_and_group(B, N, M, C, radius, nsample, 1 if use_xyz else 0, p, q, f,
                                          idx.data_ptr(), out.data_ptr(), _stream(xyz)), "query_and_group")
    return out, idx


def group_concat(xyz, new_xyz, features, idx, use_xyz=True):
    """The grouping half of query_and_group on neighbour indices the caller already holds (pointnet2_utils.py:312-320)
    -> (B, 3+C, M, nsample), or (B, C, M, nsample) without use_xyz."""
    p, q = _ptr(xyz, F32, "xyz"), _ptr(new_xyz, F32, "new_xyz")
    B, N, _ = xyz.shape
    M, nsample = idx.shape[1], idx.shape[2]
    _need(idx, B * M * nsample, "idx")
    C = 0
    f = 0
    if features is not None:
        f = _ptr(features, F32, "features")
        C = features.shape[1]
        _need(features, B * C * N, "features")
    out = torch.empty((B, C + (3 if use_xyz else 0), M, nsample), dtype=F32, device=xyz.device)
    with _on(xyz):
        _lib.check(_L.sps_group_concat(B, N, M, C, nsample, 1 if use_xyz else 0, p, q, f, _ptr(idx, I32, "idx"),
                                       out.data_ptr(), _stream(xyz)), "group_concat")
    return out


def group_concat_grad_features(grad_out, idx, n, channel_offset):
    """The feature rows [channel_offset:] of the gradient group_concat's output receives, (B, C_all, M, nsample) contiguous,
    scattered back to the points -> (B, C_all - channel_offset, n): group_points_grad read in place from the wider tensor
    (the reference slices it first: torch.cat's backward, pointnet2_utils.py:316-320)."""
    B, call, M, nsample = grad_out.shape
    C = call - channel_offset
    _need(idx, B * M * nsample, "idx")
    out = torch.zeros((B, C, n), dtype=F32, device=grad_out.device)
    if C > 0 and B > 0:
        cols = M * nsample
        with _on(grad_out):
            _lib.check(_L.sps_group_points_grad_strided(B, C, n, M, nsample, _ptr(grad_out, F32, "grad_out") + 4 * channel_offset * cols,
                                                        call * cols, _ptr(idx, I32, "idx"), out.data_ptr(), _stream(grad_out)),
                       "group_points_grad_strided")
    return out


# ---- the fused train-mode grouped MLP (csrc/mlp_train.hip) ----------------------------------------------------------------
TIN_RAW, TIN_BNRELU, TIN_BNBWD, TIN_BNBWD_POOL = 0, 1, 2, 3
TEPI_NONE, TEPI_STATS, TEPI_BWD = 0, 1, 2
TRAIN_PARAMS = 8     # floats per channel of a parameter block: mean, invstd, scale, shift, gamma, beta, c1, c2


def _p(t):
    return 0 if t is None else t.data_ptr()


def tconv_parts(b, l, co):
    """= sps_tconv_parts (restated here to save the call; test_abi_cpu checks the two against each other)"""
    if b <= 0 or l <= 0 or co <= 0:
        return 0
    return min((b * (l >> 6) + 3) // 4, 256 if co > 64 else 512)


def tconv(w, wamax, in_mode, epi_mode, out, operand=None, y=None, gout=None, arg=None, nsample=0, pin=None, epi_y=None, pout=None,
          transposed=False, overflow=None, amax_in=None, amax_out=None):
    """out (b, co, l) = A . T(operand): one convolution (or data-gradient) launch of the fused train-mode grouped MLP, see
    include/spsnet_sa.h sps_tconv.  -> the per-workgroup statistics (parts, co, 2) float64 for epi_mode != TEPI_NONE."""
    b, co = out.shape[0], out.shape[1]
    l = out.numel() // max(1, b * co)
    ref = y if y is not None else operand
    ci = ref.shape[1]
    for t in (operand, y, epi_y, out):
        if t is not None:
            _ptr(t, F32, "tconv tensor")
    m = gout.shape[2] if gout is not None else 0
    partial = None
    if epi_mode != TEPI_NONE:
        partial = torch.empty((tconv_parts(b, l, co), co, 2), dtype=torch.float64, device=out.device)
    with _on(out):
        _lib.check(_L.sps_tconv(b, ci, co, l, in_mode, epi_mode, 1 if transposed else 0, _ptr(w, F32, "w"), _p(operand), _p(y),
                                _p(gout), _p(arg), nsample, m, _p(pin), out.data_ptr(), _p(epi_y), _p(pout), _p(partial),
                                _p(amax_in), _p(amax_out), wamax.data_ptr(), _p(overflow), _stream(out)), "tconv")
    return partial


def weights_amax(ws):
    """-> (len(ws),) float32: the largest magnitude of each (contiguous fp32) weight matrix, one launch for up to four"""
    out = torch.empty((len(ws),), dtype=F32, device=ws[0].device)
    for k0 in range(0, len(ws), 4):
        grp = list(ws[k0:k0 + 4])
        args = []
        for k in range(4):
            args += [_ptr(grp[k], F32, "w"), grp[k].numel()] if k < len(grp) else [0, 0]
        with _on(out):
            _lib.check(_L.sps_tamax4(len(grp), *args, out[k0:].data_ptr(), _stream(out)), "tamax4")
    return out


def tbn_finalize(partial, count, bn, params, count_dev=None):
    """statistics -> params[:, 0:6] (+ the module's running statistics and batch counter, torch semantics).
    count_dev (device float64 scalar): the GLOBAL element count under SyncBatchNorm, used instead of `count`."""
    parts, c = partial.shape[0], partial.shape[1]
    with _on(params):
        _lib.check(_L.sps_tbn_finalize_dc(c, parts, float(count), 0 if count_dev is None else _ptr(count_dev, torch.float64, "count"),
                                          partial.data_ptr(), _p(bn.weight), _p(bn.bias), float(bn.eps),
                                          float(bn.momentum), _p(bn.running_mean), _p(bn.running_var), params.data_ptr(),
                                          _p(bn.num_batches_tracked), _stream(params)), "tbn_finalize")
    _bump_versions(bn.running_mean, bn.running_var, bn.num_batches_tracked)


def tbn_bwd_finalize(partial, count, params, count_dev=None):
    """BatchNorm-backward sums -> params[:, 6:8]; returns (d gamma, d beta)"""
    parts, c = partial.shape[0], partial.shape[1]
    dg = torch.empty((c,), dtype=F32, device=params.device)
    db = torch.empty((c,), dtype=F32, device=params.device)
    with _on(params):
        _lib.check(_L.sps_tbn_bwd_finalize_dc(c, parts, float(count), 0 if count_dev is None else _ptr(count_dev, torch.float64, "count"),
                                              partial.data_ptr(), params.data_ptr(), dg.data_ptr(),
                                              db.data_ptr(), _stream(params)), "tbn_bwd_finalize")
    return dg, db


def tpool_fwd(y, params):
    """(B, C, M, ns) pre-BatchNorm outputs -> (pooled (B, C, M), arg (B, C, M) u8, y at the arg-max (B, C, M))"""
    B, C, M, ns = y.shape
    out = torch.empty((B, C, M), dtype=F32, device=y.device)
    yarg = torch.empty((B, C, M), dtype=F32, device=y.device)
    arg = torch.empty((B, C, M), dtype=torch.uint8, device=y.device)
    with _on(y):
        _lib.check(_L.sps_tpool_fwd(B, C, M, ns, _ptr(y, F32, "y"), params.data_ptr(), out.data_ptr(), arg.data_ptr(),
                                    yarg.data_ptr(), _stream(y)), "tpool_fwd")
    return out, arg, yarg


def tpool_bwd_stats(yarg, gout, params, amax_out=None):
    """-> the last layer's BatchNorm-backward sums (B, C, 2) float64; amax_out (zeroed scalar) receives max |gout|"""
    B, C, M = yarg.shape
    partial = torch.empty((B, C, 2), dtype=torch.float64, device=yarg.device)
    with _on(yarg):
        _lib.check(_L.sps_tpool_bwd_stats(B, C, M, _ptr(yarg, F32, "yarg"), _ptr(gout, F32, "gout"), params.data_ptr(),
                                          partial.data_ptr(), _p(amax_out), _stream(yarg)), "tpool_bwd_stats")
    return partial


def tbn_apply_relu(y, params):
    """relu(batch_norm(y)) from the parameter block: the output of a stack that has no pool behind it"""
    b, c = y.shape[0], y.shape[1]
    out = torch.empty_like(y)
    with _on(y):
        _lib.check(_L.sps_tbn_apply_relu(b, c, y.numel() // max(1, b * c), _ptr(y, F32, "y"), params.data_ptr(), out.data_ptr(),
                                         _stream(y)), "tbn_apply_relu")
    return out


def tbn_bwd_stats(y, dA, params, amax_out=None):
    """-> the last layer's BatchNorm-backward sums (B, C, 2) float64 from a dense incoming gradient"""
    b, c = y.shape[0], y.shape[1]
    partial = torch.empty((b, c, 2), dtype=torch.float64, device=y.device)
    with _on(y):
        _lib.check(_L.sps_tbn_bwd_stats(b, c, y.numel() // max(1, b * c), _ptr(y, F32, "y"), _ptr(dA, F32, "dA"), params.data_ptr(),
                                        partial.data_ptr(), _p(amax_out), _stream(y)), "tbn_bwd_stats")
    return partial


def twgrad(y, pd, x, px, amax_in, dA=None, gout=None, arg=None, nsample=0, overflow=None):
    """dW (co, ci) of one layer: dY recomputed from (dA | pooled gradient, y, pd), the other operand x raw (px None) or
    through BatchNorm + ReLU (px)."""
    b, co, ci = y.shape[0], y.shape[1], x.shape[1]
    l = y.numel() // max(1, b * co)
    m = gout.shape[2] if gout is not None else 0
    dw = torch.empty((co, ci), dtype=F32, device=y.device)
    work = torch.empty((int(_L.sps_twgrad_workspace_floats(b, co, ci, l)),), dtype=F32, device=y.device)
    with _on(y):
        _lib.check(_L.sps_twgrad(b, co, ci, l, TIN_BNBWD if dA is not None else TIN_BNBWD_POOL,
                                 TIN_BNRELU if px is not None else TIN_RAW, _p(dA), _ptr(y, F32, "y"), _p(gout), _p(arg), nsample,
                                 m, pd.data_ptr(), _ptr(x, F32, "x"), _p(px), amax_in.data_ptr(), dw.data_ptr(), work.data_ptr(),
                                 _p(overflow),
                                 _stream(y)), "twgrad")
    return dw
