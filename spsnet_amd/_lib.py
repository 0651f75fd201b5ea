"""ctypes binding of libspsnet_sa.so (the C ABI declared in include/spsnet_sa.h).

Fails loudly when the library is missing: there is no fallback path.
"""
import ctypes
import os

# torch bundles its own libamdhip64.so.7; importing it FIRST makes the loader bind libspsnet_sa's
# libamdhip64 dependency to that same runtime instance (one HIP runtime per process -- otherwise
# device pointers and streams handed over from torch belong to a different runtime).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SPS_LIBSPSNET_SA: a DIAGNOSTIC build of the same library -- e.g. fps_pruned_cluster.hip with -DSPS_PC_PROFILE, tools/fps_cluster_profile.py)
LIB_PATH = os.environ.get("SPS_LIBSPSNET_SA") or os.path.join(_HERE, "lib", "libspsnet_sa.so")

SPS_OK = 0
ABI_VERSION = 2

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float

# name -> argument types (every entry point returns int status unless noted)
_PROTOS = {
    "sps_init": [_i, _vp],
    "sps_mlp_train_forward": [_vp, _vp],
    "sps_mlp_train_backward": [_vp, _vp],
    "sps_is_initialized": [_i],
    "sps_set_fps_mode": [_i],
    "sps_debug_where": [_i, _i, _i, _vp, _vp],
    "sps_stream_create_cu_mask": [_i, _vp, _vp],
    "sps_stream_destroy": [_vp],
    "sps_streams_run_concurrently": [_vp, _vp, _vp, _i, _vp],
    "sps_fps_ordered_prefix": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_fps_ordered_prefix_begin": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "sps_fps_ordered_prefix_finish": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_fps_ordered_prefix_check_range": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_fps_ordered_prefix_finish_from": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_debug_fps_profile": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "sps_farthest_point_sampling_kernel_launcher": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "sps_furthest_point_sampling_with_dist_kernel_launcher": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "sps_gather_points_kernel_launcher_fast": [_i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sps_gather_points_grad_kernel_launcher_fast": [_i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sps_ball_query_kernel_launcher_fast": [_i, _i, _i, _f, _i, _vp, _vp, _vp, _vp],
    "sps_ball_query_dilated_kernel_launcher_fast": [_i, _i, _i, _f, _f, _i, _vp, _vp, _vp, _vp],
    "sps_group_points_kernel_launcher_fast": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sps_group_points_grad_kernel_launcher_fast": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sps_three_nn_kernel_launcher_fast": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "sps_three_interpolate_kernel_launcher_fast": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "sps_three_interpolate_grad_kernel_launcher_fast": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "sps_score_topk": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "sps_score_topk_gather": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_query_and_group": [_i, _i, _i, _i, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_group_concat": [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_group_points_grad_strided": [_i, _i, _i, _i, _i, _vp, ctypes.c_longlong, _vp, _vp, _vp],
    "sps_gather_xyz": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "sps_ball_query_full": [_i, _i, _i, _f, _i, _vp, _vp, _vp, _vp],
    "sps_ball_query_full2": [_i, _i, _i, _f, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_ball_query_full2_wave": [_i, _i, _i, _f, _i, _f, _i, _vp, _vp, _vp, _vp, _vp],
    "sps_sa_group_mlp": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                         _i, _i, _vp],
    "sps_sa_group_mlp_supported": [_i, _i, _i],
    "sps_set_mlp_precision": [_i],
    "sps_set_train_precision": [_i],
    "sps_index_add_deterministic": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "sps_sa_group_mlp_supported_stream": [_i, _i, _i, _i],
    "sps_sa_group_mlp_pm_supported": [_i, _i, _i, _i, _i],
    "sps_sa_layer1_per_point": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "sps_sa_layer1_per_point_supported": [_i, _i, _i],
    "sps_fps_with_workspace": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "sps_pointwise_mlp_range": [_i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_pointwise_mlp_ex": [_i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp],
    "sps_pointwise_mlp": [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_ball_query_grid": [_i, _i, _i, _f, _f, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "sps_ball_query_grid2": [_i, _i, _i, _f, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_ball_query_kernel_launcher_stack": [_i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_voxel_query_kernel_launcher_stack": [_i, _i, _i, _i, _i, _f, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_stack_farthest_point_sampling_kernel_launcher": [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_group_points_kernel_launcher_stack": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_group_points_grad_kernel_launcher_stack": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_three_nn_kernel_launcher_stack": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_three_interpolate_kernel_launcher_stack": [_i, _i, _vp, _vp, _vp, _vp, _vp],
    "sps_three_interpolate_grad_kernel_launcher_stack": [_i, _i, _vp, _vp, _vp, _vp, _vp],
    "sps_fp_module_mlp": [_i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_fp_module_mlp_ex": [_i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "sps_tconv_parts": [_i, ctypes.c_longlong, _i],
    "sps_tconv": [_i, _i, _i, ctypes.c_longlong, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                  _vp, _vp],
    "sps_tbn_finalize": [_i, _i, ctypes.c_double, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp],
    "sps_tamax4": [_i, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, _vp],
    "sps_tbn_bwd_finalize": [_i, _i, ctypes.c_double, _vp, _vp, _vp, _vp, _vp],
    "sps_tbn_finalize_dc": [_i, _i, ctypes.c_double, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp],
    "sps_tbn_bwd_finalize_dc": [_i, _i, ctypes.c_double, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_tbn_apply_relu": [_i, _i, ctypes.c_longlong, _vp, _vp, _vp, _vp],
    "sps_tbn_bwd_stats": [_i, _i, ctypes.c_longlong, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_tpool_fwd": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_tpool_bwd_stats": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_twgrad": [_i, _i, _i, ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_pool_max_fwd": [ctypes.c_longlong, _i, _vp, _vp, _vp, _vp],
    "sps_pool_max_bwd": [ctypes.c_longlong, _i, _vp, _vp, _vp, _vp],
    "sps_bn_relu_train_fwd": [_i, _i, ctypes.c_longlong, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_bn_relu_train_bwd": [_i, _i, ctypes.c_longlong, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_query_stacked_local_neighbor_idxs_kernel_launcher_stack": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _i, _i, _i, _i, _vp],
    "sps_query_three_nn_by_stacked_local_idxs_kernel_launcher_stack": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "sps_vector_pool_kernel_launcher_stack": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _i, _i, _i, _i, _i,
                                              _i, _i, _i, _i, _i, _vp, _vp],
    "sps_vector_pool_grad_kernel_launcher_stack": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "sps_dense_edge_conv_bwd": [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_dense_edge_conv_bwd_blocks": [],
    "sps_linear_rows_bwd": [ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp],
    "sps_linear_rows_bwd_blocks": [],
    "sps_conv1x1_apply": [_i, _i, _i, ctypes.c_longlong, _vp, _vp, _vp, _vp],
    "sps_conv1x1_wgrad": [_i, _i, _i, ctypes.c_longlong, _vp, _vp, _vp, _vp, _vp],
    "sps_dense_edge_conv": [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_linear_rows": [ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _i, _vp, _vp],
    "sps_sa_group_mlp_ex": [_i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp,
                            _vp, _vp, _i, _i, _i, _vp, _vp],
    "sps_pack_columns": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, ctypes.c_longlong, _vp],
    "sps_pack_columns2": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _vp, _vp, _vp, _vp, ctypes.c_longlong, _vp],
    "sps_pack_columns2_late": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                               ctypes.c_longlong, _vp, _vp, _i, _vp],
    "sps_ball_query_full2_points": [_i, _i, _i, _i, _i, _f, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "sps_ball_query_full2_points_gather": [_i, _i, _i, _i, _i, _f, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "sps_sa_group_mlp_packed_merge": [_i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _i, _i,
                                      _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "sps_sa_group_mlp_packed": [_i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _i, _i,
                                _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sps_fps_redo_where": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "sps_fps_publish": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "sps_fps_publish_ws": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_wait_progress": [_vp, _i, _i, _vp, _vp],
    "sps_wait_progress_ex": [_vp, _i, _i, _vp, _i, _vp],
    "sps_gather_xyz_range": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "sps_ball_query_full2_range": [_i, _i, _i, _i, _i, _f, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sps_sa_group_mlp_range": [_i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp,
                               _vp, _vp, _i, _i, _vp],
}
EXPORTS = ["sps_abi_version", "sps_last_error", "sps_opt_n_threads", "sps_fps_workspace_floats",
           "sps_index_add_workspace_ints", "sps_ball_query_grid_workspace_ints", "sps_bn_train_workspace_doubles", "sps_conv1x1_wgrad_workspace_floats",
           "sps_pack_columns_capacity", "sps_debug_set_wait_spins", "sps_debug_set_exchange_spins",
           "sps_twgrad_workspace_floats", "sps_mlp_train_partial_doubles", "sps_struct_size"] + list(_PROTOS)

_lib = None


class MlpTrainDesc(ctypes.Structure):
    """struct sps_mlp_train_desc of include/spsnet_sa.h (one call per grouped MLP of a training step)"""
    _fields_ = [("n", _i), ("b", _i), ("m", _i), ("ns", _i), ("c", _i * 5),
                ("w", _vp * 4), ("gamma", _vp * 4), ("beta", _vp * 4), ("eps", _f * 4), ("momentum", _f * 4),
                ("running_mean", _vp * 4), ("running_var", _vp * 4), ("num_batches_tracked", _vp * 4),
                ("x", _vp), ("y", _vp * 4), ("params", _vp * 4), ("wamax", _vp), ("partial", _vp),
                ("out", _vp), ("arg", _vp), ("yarg", _vp), ("overflow", _vp),
                ("gout", _vp), ("dA", _vp * 4), ("dw", _vp * 4), ("dgamma", _vp * 4), ("dbeta", _vp * 4),
                ("amax", _vp), ("work", _vp)]


DESC_MIRRORS = [MlpTrainDesc]       # index = `which` of sps_struct_size()


class SpsError(RuntimeError):
    """A libspsnet_sa call returned a non-zero status (the reference would exit(-1))."""


def load():
    """Load libspsnet_sa.so and declare the prototypes.  Raises ImportError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C spsnet_amd/csrc` "
            "(or __graft_entry__.build()).  spsnet_amd has no CPU/PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.sps_abi_version.restype = _i
    lib.sps_abi_version.argtypes = []
    lib.sps_last_error.restype = ctypes.c_char_p
    lib.sps_last_error.argtypes = []
    lib.sps_opt_n_threads.restype = _i
    lib.sps_opt_n_threads.argtypes = [_i]
    lib.sps_fps_workspace_floats.restype = ctypes.c_longlong
    lib.sps_fps_workspace_floats.argtypes = [_i]
    lib.sps_index_add_workspace_ints.restype = ctypes.c_longlong
    lib.sps_ball_query_grid_workspace_ints.argtypes = [_i, _i, _i]
    lib.sps_ball_query_grid_workspace_ints.restype = ctypes.c_longlong
    lib.sps_bn_train_workspace_doubles.argtypes = [_i, _i, ctypes.c_longlong]
    lib.sps_bn_train_workspace_doubles.restype = ctypes.c_longlong
    lib.sps_conv1x1_wgrad_workspace_floats.argtypes = [_i, _i, _i, ctypes.c_longlong]
    lib.sps_conv1x1_wgrad_workspace_floats.restype = ctypes.c_longlong
    lib.sps_index_add_workspace_ints.argtypes = [_i, _i, _i]
    lib.sps_debug_set_wait_spins.argtypes = [ctypes.c_uint]
    lib.sps_debug_set_wait_spins.restype = ctypes.c_uint
    lib.sps_debug_set_exchange_spins.argtypes = [ctypes.c_uint]
    lib.sps_debug_set_exchange_spins.restype = ctypes.c_uint
    lib.sps_twgrad_workspace_floats.argtypes = [_i, _i, _i, ctypes.c_longlong]
    lib.sps_twgrad_workspace_floats.restype = ctypes.c_longlong
    lib.sps_mlp_train_partial_doubles.argtypes = [_vp]
    lib.sps_mlp_train_partial_doubles.restype = ctypes.c_longlong
    lib.sps_struct_size.argtypes = [_i]
    lib.sps_struct_size.restype = ctypes.c_longlong
    lib.sps_pack_columns_capacity.argtypes = [_i, _i, _i]
    lib.sps_pack_columns_capacity.restype = ctypes.c_longlong
    for name, args in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = _i
        fn.argtypes = args
    got = lib.sps_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"libspsnet_sa ABI {got} != expected {ABI_VERSION}; rebuild spsnet_amd/csrc")
    for which, mirror in enumerate(DESC_MIRRORS):
        if lib.sps_struct_size(which) != ctypes.sizeof(mirror):
            raise ImportError(f"{mirror.__name__}: {ctypes.sizeof(mirror)} bytes here, {lib.sps_struct_size(which)} in libspsnet_sa; "
                              "rebuild spsnet_amd/csrc")
    _lib = lib
    return lib


def check(status, what):
    if status != SPS_OK:
        msg = load().sps_last_error().decode("utf-8", "replace")
        raise SpsError(f"{what} failed (status {status}): {msg}")


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def raw_stream(device) -> int:
    """Handle of torch's CURRENT stream on `device` (what every launch is enqueued on).  torch.cuda.current_stream() builds a
    Stream object per call -- ~2.5 us, a hundred times per pass -- where the raw handle is one C call."""
    idx = device.index
    if _RAW_STREAM is not None and idx is not None:
        return _RAW_STREAM(idx)
    return torch.cuda.current_stream(device).cuda_stream


_INITIALISED = set()


def ensure_init(device) -> bool:
    """sps_init() for `device`, once per process and device: the library's one allocating / synchronising set-up call (the
    flag pool of the FPS sorting pre-pass).  Called by the wrappers in front of their first FPS launch on a device -- but never
    while the current stream is being captured into a graph (the call allocates and synchronises): a capture of a fresh
    process's first pass then records the FPS kernel that sorts for itself; call `spsnet_amd.init(device)` before capturing to
    record the pre-pass.  -> whether the device is initialised."""
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    if idx in _INITIALISED:
        return True
    if torch.cuda.is_current_stream_capturing():
        return bool(load().sps_is_initialized(idx))
    check(load().sps_init(idx, ctypes.c_void_p(raw_stream(torch.device("cuda", idx)))), "sps_init")
    _INITIALISED.add(idx)
    return True
