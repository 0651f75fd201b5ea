"""Tier-2 oracle: the reference's own Python layers, run on CPU over the C oracle.

TEST INFRASTRUCTURE ONLY, and usable ONLY in the build container: it needs
/root/reference, which never travels to the GPU box.  It is how the golden vectors in
tests/golden/ were produced (oracle/gen_golden.py) and how the C restatement is
cross-checked against the reference's wrappers.

Recipe: the reference's pointnet2_utils.py does
`from . import pointnet2_batch_cuda as pointnet2` (pointnet2_utils.py:7); we register a
stand-in module of that name whose 11 functions have the pybind signatures of
src/pointnet2_api.cpp:10-26 and forward to oracle/sa_oracle.c, alias
torch.cuda.{Int,Float}Tensor (used by the wrappers for allocation,
pointnet2_utils.py:25-26,83,122-123,200,246) to the CPU tensor types, and import the
reference package from /root/reference.
"""
import ctypes
import importlib
import os
import sys
import types

import torch

from . import oracle as O

REFERENCE_ROOT = "/root/reference"
_PKG = "pcdet.ops.pointnet2.pointnet2_batch"


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "pcdet", "ops", "pointnet2", "pointnet2_batch"))


def _fp(t):
    assert t.dtype == torch.float32 and t.is_contiguous() and t.device.type == "cpu"
    return ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_float))


def _ip(t):
    assert t.dtype == torch.int32 and t.is_contiguous() and t.device.type == "cpu"
    return ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_int))


def _standin_module():
    L = O.lib()
    cf = ctypes.c_float
    m = types.ModuleType(_PKG + ".pointnet2_batch_cuda")

    def ball_query_wrapper(b, n, mm, radius, nsample, new_xyz, xyz, idx):
        L.orc_ball_query(b, n, mm, cf(radius), nsample, _fp(new_xyz), _fp(xyz), _ip(idx)); return 1

    def ball_query_dilated_wrapper(b, n, mm, max_radius, min_radius, nsample, new_xyz, xyz, idx):
        L.orc_ball_query_dilated(b, n, mm, cf(max_radius), cf(min_radius), nsample, _fp(new_xyz), _fp(xyz), _ip(idx)); return 1

    def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
        L.orc_group_points(b, c, n, npoints, nsample, _fp(points), _ip(idx), _fp(out)); return 1

    def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
        L.orc_group_points_grad(b, c, n, npoints, nsample, _fp(grad_out), _ip(idx), _fp(grad_points)); return 1

    def gather_points_wrapper(b, c, n, npoints, points, idx, out):
        L.orc_gather_points(b, c, n, npoints, _fp(points), _ip(idx), _fp(out)); return 1

    def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
        L.orc_gather_points_grad(b, c, n, npoints, _fp(grad_out), _ip(idx), _fp(grad_points)); return 1

    def farthest_point_sampling_wrapper(b, n, mm, points, temp, idx):
        L.orc_fps(b, n, mm, _fp(points), _fp(temp), _ip(idx)); return 1

    def furthest_point_sampling_with_dist_wrapper(b, n, mm, points, temp, idx):
        L.orc_fps_with_dist(b, n, mm, _fp(points), _fp(temp), _ip(idx)); return 2

    def three_nn_wrapper(b, n, mm, unknown, known, dist2, idx):
        L.orc_three_nn(b, n, mm, _fp(unknown), _fp(known), _fp(dist2), _ip(idx))

    def three_interpolate_wrapper(b, c, mm, n, points, idx, weight, out):
        L.orc_three_interpolate(b, c, mm, n, _fp(points), _ip(idx), _fp(weight), _fp(out))

    def three_interpolate_grad_wrapper(b, c, n, mm, grad_out, idx, weight, grad_points):
        L.orc_three_interpolate_grad(b, c, n, mm, _fp(grad_out), _ip(idx), _fp(weight), _fp(grad_points))

    for f in (ball_query_wrapper, ball_query_dilated_wrapper, group_points_wrapper,
              group_points_grad_wrapper, gather_points_wrapper, gather_points_grad_wrapper,
              farthest_point_sampling_wrapper, furthest_point_sampling_with_dist_wrapper,
              three_nn_wrapper, three_interpolate_wrapper, three_interpolate_grad_wrapper):
        setattr(m, f.__name__, f)
    return m


_loaded = None


def load_reference():
    """-> (pointnet2_utils, pointnet2_modules) of the reference, running on CPU over the oracle."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present (Tier-2 oracle is build-container only)")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # the wrappers allocate with the legacy torch.cuda.*Tensor constructors
    torch.cuda.IntTensor = torch.IntTensor
    torch.cuda.FloatTensor = torch.FloatTensor
    pkg = importlib.import_module(_PKG)
    standin = _standin_module()
    sys.modules[_PKG + ".pointnet2_batch_cuda"] = standin
    setattr(pkg, "pointnet2_batch_cuda", standin)
    utils = importlib.import_module(_PKG + ".pointnet2_utils")
    modules = importlib.import_module(_PKG + ".pointnet2_modules")
    _loaded = (utils, modules)
    return _loaded
