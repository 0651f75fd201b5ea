"""The C-ABI library loads on a GPU-less host and exports exactly what include/spsnet_sa.h declares.
No kernel is launched here; argument validation is exercised because it runs before any HIP call."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(headers=("spsnet_sa.h", "spsnet_sa_debug.h")):
    names = set()
    for header in headers:
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(sps_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_diagnostics_live_in_their_own_header():
    """The drop-in header declares no sps_debug_* entry point; the diagnostic header declares nothing else."""
    assert not [n for n in declared_functions(("spsnet_sa.h",)) if n.startswith("sps_debug_")]
    assert all(n.startswith("sps_debug_") for n in declared_functions(("spsnet_sa_debug.h",)))


def test_descriptor_mirrors_have_the_library_size():
    import ctypes as C
    from spsnet_amd import _lib
    L = _lib.load()
    for which, mirror in enumerate(_lib.DESC_MIRRORS):
        assert L.sps_struct_size(which) == C.sizeof(mirror), mirror.__name__
    assert L.sps_struct_size(len(_lib.DESC_MIRRORS)) == -1


def test_header_declares_the_reference_surface():
    names = declared_functions()
    # the 11 launchers behind pointnet2_api.cpp:10-26
    for stem in ("farthest_point_sampling_kernel_launcher", "furthest_point_sampling_with_dist_kernel_launcher",
                 "gather_points_kernel_launcher_fast", "gather_points_grad_kernel_launcher_fast",
                 "ball_query_kernel_launcher_fast", "ball_query_dilated_kernel_launcher_fast",
                 "group_points_kernel_launcher_fast", "group_points_grad_kernel_launcher_fast",
                 "three_nn_kernel_launcher_fast", "three_interpolate_kernel_launcher_fast",
                 "three_interpolate_grad_kernel_launcher_fast"):
        assert "sps_" + stem in names


def test_library_exports_every_declared_symbol():
    from spsnet_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/spsnet_sa.h but not exported"
    assert set(_lib.EXPORTS) == set(declared_functions())
    assert _lib.load().sps_abi_version() == _lib.ABI_VERSION


def test_opt_n_threads_matches_reference_rule():
    from spsnet_amd import _lib
    L = _lib.load()
    assert [L.sps_opt_n_threads(n) for n in (1, 3, 64, 1000, 1024, 16384)] == [1, 2, 64, 512, 1024, 1024]


def test_invalid_arguments_are_reported_not_fatal():
    from spsnet_amd import _lib
    L = _lib.load()
    assert L.sps_farthest_point_sampling_kernel_launcher(-1, 16, 4, None, None, None, None) == 1
    assert b"bad shape" in L.sps_last_error()
    assert L.sps_farthest_point_sampling_kernel_launcher(1, 16, 4, None, None, None, None) == 1
    assert b"null" in L.sps_last_error()
    assert L.sps_score_topk(1, 100000, 3, 10, None, None, None, None, None) == 1
    with pytest.raises(_lib.SpsError):
        _lib.check(1, "demo")
    # empty work is a successful no-op, like the reference's `if (m <= 0) return`
    assert L.sps_farthest_point_sampling_kernel_launcher(0, 16, 4, None, None, None, None) == 0
    assert L.sps_ball_query_kernel_launcher_fast(2, 16, 0, 1.0, 4, None, None, None, None) == 0


def test_extension_module_surface():
    import spsnet_amd.pointnet2_batch_cuda as ext
    for fn in ("ball_query_wrapper", "ball_query_dilated_wrapper", "group_points_wrapper", "group_points_grad_wrapper",
               "gather_points_wrapper", "gather_points_grad_wrapper", "farthest_point_sampling_wrapper",
               "furthest_point_sampling_with_dist_wrapper", "three_nn_wrapper", "three_interpolate_wrapper",
               "three_interpolate_grad_wrapper"):
        assert callable(getattr(ext, fn))
    import torch
    with pytest.raises(ValueError):  # CPU tensors are rejected, never silently computed on the host
        ext.farthest_point_sampling_wrapper(1, 8, 2, torch.zeros(1, 8, 3), torch.zeros(1, 8), torch.zeros(1, 2, dtype=torch.int32))


def test_module_mirror_builds_with_reference_state_dict_keys():
    import numpy as np
    from spsnet_amd import pointnet2_modules as M
    g = np.load(os.path.join(ROOT, "tests", "golden", "sampler_ctr.npz"))
    ref_keys = sorted(k[3:] for k in g.files if k.startswith("sd."))
    mod = M.PointnetSAModuleMSG_WithSampling(
        npoint_list=[256], sample_range_list=[-1], sample_type_list=['ctr_aware'], radii=[1.6, 4.8], nsamples=[8, 16],
        mlps=[[6, 8, 16], [6, 8, 24]], use_xyz=True, dilated_group=False, aggregation_mlp=[32], confidence_mlp=[16],
        num_class=3)
    assert sorted(mod.state_dict().keys()) == ref_keys
    for k, v in mod.state_dict().items():
        assert tuple(v.shape) == tuple(g["sd." + k].shape), k


def test_surface_feature_mirror_has_reference_state_dict_keys():
    import numpy as np
    from spsnet_amd import surface_feature as SF
    g = np.load(os.path.join(ROOT, "tests", "golden", "surface_feature.npz"))
    ref = {k[len("sd_static."):]: g[k].shape for k in g.files if k.startswith("sd_static.")}
    net = SF.FeatureExtraction(dynamic_graph=False)
    mine = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert mine == {k: tuple(v) for k, v in ref.items()}
    assert net.out_channels == 60  # the "surface feature C=60" of SURVEY.md 8a-3


def test_backbone_mirrors_have_reference_state_dict_keys():
    """IASSD_Backbone / PAGNet_Backbone: same parameter names and shapes as the reference's classes (so checkpoints
    load, and so scenes.fill_parameters gives both the same weights).  Needs /root/reference: build container only."""
    import copy
    import pytest
    from oracle import ref_harness
    if not ref_harness.available():
        pytest.skip("reference tree not present (GPU box)")
    from oracle import gen_golden
    from spsnet_amd import backbones as BB
    ref_harness.load_reference()
    for filename, cls, mine_cls, cfg in (("IASSD_backbone.py", "IASSD_Backbone", BB.IASSD_Backbone, BB.IASSD_KITTI_CFG),
                                         ("PAGNet_backbone.py", "PAGNet_Backbone", BB.PAGNet_Backbone, BB.SPSNET_KITTI_CFG)):
        ref = gen_golden._load_backbone(filename, cls)(gen_golden._attr(copy.deepcopy(cfg)), num_class=3, input_channels=4)
        mine = mine_cls(copy.deepcopy(cfg), num_class=3, input_channels=4)
        want = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        got = {k: tuple(v.shape) for k, v in mine.state_dict().items()}
        assert got == want
        assert mine.num_point_features == ref.num_point_features
    # PointRCNN's backbone (pointnet2_backbone.py imports the stacked-op package at module level: both harnesses)
    from oracle import ref_harness_stack
    ref_harness_stack.load_reference()
    cfg = BB.POINTRCNN_KITTI_CFG
    ref = gen_golden._load_backbone("pointnet2_backbone.py", "PointNet2MSG")(gen_golden._attr(copy.deepcopy(cfg)), input_channels=4)
    mine = BB.PointNet2MSG(copy.deepcopy(cfg), input_channels=4)
    assert {k: tuple(v.shape) for k, v in mine.state_dict().items()} == {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    assert mine.num_point_features == ref.num_point_features


def test_fill_parameters_is_deterministic_and_name_keyed():
    import torch
    from spsnet_amd import scenes
    a = torch.nn.Sequential(torch.nn.Conv1d(4, 8, 1), torch.nn.BatchNorm1d(8))
    b = torch.nn.Sequential(torch.nn.Conv1d(4, 8, 1), torch.nn.BatchNorm1d(8))
    scenes.fill_parameters(a, 3)
    scenes.fill_parameters(b, 3)
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    assert float(a[1].running_var.min()) >= 0.5
    scenes.fill_parameters(b, 4)
    assert not torch.equal(a[0].weight, b[0].weight)


def test_equal_counts_check_mirrors_the_reference_assert():
    """backbones.equal_counts_check: the reference's `assert xyz_batch_cnt.min() == xyz_batch_cnt.max()`
    (IASSD_backbone.py:109-113) as a deferred verdict."""
    import pytest
    import torch
    from spsnet_amd.backbones import equal_counts_check
    assert equal_counts_check(torch.tensor([0., 0., 1., 1., 2., 2.]), 3)() is True
    assert equal_counts_check(torch.tensor([1., 0., 1., 0.]), 2)() is True          # equal counts, any order
    with pytest.raises(AssertionError):
        equal_counts_check(torch.tensor([0., 0., 0., 1.]), 2)()
    with pytest.raises(AssertionError):
        equal_counts_check(torch.tensor([0., 0., 1., 1.]), 3)()                     # an empty scene


def test_tconv_parts_restatement_matches_library():
    """pointnet2_batch_cuda.tconv_parts restates sps_tconv_parts in Python (one ctypes call less per launch): same values."""
    from spsnet_amd import _lib
    import importlib
    L = _lib.load()
    # (the Python restatement lives in a module that needs no GPU to import its pure functions)
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spsnet_amd", "pointnet2_batch_cuda.py")).read()
    ns = {}
    start = src.index("def tconv_parts(b, l, co):")
    exec(src[start:src.index("\n\n\n", start)], ns)
    for b in (0, 1, 2, 8, 64):
        for l in (0, 64, 192, 2048, 16384, 131072):
            for co in (0, 3, 16, 64, 65, 128, 256):
                assert ns["tconv_parts"](b, l, co) == L.sps_tconv_parts(b, l, co), (b, l, co)


def test_chunk_schedule_env_is_validated():
    """SPS_CHUNK_ENDS / SPS_EARLY_POOL_AT (sa_stack): a chunk list that does not end at 16/16 would never reach the layer's
    last chunk (no repair, no verify.finish, centroids never computed); an early stage only fires where a chunk ends."""
    import pytest
    from spsnet_amd import sa_stack
    assert sa_stack._parse_chunk_ends("4,8,12,14,15,16") == (4, 8, 12, 14, 15, 16)
    for bad in ("4,8", "0,16", "8,4,16", "4,4,16", "4,x,16", "17", ""):
        with pytest.raises(ValueError):
            sa_stack._parse_chunk_ends(bad)
    ends = sa_stack._parse_chunk_ends("3,6,9,12,14,15,16")
    assert sa_stack._parse_early_pool_at("6,9,12", ends) == (6, 9, 12) and sa_stack._parse_early_pool_at("", ends) == ()
    for bad in ("5", "6,16", "9,6", "a"):
        with pytest.raises(ValueError):
            sa_stack._parse_early_pool_at(bad, ends)


def test_pack_scale_declines_what_no_kernel_serves():
    """fused.pack_scale: the per-wave kernels stage at most 256 last-layer biases in LDS -- a scale such as [.., 128, 256, 512]
    must be DECLINED (None -> the layer takes the op-by-op path), not packed and then rejected by the launcher; the
    256/512/1024-wide scales of IA-SSD layer 5 are packed for the point-major fp32 kernel in strict fp32 (the default) and
    for the shared-stream kernel in the split-fp16 arithmetic; without a point-major twin strict fp32 declines them."""
    import torch
    from spsnet_amd import fused, pointnet2_modules as M
    assert fused.PRECISION == "fp32" or "SPS_MLP_PRECISION" in os.environ          # the library default
    mk = lambda widths: M._conv_bn_relu_stack(list(widths), torch.nn.Conv2d, torch.nn.BatchNorm2d).eval()
    old = fused.set_precision("fp32")
    try:
        assert fused.pack_scale(mk([131, 128, 256, 512]), 32) is None
        assert fused.pack_scale(mk([131, 128, 256, 512]), 32, point_major=True) is None
        assert fused.pack_scale(mk([131, 128, 256, 256]), 32) is not None
        wide = mk([259, 256, 512, 1024])
        assert fused.needs_point_major(wide, 32) and not fused.needs_point_major(mk([131, 128, 256, 256]), 32)
        assert fused.pack_scale(wide, 32) is None
        p = fused.pack_scale(wide, 32, point_major=True)
        assert p is not None and p.split == 0 and p.point_major and (p.c1, p.c2, p.c3) == (256, 512, 1024)
        fused.set_precision("fp16x2")
        assert not fused.needs_point_major(wide, 32)
        p = fused.pack_scale(wide, 32)
        assert p is not None and p.split == 2
        assert fused.pack_scale(mk([131, 128, 256, 512]), 32) is None
    finally:
        fused.set_precision(old)
