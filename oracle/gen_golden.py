"""Generate tests/golden/*.npz by running the REFERENCE's Python layers over the C oracle.

Build-container only (needs /root/reference; see oracle/ref_harness.py).  Run:
    python -m oracle.gen_golden
The fixtures hold inputs, module weights and the reference's outputs -- data only; no reference
source text travels.  tests/test_golden_cpu.py re-checks the C oracle against them without the
reference, tests/test_parity_gpu.py checks the HIP path against them on the GPU box.
"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_harness  # noqa: E402
from spsnet_amd import scenes  # noqa: E402  (pure numpy, no HIP dependency)

OUT = os.path.join(ROOT, "tests", "golden")


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else t


def _state(mod, prefix="sd."):
    return {prefix + k: _np(v) for k, v in mod.state_dict().items()}


def _randomize_bn(mod, seed):
    gen = torch.Generator().manual_seed(seed)
    for m in mod.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            with torch.no_grad():
                m.running_mean.copy_(torch.randn(m.num_features, generator=gen) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
                m.weight.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(m.num_features, generator=gen) * 0.1)


def ops_small(U):
    rng = np.random.default_rng(1234)
    B, N = 2, 256
    xyz = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    xyz[:, 40:60] = xyz[:, 10:30]            # exact duplicates -> FPS / three_nn ties
    xyz[1, 200] = xyz[1, 0]
    feats = rng.normal(size=(B, 5, N)).astype(np.float32)
    t = torch.from_numpy
    fps_idx = U.furthest_point_sample(t(xyz), 64)
    new_xyz = U.gather_operation(t(xyz).transpose(1, 2).contiguous(), fps_idx).transpose(1, 2).contiguous()
    bq = U.ball_query(0.3, 8, t(xyz), new_xyz)
    bqd = U.ball_query_dilated(0.4, 0.1, 8, t(xyz), new_xyz)
    bqd0 = U.ball_query_dilated(0.3, 0.0, 8, t(xyz), new_xyz)
    grouped = U.grouping_operation(t(feats), bq)
    qg = U.QueryAndGroup(0.3, 8, use_xyz=True)(t(xyz), new_xyz, t(feats))
    qg_nofeat = U.QueryAndGroup(0.3, 8, use_xyz=True)(t(xyz), new_xyz, None)
    unknown = rng.uniform(-1, 1, (B, 100, 3)).astype(np.float32)
    unknown[:, :10] = _np(new_xyz)[:, :10]   # zero distances
    dist, nn_idx = U.three_nn(t(unknown), new_xyz)
    w = 1.0 / (dist + 1e-8)
    w = (w / w.sum(2, keepdim=True)).contiguous()
    known_feats = t(rng.normal(size=(B, 7, 64)).astype(np.float32))
    interp = U.three_interpolate(known_feats, nn_idx, w)
    dmat = rng.uniform(0, 4, (B, 64, 64)).astype(np.float32)
    dmat[:, :, 5] = dmat[:, :, 9]            # tied columns
    fps_d = U.furthest_point_sample_with_dist(t(dmat), 16)
    # gradients (fp32 sums in ascending-index order in the oracle)
    f2 = t(feats).clone().requires_grad_(True)
    go = torch.from_numpy(rng.normal(size=tuple(grouped.shape)).astype(np.float32))
    U.grouping_operation(f2, bq).backward(go)
    f3 = t(feats).clone().requires_grad_(True)
    gg = torch.from_numpy(rng.normal(size=(B, 5, 64)).astype(np.float32))
    U.gather_operation(f3, fps_idx).backward(gg)
    k2 = known_feats.clone().requires_grad_(True)
    gi = torch.from_numpy(rng.normal(size=tuple(interp.shape)).astype(np.float32))
    U.three_interpolate(k2, nn_idx, w).backward(gi)
    np.savez_compressed(
        os.path.join(OUT, "ops_small.npz"), xyz=xyz, feats=feats, fps_idx=_np(fps_idx), new_xyz=_np(new_xyz),
        bq=_np(bq), bqd=_np(bqd), bqd0=_np(bqd0), grouped=_np(grouped), qg=_np(qg), qg_nofeat=_np(qg_nofeat),
        unknown=unknown, nn_dist=_np(dist), nn_idx=_np(nn_idx), interp_w=_np(w), known_feats=_np(known_feats),
        interp=_np(interp), dmat=dmat, fps_d=_np(fps_d), group_go=_np(go), group_grad=_np(f2.grad),
        gather_go=_np(gg), gather_grad=_np(f3.grad), interp_go=_np(gi), interp_grad=_np(k2.grad))


def _sa_case(M, name, xyz, feats, ctor_kw, fwd_kw=None, seed=0, cls_in=None):
    torch.manual_seed(seed)
    # deep copy: the constructor adds 3 to mlps[i][0] in place (pointnet2_modules.py:199-201)
    mod = M.PointnetSAModuleMSG_WithSampling(**copy.deepcopy(ctor_kw)).eval()
    _randomize_bn(mod, seed + 1)
    fwd_kw = fwd_kw or {}
    with torch.no_grad():
        out = mod(torch.from_numpy(xyz), torch.from_numpy(feats),
                  None if cls_in is None else torch.from_numpy(cls_in), **fwd_kw)
    new_xyz, new_feat, cls, idx, stds = out
    d = dict(xyz=xyz, feats=feats, new_xyz=_np(new_xyz), new_features=_np(new_feat), idx=_np(idx))
    if cls is not None:
        d["cls"] = _np(cls)
    if cls_in is not None:
        d["cls_in"] = cls_in
    if stds is not None:
        d["stds_out"] = _np(stds)
    for k, v in fwd_kw.items():
        d["kw_" + k] = _np(v)
    d.update(_state(mod))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)


def config1(M):
    """BASELINE config 1: one synthetic KITTI scene, 4 096 -> 512, one scale r=0.8 ns=16."""
    xyz, feats = scenes.make_batch("kitti-lidar-v1", 1, 4096, seed0=100)
    _sa_case(M, "config1_sa", xyz, feats, dict(
        npoint_list=[512], sample_range_list=[-1], sample_type_list=['D-FPS'], radii=[0.8], nsamples=[16],
        mlps=[[1, 16, 16, 32]], use_xyz=True, dilated_group=False, aggregation_mlp=[32], confidence_mlp=[16],
        num_class=3))


def samplers(M):
    rng = np.random.default_rng(7)
    xyz, feats = scenes.make_batch("kitti-lidar-v1", 2, 1024, seed0=200)
    feats = rng.normal(size=(2, 6, 1024)).astype(np.float32)
    cls_in = rng.normal(size=(2, 1024, 3)).astype(np.float32)
    common = dict(sample_range_list=[-1], radii=[1.6, 4.8], nsamples=[8, 16], mlps=[[6, 8, 16], [6, 8, 24]],
                  use_xyz=True, dilated_group=False, aggregation_mlp=[32], confidence_mlp=[16], num_class=3)
    _sa_case(M, "sampler_ctr", xyz, feats, dict(npoint_list=[256], sample_type_list=['ctr_aware'], **common),
             cls_in=cls_in, seed=3)
    stds = torch.from_numpy(rng.uniform(0, 40, (2, 1024)).astype(np.float32))
    _sa_case(M, "sampler_sss", xyz, feats, dict(npoint_list=[256], sample_type_list=['sss_aware'], **common),
             fwd_kw=dict(stds=stds), cls_in=cls_in, seed=4)
    # no grouping (IA-SSD layer 3): feature gather only
    _sa_case(M, "sampler_nogroup", xyz, feats, dict(
        npoint_list=[128], sample_range_list=[-1], sample_type_list=['ctr_aware'], radii=[], nsamples=[], mlps=[],
        use_xyz=True, dilated_group=False, aggregation_mlp=[32], confidence_mlp=None, num_class=3),
        cls_in=cls_in, seed=5)
    # D-FPS carrying stds along, dilated grouping
    _sa_case(M, "sampler_dfps_stds_dilated", xyz, feats, dict(
        npoint_list=[128], sample_range_list=[-1], sample_type_list=['D-FPS'], radii=[0.8, 1.6], nsamples=[8, 8],
        mlps=[[6, 8, 8], [6, 8, 8]], use_xyz=True, dilated_group=True, aggregation_mlp=[16], confidence_mlp=None,
        num_class=3), fwd_kw=dict(stds=stds), seed=6)
    # feature-FPS and fusion sampling (F-FPS / FS), small N because of the N x N matrix
    xyz_s, _ = scenes.make_batch("uniform-v1", 2, 256, seed0=300)
    feats_s = rng.normal(size=(2, 4, 256)).astype(np.float32)
    small = dict(sample_range_list=[-1], radii=[8.0], nsamples=[8], mlps=[[4, 8, 8]], use_xyz=True,
                 dilated_group=False, aggregation_mlp=None, confidence_mlp=None, num_class=3)
    _sa_case(M, "sampler_ffps", xyz_s, feats_s, dict(npoint_list=[64], sample_type_list=['F-FPS'], **small), seed=7)
    _sa_case(M, "sampler_fs", xyz_s, feats_s, dict(npoint_list=[32], sample_type_list=['FS'], **small), seed=8)


def samplers_partitioned(M):
    """The remaining branches of the sampling dispatcher (pointnet2_modules.py:314-419): S-FPS on both sides of its
    hard-coded 3500-distinct-points fallback, ds-FPS and ry-FPS."""
    rng = np.random.default_rng(11)
    small = dict(sample_range_list=[-1], radii=[1.6], nsamples=[8], mlps=[[2, 8, 8]], use_xyz=True, dilated_group=False,
                 aggregation_mlp=None, confidence_mlp=None, num_class=3)
    xyz, _ = scenes.make_batch("kitti-lidar-v1", 2, 1024, seed0=400)
    feats = rng.normal(size=(2, 2, 1024)).astype(np.float32)
    stds = torch.from_numpy(rng.uniform(0, 40, (2, 1024)).astype(np.float32))
    # 256 picks: fewer than 3500 distinct -> the picks fall back to plain FPS, `stds` keeps the stable picks' values
    _sa_case(M, "sampler_sfps_fallback", xyz, feats, dict(npoint_list=[256], sample_type_list=['S-FPS'], ss_radii=[0.8],
             ss_nsamples=[8], **small), fwd_kw=dict(stds=stds), seed=12)
    _sa_case(M, "sampler_dsfps", xyz, feats, dict(npoint_list=[256], sample_type_list=['ds-FPS'], **small), seed=13)
    _sa_case(M, "sampler_ryfps", xyz, feats, dict(npoint_list=[256], sample_type_list=['ry-FPS'], **small), seed=14)
    # Rand (pointnet2_modules.py:370-371): one torch.randperm prefix shared by the batch.  The fixture pins what happens WITH the
    # permutation (prefix, int32, repeated per scene, then gather / group / MLP); the test feeds the same permutation back in.
    _sa_case(M, "sampler_rand", xyz, feats, dict(npoint_list=[256], sample_type_list=['Rand'], **small), seed=16)
    # 4096 of 8192 picks with a tight stability ball: scene 0 keeps >= 3500 distinct picks -> the stable picks are used
    xyz_l, _ = scenes.make_batch("kitti-lidar-v1", 2, 8192, seed0=401)
    feats_l = rng.normal(size=(2, 2, 8192)).astype(np.float32)
    stds_l = torch.from_numpy(rng.uniform(0, 40, (2, 8192)).astype(np.float32))
    _sa_case(M, "sampler_sfps", xyz_l, feats_l, dict(npoint_list=[4096], sample_type_list=['S-FPS'], ss_radii=[0.15],
             ss_nsamples=[4], **small), fwd_kw=dict(stds=stds_l), seed=15)
    g = np.load(os.path.join(OUT, "sampler_sfps.npz"))
    distinct = np.unique(g["idx"][0]).size
    assert distinct >= 3500, f"sampler_sfps fell back to plain FPS ({distinct} distinct picks): tighten ss_radii"
    print("sampler_sfps: distinct picks in scene 0:", distinct)


def stack3(M):
    """Three chained SA layers (D-FPS, D-FPS, ctr_aware) at reduced size/width: the IA-SSD pattern."""
    from spsnet_amd import sa_stack
    cfg = sa_stack.scaled_config(npoints=[512, 128, 64])
    cfg['mlps'] = [[[8, 8, 16], [8, 8, 16]], [[16, 16, 32], [16, 24, 32]], [[32, 32, 64], [32, 64, 64]]]
    cfg['aggregation_mlps'] = [[16], [32], [64]]
    cfg['confidence_mlps'] = [[], [32], [64]]
    xyz, feats = scenes.make_batch("kitti-lidar-v1", 2, 2048, seed0=400, dup_fraction=0.02)
    layers = sa_stack.build_sa_layers(M, cfg, seed=11)
    with torch.no_grad():
        outs = sa_stack.run_sa_layers(layers, torch.from_numpy(xyz), torch.from_numpy(feats))
    d = dict(xyz=xyz, feats=feats)
    for k, (nx, nf, cls, idx) in enumerate(outs):
        d[f"l{k}_new_xyz"], d[f"l{k}_new_features"], d[f"l{k}_idx"] = _np(nx), _np(nf), _np(idx)
        if cls is not None:
            d[f"l{k}_cls"] = _np(cls)
    d.update(_state(layers))
    np.savez_compressed(os.path.join(OUT, "stack3_small.npz"), **d)


def generator_layer(M):
    """stability_generate/model.py:84-95 shape: PointnetSampling with npoint >= N (identity sampling)."""
    xyz, feats = scenes.make_batch("kitti-lidar-v1", 2, 512, seed0=500)
    torch.manual_seed(21)
    mod = M.PointnetSampling(npoint_list=[512], sample_range_list=[-1], sample_type_list=['D-FPS'],
                             radii=[0.2, 0.8], nsamples=[16, 32], mlps=[[1, 16, 16, 32], [1, 32, 32, 64]],
                             use_xyz=True, dilated_group=False, aggregation_mlp=[64]).eval()
    _randomize_bn(mod, 22)
    with torch.no_grad():
        nx, nf, idx = mod(torch.from_numpy(xyz), torch.from_numpy(feats))
    d = dict(xyz=xyz, feats=feats, new_xyz=_np(nx), new_features=_np(nf), idx=_np(idx))
    d.update(_state(mod))
    np.savez_compressed(os.path.join(OUT, "generator_layer.npz"), **d)


def fp_module(M):
    rng = np.random.default_rng(31)
    unknown = rng.uniform(-2, 2, (2, 200, 3)).astype(np.float32)
    known = rng.uniform(-2, 2, (2, 50, 3)).astype(np.float32)
    uf = rng.normal(size=(2, 4, 200)).astype(np.float32)
    kf = rng.normal(size=(2, 6, 50)).astype(np.float32)
    torch.manual_seed(32)
    mod = M.PointnetFPModule(mlp=[10, 16, 8]).eval()
    _randomize_bn(mod, 33)
    with torch.no_grad():
        out = mod(*(torch.from_numpy(a) for a in (unknown, known, uf, kf)))
    d = dict(unknown=unknown, known=known, uf=uf, kf=kf, out=_np(out))
    d.update(_state(mod))
    np.savez_compressed(os.path.join(OUT, "fp_module.npz"), **d)


def surface_features():
    """surface_feature.py FeatureExtraction, static and dynamic graph (reference Python over the oracle)."""
    import importlib
    SF = importlib.import_module("pcdet.ops.pointnet2.pointnet2_batch.surface_feature")
    xyz, _ = scenes.make_batch("kitti-lidar-v1", 2, 384, seed0=600)
    xyz = (xyz - xyz.mean(1, keepdims=True)).astype(np.float32) * 0.2   # dense enough for r = 0.8 balls
    d = dict(xyz=xyz)
    for name, dyn in (("static", False), ("dynamic", True)):
        torch.manual_seed(41)
        net = SF.FeatureExtraction(dynamic_graph=dyn).eval()
        with torch.no_grad():
            out = net(torch.from_numpy(xyz))
        d["out_" + name] = _np(out)
        d.update(_state(net, prefix=f"sd_{name}."))
    np.savez_compressed(os.path.join(OUT, "surface_feature.npz"), **d)


class _AttrDict(dict):
    """What the reference's backbones need of an EasyDict: attribute access over a dict."""
    __getattr__ = dict.__getitem__


def _attr(obj):
    if isinstance(obj, dict):
        return _AttrDict({k: _attr(v) for k, v in obj.items()})
    return obj


def _load_backbone(filename, cls):
    """Execute ONE reference backbone file (pcdet/models/backbones_3d/<filename>) under its package name without
    running pcdet/models/__init__.py, which imports spconv-based code that is not installed here."""
    import importlib.util
    import types
    for name in ("pcdet.models", "pcdet.models.backbones_3d"):
        if name not in sys.modules:
            pkg = types.ModuleType(name)
            pkg.__path__ = []
            sys.modules[name] = pkg
    modname = "pcdet.models.backbones_3d." + filename[:-3]
    spec = importlib.util.spec_from_file_location(
        modname, os.path.join(ref_harness.REFERENCE_ROOT, "pcdet", "models", "backbones_3d", filename))
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = "pcdet.models.backbones_3d"
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return getattr(mod, cls)


BACKBONE_NPOINTS = [1024, 256, 128, 64, -1, 64]
BACKBONE_SEED = 77


def backbones():
    """IASSD_Backbone / PAGNet_Backbone end to end (IASSD_backbone.py:93-178, PAGNet_backbone.py:102-197) at the shipped
    widths on 2 x 4096 points.  The weights are not stored: scenes.fill_parameters(seed) regenerates them by key name."""
    from spsnet_amd import backbones as B
    for tag, filename, cls, base in (("iassd", "IASSD_backbone.py", "IASSD_Backbone", B.IASSD_KITTI_CFG),
                                     ("pagnet", "PAGNet_backbone.py", "PAGNet_Backbone", B.SPSNET_KITTI_CFG)):
        cfg = B.scaled_cfg(base, BACKBONE_NPOINTS)
        net = _load_backbone(filename, cls)(_attr(copy.deepcopy(cfg)), num_class=3, input_channels=4).eval()
        scenes.fill_parameters(net, BACKBONE_SEED)
        xyz, feats = scenes.make_batch("kitti-lidar-v1", 2, 4096, seed0=700, dup_fraction=0.01)
        bidx = np.repeat(np.arange(2, dtype=np.float32), 4096)[:, None]
        points = np.concatenate([bidx, xyz.reshape(-1, 3), feats.transpose(0, 2, 1).reshape(-1, 1)], 1).astype(np.float32)
        batch = dict(batch_size=2, points=torch.from_numpy(points))
        d = dict(points=points, npoints=np.asarray(BACKBONE_NPOINTS), seed=np.asarray(BACKBONE_SEED))
        if tag == "pagnet":
            stds = np.random.default_rng(9).uniform(0, 40, (2, 4096)).astype(np.float32)
            batch['stds'] = torch.from_numpy(stds)
            d['stds'] = stds
        with torch.no_grad():
            out = net(batch)
        for key in ("ctr_offsets", "centers", "centers_origin", "centers_features", "ctr_batch_idx"):
            d[key] = _np(out[key])
        for k, t in enumerate(out["encoder_xyz"]):
            d[f"encoder_xyz_{k}"] = _np(t)
        for k, t in enumerate(out["encoder_features"]):
            if t is not None and k > 0:
                d[f"encoder_features_{k}"] = _np(t)
        for k, t in enumerate(out["sa_ins_preds"]):
            if isinstance(t, torch.Tensor):
                d[f"sa_ins_preds_{k}"] = _np(t)
        d["n_encoder_coords"] = np.asarray(len(out["encoder_coords"]))
        np.savez_compressed(os.path.join(OUT, f"backbone_{tag}.npz"), **d)


POINTNET2MSG_NPOINTS = [1024, 256, 64, 16]
POINTNET2MSG_SEED = 31


def pointnet2msg():
    """PointNet2MSG end to end (pcdet/models/backbones_3d/pointnet2_backbone.py:9-100, PointRCNN's backbone: four MSG SA
    layers + four feature-propagation layers) at the shipped widths on 2 x 4096 points, centroid counts scaled down.
    The reference file also imports the stacked-op package at module level, hence both harnesses."""
    from oracle import ref_harness_stack
    from spsnet_amd import backbones as B
    ref_harness_stack.load_reference()
    cfg = copy.deepcopy(B.POINTRCNN_KITTI_CFG)
    cfg['SA_CONFIG']['NPOINTS'] = list(POINTNET2MSG_NPOINTS)
    net = _load_backbone("pointnet2_backbone.py", "PointNet2MSG")(_attr(copy.deepcopy(cfg)), input_channels=4).eval()
    scenes.fill_parameters(net, POINTNET2MSG_SEED)
    xyz, feats = scenes.make_batch("kitti-lidar-v1", 2, 4096, seed0=900, dup_fraction=0.01)
    bidx = np.repeat(np.arange(2, dtype=np.float32), 4096)[:, None]
    points = np.concatenate([bidx, xyz.reshape(-1, 3), feats.transpose(0, 2, 1).reshape(-1, 1)], 1).astype(np.float32)
    with torch.no_grad():
        out = net(dict(batch_size=2, points=torch.from_numpy(points)))
    np.savez_compressed(os.path.join(OUT, "backbone_pointnet2msg.npz"), points=points, npoints=np.asarray(POINTNET2MSG_NPOINTS),
                        seed=np.asarray(POINTNET2MSG_SEED), point_features=_np(out["point_features"]),
                        point_coords=_np(out["point_coords"]), num_point_features=np.asarray(net.num_point_features),
                        n_state=np.asarray(len(net.state_dict())))


def main():
    os.makedirs(OUT, exist_ok=True)
    U, M = ref_harness.load_reference()
    torch.set_num_threads(4)
    if len(sys.argv) > 1 and sys.argv[1] == "samplers_partitioned":   # (added later: regenerate only these)
        samplers_partitioned(M)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "pointnet2msg":
        pointnet2msg()
        return
    ops_small(U)
    config1(M)
    samplers(M)
    samplers_partitioned(M)
    stack3(M)
    generator_layer(M)
    fp_module(M)
    surface_features()
    backbones()
    pointnet2msg()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
