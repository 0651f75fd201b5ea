"""Generate tests/golden/stackmod_*.npz: the REFERENCE's own pointnet2_stack Python (modules and autograd layer),
run on CPU over the C oracle (oracle/ref_harness_stack.py).  Build-container only; run
    python -m oracle.gen_golden_stack
Fixtures hold inputs, module weights (by state_dict key) and the reference's outputs / input gradients -- data only.
tests/test_stack_modules_cpu.py replays them through the build's modules over the same CPU stand-in,
tests/test_parity_gpu.py through the HIP extension on the GPU box.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_harness_stack as H  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


class Cfg(dict):
    """Minimal EasyDict stand-in (attribute access + .get) for the vector-pool configs."""
    __getattr__ = dict.__getitem__


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def _state(mod, prefix="sd."):
    return {prefix + k: _np(v) for k, v in mod.state_dict().items()}


def _settle(mod, seed):
    """eval mode with non-trivial BatchNorm statistics and affine terms (deterministic)."""
    gen = torch.Generator().manual_seed(seed)
    for m in mod.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            with torch.no_grad():
                m.running_mean.copy_(torch.randn(m.num_features, generator=gen) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
                m.weight.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(m.num_features, generator=gen) * 0.1)
    return mod.eval()


def ragged_scene(seed, counts=(300, 200), centres=(40, 24), channels=6):
    """Two scenes of different size, a few exact duplicates, centres = stacked FPS picks plus one far-away centre per
    scene (an empty ball)."""
    rng = np.random.default_rng(seed)
    xyz = np.concatenate([rng.uniform(-2, 2, (n, 3)) for n in counts]).astype(np.float32)
    xyz[10:20] = xyz[30:40]
    feats = rng.normal(size=(sum(counts), channels)).astype(np.float32)
    cnt = np.asarray(counts, np.int32)
    return xyz, feats, cnt, np.asarray(centres, np.int32)


def _centres(U, xyz, cnt, per_scene):
    t = torch.from_numpy
    picks = U.stack_farthest_point_sample(t(xyz), t(cnt), t(per_scene - 1))
    picks = _np(picks).astype(np.int64)
    out, start = [], 0
    for b, k in enumerate(per_scene - 1):
        out.append(xyz[picks[start:start + k]])
        out.append(np.asarray([[50.0 + b, 50.0, 50.0]], np.float32))     # nothing within any radius: empty ball
        start += k
    return np.concatenate(out).astype(np.float32), picks.astype(np.int32)


def sa_and_fp(U, M):
    t = torch.from_numpy
    xyz, feats, cnt, per_scene = ragged_scene(1)
    new_xyz, picks = _centres(U, xyz, cnt, per_scene)
    torch.manual_seed(3)
    sa = _settle(M.StackSAModuleMSG(radii=[0.5, 1.0], nsamples=[8, 16], mlps=[[6, 16, 16], [6, 16, 24]], use_xyz=True,
                                    pool_method='max_pool'), 5)
    f = t(feats).requires_grad_(True)
    _, out = sa(t(xyz), t(cnt), t(new_xyz), t(per_scene), f)
    probe = torch.from_numpy(np.random.default_rng(7).normal(size=tuple(out.shape)).astype(np.float32))
    (out * probe).sum().backward()
    idx, empty = U.ball_query(0.5, 8, t(xyz), t(cnt), t(new_xyz), t(per_scene))
    grouped, _ = U.QueryAndGroup(0.5, 8, use_xyz=True)(t(xyz), t(cnt), t(new_xyz), t(per_scene), t(feats))
    d = dict(xyz=xyz, feats=feats, cnt=cnt, new_xyz=new_xyz, new_cnt=per_scene, fps_picks=picks, out=_np(out), probe=_np(probe),
             grad_feats=_np(f.grad), bq_idx=_np(idx), bq_empty=_np(empty), grouped=_np(grouped))
    d.update(_state(sa))
    np.savez_compressed(os.path.join(OUT, "stackmod_sa.npz"), **d)

    torch.manual_seed(4)
    fp = _settle(M.StackPointnetFPModule(mlp=[6 + 5, 16, 12]), 6)
    known_feats = np.random.default_rng(8).normal(size=(new_xyz.shape[0], 5)).astype(np.float32)
    kf = t(known_feats).requires_grad_(True)
    out = fp(t(xyz), t(cnt), t(new_xyz), t(per_scene), unknown_feats=t(feats), known_feats=kf)
    probe = torch.from_numpy(np.random.default_rng(9).normal(size=tuple(out.shape)).astype(np.float32))
    (out * probe).sum().backward()
    d = dict(xyz=xyz, feats=feats, cnt=cnt, known=new_xyz, known_cnt=per_scene, known_feats=known_feats, out=_np(out),
             probe=_np(probe), grad_known_feats=_np(kf.grad))
    d.update(_state(fp))
    np.savez_compressed(os.path.join(OUT, "stackmod_fp.npz"), **d)


def vector_pool(U, M):
    t = torch.from_numpy
    xyz, feats, cnt, per_scene = ragged_scene(2, channels=8)
    new_xyz, _ = _centres(U, xyz, cnt, per_scene)
    d = dict(xyz=xyz, feats=feats, cnt=cnt, new_xyz=new_xyz, new_cnt=per_scene)
    for tag, kind in (("interp", "local_interpolation"), ("avg", "voxel_avg_pool"), ("first", "voxel_random_choice")):
        torch.manual_seed(10)
        mod = _settle(M.VectorPoolAggregationModule(
            input_channels=8, num_local_voxel=(2, 2, 2), local_aggregation_type=kind, num_reduced_channels=4,
            num_channels_of_local_aggregation=8, post_mlps=(16,), max_neighbor_distance=1.2, neighbor_nsample=-1,
            neighbor_type=0, neighbor_distance_multiplier=2.0), 11)
        mod.num_mean_points_per_grid = 2          # small buffers: the overflow-and-retry protocol runs
        if mod.local_interpolate_module is not None:
            mod.local_interpolate_module.num_avg_length_of_neighbor_idxs = 3
        f = t(feats).requires_grad_(True)
        _, out = mod(xyz=t(xyz), xyz_batch_cnt=t(cnt), new_xyz=t(new_xyz), new_xyz_batch_cnt=t(per_scene), features=f)
        probe = torch.from_numpy(np.random.default_rng(12).normal(size=tuple(out.shape)).astype(np.float32))
        (out * probe).sum().backward()
        d[f"{tag}_out"], d[f"{tag}_probe"], d[f"{tag}_grad_feats"] = _np(out), _np(probe), _np(f.grad)
        d.update(_state(mod, prefix=f"{tag}_sd."))
    cfg = Cfg(NUM_GROUPS=2, LOCAL_AGGREGATION_TYPE='voxel_avg_pool', NUM_REDUCED_CHANNELS=4,
              NUM_CHANNELS_OF_LOCAL_AGGREGATION=8, MSG_POST_MLPS=[24],
              GROUP_CFG_0=Cfg(NUM_LOCAL_VOXEL=[2, 2, 2], MAX_NEIGHBOR_DISTANCE=0.8, NEIGHBOR_NSAMPLE=-1, POST_MLPS=[16, 16]),
              GROUP_CFG_1=Cfg(NUM_LOCAL_VOXEL=[3, 3, 3], MAX_NEIGHBOR_DISTANCE=1.6, NEIGHBOR_NSAMPLE=-1, POST_MLPS=[16]))
    torch.manual_seed(13)
    msg = _settle(M.VectorPoolAggregationModuleMSG(input_channels=8, config=cfg), 14)
    _, out = msg(xyz=t(xyz), xyz_batch_cnt=t(cnt), new_xyz=t(new_xyz), new_xyz_batch_cnt=t(per_scene), features=t(feats))
    d["msg_out"] = _np(out)
    d.update(_state(msg, prefix="msg_sd."))
    np.savez_compressed(os.path.join(OUT, "stackmod_vp.npz"), **d)


def voxel_sa(VQ, VP):
    """NeighborVoxelSAModuleMSG over a dense voxel -> point table; equally many queries per scene (the reference's
    VoxelQueryAndGrouping assumes it, voxel_query_utils.py:84-90)."""
    t = torch.from_numpy
    rng = np.random.default_rng(20)
    counts = np.asarray([260, 180], np.int32)
    grid, size = (8, 10, 12), 0.5                                   # (Z, Y, X) voxels of edge `size`
    xyz_parts, table = [], -np.ones((2,) + grid, np.int32)
    base = 0
    for b, n in enumerate(counts):
        pts = rng.uniform(0, 1, (n, 3)).astype(np.float32) * np.asarray([grid[2], grid[1], grid[0]], np.float32) * size
        xyz_parts.append(pts)
        vox = np.floor(pts / size).astype(np.int64)                    # (x, y, z) voxel of every point
        for k in range(n):                                             # one point per voxel: the last writer
            table[b, vox[k, 2], vox[k, 1], vox[k, 0]] = base + k
        base += n
    xyz = np.concatenate(xyz_parts)
    feats = rng.normal(size=(xyz.shape[0], 6)).astype(np.float32)
    per = 20
    new_xyz, new_coords = [], []
    for b in range(2):
        c = rng.uniform(0.5, 3.5, (per, 3)).astype(np.float32)
        new_xyz.append(c)
        v = np.floor(c / size).astype(np.int32)
        new_coords.append(np.concatenate([np.full((per, 1), b, np.int32), v], axis=1))   # [b, x, y, z] as callers pass it
    new_xyz, new_coords = np.concatenate(new_xyz), np.concatenate(new_coords)
    new_cnt = np.asarray([per, per], np.int32)
    torch.manual_seed(21)
    mod = _settle(VP.NeighborVoxelSAModuleMSG(query_ranges=[[2, 2, 2], [3, 3, 3]], radii=[0.8, 1.4], nsamples=[6, 12],
                                              mlps=[[6, 12, 16], [6, 12, 20]]), 22)
    f = t(feats).requires_grad_(True)
    out = mod(t(xyz), t(counts), t(new_xyz), t(new_cnt), t(new_coords), f, t(table))
    probe = torch.from_numpy(np.random.default_rng(23).normal(size=tuple(out.shape)).astype(np.float32))
    (out * probe).sum().backward()
    d = dict(xyz=xyz, feats=feats, cnt=counts, new_xyz=new_xyz, new_cnt=new_cnt, new_coords=new_coords, table=table,
             out=_np(out), probe=_np(probe), grad_feats=_np(f.grad))
    d.update(_state(mod))
    np.savez_compressed(os.path.join(OUT, "stackmod_voxel.npz"), **d)


def main():
    U, M, VQ, VP = H.load_reference()
    os.makedirs(OUT, exist_ok=True)
    sa_and_fp(U, M)
    vector_pool(U, M)
    voxel_sa(VQ, VP)
    for f in sorted(os.listdir(OUT)):
        if f.startswith("stackmod_"):
            print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")


if __name__ == "__main__":
    main()
