"""Replay of tests/golden/stackmod_*.npz (made by the REFERENCE's pointnet2_stack Python over the C oracle,
oracle/gen_golden_stack.py) through the build's own stack modules, on any device.  Shared by the CPU host-logic test
(extension patched with the oracle stand-in) and the GPU parity test (HIP extension through the C ABI)."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Cfg(dict):
    __getattr__ = dict.__getitem__


def load(name):
    return np.load(os.path.join(GOLD, name))


def _load_state(mod, g, prefix="sd."):
    sd = {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}
    mod.load_state_dict(sd, strict=True)      # also proves the parameter names are the reference's
    return mod.eval()


def _close(got, want, tol=1e-5):
    got = got.detach().cpu().numpy()
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max())
    assert got.shape == want.shape and err <= tol * scale, f"off by {err} (scale {scale})"


def replay_sa_fp(dev):
    from spsnet_amd.pointnet2_stack import pointnet2_modules as M, pointnet2_utils as U
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    g = load("stackmod_sa.npz")
    xyz, cnt, new_xyz, new_cnt = t(g["xyz"]), t(g["cnt"]), t(g["new_xyz"]), t(g["new_cnt"])
    picks = U.stack_farthest_point_sample(xyz, cnt, t(g["new_cnt"] - 1))
    np.testing.assert_array_equal(picks.cpu().numpy(), g["fps_picks"])
    idx, empty = U.ball_query(0.5, 8, xyz, cnt, new_xyz, new_cnt)
    np.testing.assert_array_equal(idx.cpu().numpy(), g["bq_idx"])
    np.testing.assert_array_equal(empty.cpu().numpy(), g["bq_empty"])
    assert bool(empty.any())                                      # the fixture holds empty balls
    grouped, _ = U.QueryAndGroup(0.5, 8, use_xyz=True)(xyz, cnt, new_xyz, new_cnt, t(g["feats"]))
    np.testing.assert_array_equal(grouped.cpu().numpy(), g["grouped"])
    sa = _load_state(M.StackSAModuleMSG(radii=[0.5, 1.0], nsamples=[8, 16], mlps=[[6, 16, 16], [6, 16, 24]], use_xyz=True,
                                        pool_method='max_pool'), g).to(dev)
    f = t(g["feats"]).requires_grad_(True)
    _, out = sa(xyz, cnt, new_xyz, new_cnt, f)
    _close(out, g["out"])
    (out * t(g["probe"])).sum().backward()
    _close(f.grad, g["grad_feats"])

    g = load("stackmod_fp.npz")
    fp = _load_state(M.StackPointnetFPModule(mlp=[11, 16, 12]), g).to(dev)
    kf = t(g["known_feats"]).requires_grad_(True)
    out = fp(t(g["xyz"]), t(g["cnt"]), t(g["known"]), t(g["known_cnt"]), unknown_feats=t(g["feats"]), known_feats=kf)
    _close(out, g["out"])
    (out * t(g["probe"])).sum().backward()
    _close(kf.grad, g["grad_known_feats"])


def replay_vector_pool(dev):
    from spsnet_amd.pointnet2_stack import pointnet2_modules as M
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    g = load("stackmod_vp.npz")
    args = dict(xyz=t(g["xyz"]), xyz_batch_cnt=t(g["cnt"]), new_xyz=t(g["new_xyz"]), new_xyz_batch_cnt=t(g["new_cnt"]))
    for tag, kind in (("interp", "local_interpolation"), ("avg", "voxel_avg_pool"), ("first", "voxel_random_choice")):
        mod = _load_state(M.VectorPoolAggregationModule(
            input_channels=8, num_local_voxel=(2, 2, 2), local_aggregation_type=kind, num_reduced_channels=4,
            num_channels_of_local_aggregation=8, post_mlps=(16,), max_neighbor_distance=1.2, neighbor_nsample=-1,
            neighbor_type=0, neighbor_distance_multiplier=2.0), g, prefix=f"{tag}_sd.").to(dev)
        mod.num_mean_points_per_grid = 2          # small buffers: the overflow-and-retry protocol runs
        if mod.local_interpolate_module is not None:
            mod.local_interpolate_module.num_avg_length_of_neighbor_idxs = 3
        f = t(g["feats"]).requires_grad_(True)
        _, out = mod(features=f, **args)
        _close(out, g[f"{tag}_out"])
        (out * t(g[f"{tag}_probe"])).sum().backward()
        _close(f.grad, g[f"{tag}_grad_feats"])
        assert mod.num_mean_points_per_grid > 2 or kind == "local_interpolation"   # the buffer did grow
    cfg = Cfg(NUM_GROUPS=2, LOCAL_AGGREGATION_TYPE='voxel_avg_pool', NUM_REDUCED_CHANNELS=4,
              NUM_CHANNELS_OF_LOCAL_AGGREGATION=8, MSG_POST_MLPS=[24],
              GROUP_CFG_0=Cfg(NUM_LOCAL_VOXEL=[2, 2, 2], MAX_NEIGHBOR_DISTANCE=0.8, NEIGHBOR_NSAMPLE=-1, POST_MLPS=[16, 16]),
              GROUP_CFG_1=Cfg(NUM_LOCAL_VOXEL=[3, 3, 3], MAX_NEIGHBOR_DISTANCE=1.6, NEIGHBOR_NSAMPLE=-1, POST_MLPS=[16]))
    msg, width = M.build_local_aggregation_module(8, Cfg(NAME='VectorPoolAggregationModuleMSG', **cfg))
    assert width == 24
    msg = _load_state(msg, g, prefix="msg_sd.").to(dev)
    _, out = msg(features=t(g["feats"]), **args)
    _close(out, g["msg_out"])


def replay_voxel_sa(dev):
    from spsnet_amd.pointnet2_stack import voxel_pool_modules as VP
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    g = load("stackmod_voxel.npz")
    mod = _load_state(VP.NeighborVoxelSAModuleMSG(query_ranges=[[2, 2, 2], [3, 3, 3]], radii=[0.8, 1.4], nsamples=[6, 12],
                                                  mlps=[[6, 12, 16], [6, 12, 20]]), g).to(dev)
    f = t(g["feats"]).requires_grad_(True)
    out = mod(t(g["xyz"]), t(g["cnt"]), t(g["new_xyz"]), t(g["new_cnt"]), t(g["new_coords"]), f, t(g["table"]))
    _close(out, g["out"])
    (out * t(g["probe"])).sum().backward()
    _close(f.grad, g["grad_feats"])
