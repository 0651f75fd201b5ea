"""The C oracle against the committed golden vectors (CPU only, no reference tree needed).

tests/golden/*.npz were produced by oracle/gen_golden.py, which drives the REFERENCE's Python
layers (pointnet2_utils.py / pointnet2_modules.py) on CPU.  These tests re-derive the
index-producing and gather results with the oracle alone -- bit-exact -- and, where the golden
involves torch arithmetic (sigmoid/top-k, Conv/BN), check the build's restatement of it.
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def gather_xyz(xyz, idx):
    return np.take_along_axis(xyz, idx[..., None].astype(np.int64).repeat(3, axis=2), axis=1)


def test_ops_small(oracle):
    g = load("ops_small")
    xyz, feats = g["xyz"], g["feats"]
    np.testing.assert_array_equal(oracle.fps(xyz, 64), g["fps_idx"])
    new_xyz = gather_xyz(xyz, g["fps_idx"])
    np.testing.assert_array_equal(new_xyz, g["new_xyz"])
    np.testing.assert_array_equal(oracle.ball_query(0.3, 8, xyz, new_xyz), g["bq"])
    np.testing.assert_array_equal(oracle.ball_query_dilated(0.4, 0.1, 8, xyz, new_xyz), g["bqd"])
    np.testing.assert_array_equal(oracle.ball_query_dilated(0.3, 0.0, 8, xyz, new_xyz), g["bqd0"])
    np.testing.assert_array_equal(oracle.group_points(feats, g["bq"]), g["grouped"])
    rel = oracle.group_points(np.ascontiguousarray(xyz.transpose(0, 2, 1)), g["bq"]) - new_xyz.transpose(0, 2, 1)[..., None]
    np.testing.assert_array_equal(np.concatenate([rel, g["grouped"]], 1), g["qg"])
    np.testing.assert_array_equal(rel, g["qg_nofeat"])
    d2, idx = oracle.three_nn(g["unknown"], new_xyz)
    np.testing.assert_array_equal(idx, g["nn_idx"])
    np.testing.assert_allclose(np.sqrt(d2), g["nn_dist"], rtol=2e-7)  # torch.sqrt (vectorised) vs numpy: 1 ulp
    np.testing.assert_array_equal(oracle.three_interpolate(g["known_feats"], g["nn_idx"], g["interp_w"]), g["interp"])
    np.testing.assert_array_equal(oracle.fps_with_dist(g["dmat"], 16), g["fps_d"])
    np.testing.assert_array_equal(oracle.group_points_grad(g["group_go"], g["bq"], 256), g["group_grad"])
    np.testing.assert_array_equal(oracle.gather_points_grad(g["gather_go"], g["fps_idx"], 256), g["gather_grad"])
    np.testing.assert_array_equal(oracle.three_interpolate_grad(g["interp_go"], g["nn_idx"], g["interp_w"], 64),
                                  g["interp_grad"])


@pytest.mark.parametrize("name,npoint", [("config1_sa", 512), ("sampler_dfps_stds_dilated", 128)])
def test_dfps_layers(oracle, name, npoint):
    g = load(name)
    idx = oracle.fps(g["xyz"], npoint)
    np.testing.assert_array_equal(idx, g["idx"])
    np.testing.assert_array_equal(gather_xyz(g["xyz"], idx), g["new_xyz"])


def topk_equivalent(mine, ref, ref_scores, tol):
    """Same index at every rank, except where the reference's own scores are within `tol` (torch.topk
    leaves tie order, and torch.sigmoid its last ulp, unspecified)."""
    for b in range(ref.shape[0]):
        bad = np.flatnonzero(mine[b] != ref[b])
        for r in bad:
            assert abs(float(ref_scores[b, mine[b, r]]) - float(ref_scores[b, ref[b, r]])) <= tol, (b, r)


@pytest.mark.parametrize("name", ["sampler_ctr", "sampler_nogroup"])
def test_ctr_sampler(oracle, name):
    g = load(name)
    s = oracle.score_ctr(g["cls_in"])
    ref_scores = 1.0 / (1.0 + np.exp(-g["cls_in"].max(-1).astype(np.float64)))
    np.testing.assert_allclose(s, ref_scores, rtol=3e-7)
    idx = oracle.topk_desc(s, g["idx"].shape[1])
    topk_equivalent(idx, g["idx"], ref_scores, 1e-6)


def test_sss_sampler(oracle):
    g = load("sampler_sss")
    s = oracle.score_stability(g["cls_in"], g["kw_stds"])
    cls = 1.0 / (1.0 + np.exp(-g["cls_in"].max(-1).astype(np.float64)))
    sta = 1.0 - 1.0 / (1.0 + np.exp(-(g["kw_stds"].astype(np.float64) / 8 - 3)))
    np.testing.assert_allclose(s, cls * sta, rtol=2e-6, atol=1e-7)
    idx = oracle.topk_desc(s, g["idx"].shape[1])
    topk_equivalent(idx, g["idx"], cls * sta, 1e-6)
    if (idx == g["idx"]).all():
        np.testing.assert_array_equal(np.take_along_axis(g["kw_stds"], idx.astype(np.int64), 1), g["stds_out"])


def test_ffps_fs_samplers(oracle):
    # F-FPS: FPS over the |a|^2+|b|^2-2ab matrix of xyz (+) features (pointnet2_modules.py:19-43,357-369)
    for name, npoint in (("sampler_ffps", 64), ("sampler_fs", 32)):
        g = load(name)
        joint = np.concatenate([g["xyz"], g["feats"].transpose(0, 2, 1)], -1).astype(np.float32)
        import torch
        a = torch.from_numpy(joint)
        sq = (a * a).sum(-1, keepdim=True)
        dist = (sq.expand(-1, -1, a.shape[1]) + sq.transpose(1, 2).expand(-1, a.shape[1], -1)
                - 2.0 * torch.matmul(a, a.transpose(1, 2))).numpy()
        by_feat = oracle.fps_with_dist(dist, npoint)
        if name == "sampler_ffps":
            np.testing.assert_array_equal(by_feat, g["idx"])
        else:
            np.testing.assert_array_equal(np.concatenate([by_feat, oracle.fps(g["xyz"], npoint)], -1), g["idx"])


def test_sfps_samplers(oracle):
    """S-FPS (pointnet2_modules.py:314-353): FPS seeds, the steadiest point of each seed's ball, and the hard-coded
    fallback to the seeds when scene 0 keeps fewer than 3500 distinct picks (`stds` keeps the steadiest picks' values
    either way)."""
    for name, radius, ns in (("sampler_sfps_fallback", 0.8, 8), ("sampler_sfps", 0.15, 4)):
        g = load(name)
        xyz, stds, m = g["xyz"], g["kw_stds"], g["idx"].shape[1]
        seeds = oracle.fps(xyz, m)
        ball = oracle.ball_query(radius, ns, xyz, gather_xyz(xyz, seeds))
        ball_stds = np.take_along_axis(stds[:, None, :], ball.reshape(ball.shape[0], 1, -1).astype(np.int64), 2)
        steadiest = ball_stds.reshape(ball.shape).argmin(-1)          # first minimum, like torch.argmin
        picks = np.take_along_axis(ball, steadiest[..., None], 2)[..., 0]
        fell_back = np.unique(picks[0]).size < 3500
        assert fell_back == (name == "sampler_sfps_fallback")
        np.testing.assert_array_equal(seeds if fell_back else picks, g["idx"])
        np.testing.assert_array_equal(np.take_along_axis(stds, picks.astype(np.int64), 1), g["stds_out"])


@pytest.mark.parametrize("name", ["sampler_dsfps", "sampler_ryfps"])
def test_partitioned_fps_samplers(oracle, name):
    """ds-FPS / ry-FPS (pointnet2_modules.py:370-419): each scene sorted by range-5 (or atan(x/y)), cut into four equal
    parts, FPS of npoint/4 inside each, indices mapped back."""
    import torch
    g = load(name)
    xyz, m = g["xyz"], g["idx"].shape[1]
    out = []
    for b in range(xyz.shape[0]):
        p = torch.from_numpy(xyz[b])
        key = p.norm(dim=-1) - 5 if name == "sampler_dsfps" else torch.atan(p[:, 0] / p[:, 1])
        order = key.sort(dim=0, descending=False)[1].numpy()
        parts = order.reshape(4, -1)
        picks = oracle.fps(np.ascontiguousarray(xyz[b][parts]), m // 4)          # (4, m/4) inside the parts
        out.append(np.take_along_axis(parts, picks.astype(np.int64), 1).reshape(-1))
    np.testing.assert_array_equal(np.stack(out).astype(np.int32), g["idx"])
    np.testing.assert_array_equal(gather_xyz(xyz, g["idx"]), g["new_xyz"])


def test_generator_layer_identity_sampling(oracle):
    g = load("generator_layer")
    assert (g["idx"] == np.arange(512, dtype=np.int32)[None]).all()
    np.testing.assert_array_equal(g["new_xyz"], g["xyz"])


def test_stack3_indices(oracle):
    g = load("stack3_small")
    i0 = oracle.fps(g["xyz"], 512)
    np.testing.assert_array_equal(i0, g["l0_idx"])
    x1 = gather_xyz(g["xyz"], i0)
    np.testing.assert_array_equal(x1, g["l0_new_xyz"])
    i1 = oracle.fps(x1, 128)
    np.testing.assert_array_equal(i1, g["l1_idx"])
    x2 = gather_xyz(x1, i1)
    s = oracle.score_ctr(g["l1_cls"])
    ref_scores = 1.0 / (1.0 + np.exp(-g["l1_cls"].max(-1).astype(np.float64)))
    topk_equivalent(oracle.topk_desc(s, 64), g["l2_idx"], ref_scores, 1e-6)
    np.testing.assert_array_equal(gather_xyz(x2, g["l2_idx"]), g["l2_new_xyz"])


def test_rand_sampler_fixture_is_one_shared_permutation_prefix():
    """pointnet2_modules.py:370-371: `randperm(N)[None, :npoint].int().repeat(B, 1)` -- the reference's rows are one prefix of
    one permutation (distinct, in range, identical for every scene) and its centroids are exactly those points."""
    g = load("sampler_rand")
    idx, xyz = g["idx"], g["xyz"]
    assert idx.dtype == np.int32 and idx.shape == (xyz.shape[0], 256)
    assert (idx == idx[:1]).all() and np.unique(idx[0]).size == 256 and idx.min() >= 0 and idx.max() < xyz.shape[1]
    np.testing.assert_array_equal(g["new_xyz"], np.take_along_axis(xyz, idx[..., None].astype(np.int64), 1))
