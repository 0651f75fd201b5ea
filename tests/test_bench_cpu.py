"""bench.py's host-side logic that needs no GPU: the oracle verdict attached to `validated` (the timed batch against the CPU
oracle's pass over the same scenes and weights)."""
import copy
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("sps_bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_oracle_verdict_accepts_the_oracle_and_names_what_differs(oracle):
    from oracle import cpu_stack
    from spsnet_amd import pointnet2_modules as M, sa_stack, scenes
    bench = _bench()
    cfg = sa_stack.scaled_config(npoints=[256, 64, 32])
    layers = cpu_stack.cpu_copy(sa_stack.build_sa_layers(M, cfg, seed=2))
    xyz, feats = scenes.make_batch("kitti-lidar-v1", 2, 1024, seed0=5)
    want = cpu_stack.sa_stack_cpu(layers, xyz, feats)
    samplers = [m[0] for m in cfg["sample_method_list"]]
    rec, bad = bench.oracle_verdict(copy.deepcopy(want), want, samplers, 1e-4)
    assert not bad and rec["ok"] and rec["idx_exact"] and rec["new_xyz_exact"] and rec["max_abs_err_features"] == 0.0
    assert [r["sampler"] for r in rec["layers"]] == samplers and rec["scenes"] == 2

    got = [tuple(None if a is None else a.copy() for a in o) for o in want]
    got[0][3][1, 7] += 1                                   # one D-FPS pick off by one
    rec, bad = bench.oracle_verdict(got, want, samplers, 1e-4)
    assert not rec["ok"] and not rec["idx_exact"] and any("layer 0" in b and "indices" in b for b in bad)

    got = [tuple(None if a is None else a.copy() for a in o) for o in want]
    got[1][1][0, 3, 5] += 1e-2 * max(1.0, float(np.abs(want[1][1]).max()))
    rec, bad = bench.oracle_verdict(got, want, samplers, 1e-4)
    assert not rec["ok"] and rec["idx_exact"] and any("layer 1: features" in b for b in bad)

    # a score-sampled layer whose picks swapped a near-tie: rows are matched by sampled index, not by position
    got = [tuple(None if a is None else a.copy() for a in o) for o in want]
    perm = np.arange(want[2][3].shape[1])
    perm[[3, 4]] = perm[[4, 3]]
    got[2] = (got[2][0][:, perm], got[2][1][:, :, perm], got[2][2][:, perm], got[2][3][:, perm])
    rec, bad = bench.oracle_verdict(got, want, samplers, 1e-4)
    assert not bad and rec["layers"][2]["picks_shared_with_oracle"] == 1.0
