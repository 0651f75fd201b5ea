"""A randomised parity sweep beyond the fixed cases of test_parity_gpu.py (run by hand on a GPU box:
`python tests/sweep_parity_gpu.py [cases] [seed]`; not collected by pytest).  Every case draws a cloud kind (uniform, LiDAR-like,
lattice with masses of exact ties, duplicated points), a shape and the op's parameters, runs the HIP path through the C ABI and
the CPU oracle on the same input and demands bit-identical results: FPS (indices AND final running distances; all kernels:
register-resident, streaming, workspace / clustered), ball query (plain, dilated), grouping, 3-NN."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O          # noqa: E402  (test infrastructure: this file lives under tests/)
from tests import gpu_util as G         # noqa: E402
import spsnet_amd.pointnet2_batch_cuda as ext  # noqa: E402
from spsnet_amd import scenes           # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
O.build()
rng = np.random.default_rng(seed)


def cloud(B, N):
    kind = rng.choice(["uniform", "lidar", "lattice", "dup"])
    if kind == "lidar" and N >= 64:
        return kind, scenes.make_batch("kitti-lidar-v1", B, N, seed0=int(rng.integers(0, 10 ** 6)))[0]
    if kind == "lattice":
        return kind, rng.integers(-4, 5, (B, N, 3)).astype(np.float32) * 0.25
    xyz = rng.uniform(-3, 3, (B, N, 3)).astype(np.float32)
    if kind == "dup" and N > 4:
        k = max(1, N // 10)
        for b in range(B):
            xyz[b, rng.choice(N, k, replace=False)] = xyz[b, rng.integers(0, N, k)]
    return kind, xyz


done = {"fps": 0, "ball_query": 0, "dilated": 0, "group": 0, "three_nn": 0}
t0 = time.time()
for case in range(cases):
    op = rng.choice(["fps", "fps", "fps_big", "ball_query", "dilated", "group", "three_nn"])
    if op in ("fps", "fps_big"):
        B = int(rng.integers(1, 5))
        N = int(rng.choice([1, 2, 63, 64, 65, 500, 1024, 2049, 4096, 6143, 6144, 9000, 16384])) if op == "fps" else int(rng.choice([16385, 20000, 33000, 50000]))
        m = int(rng.integers(1, min(N, 3000 if op == "fps" else 1500) + 1))
        kind, xyz = cloud(B, N)
        want, want_t = O.fps(xyz, m, return_temp=True)
        got, got_t = G.fps(ext, xyz, m)
        assert np.array_equal(got, want) and np.array_equal(got_t, want_t), (op, kind, B, N, m)
        done["fps"] += 1
    elif op in ("ball_query", "dilated"):
        B, N, M = int(rng.integers(1, 4)), int(rng.choice([1, 50, 777, 4096, 16384])), int(rng.choice([1, 33, 256, 1000]))
        ns = int(rng.choice([1, 4, 16, 32, 64]))
        kind, xyz = cloud(B, N)
        new_xyz = xyz[:, rng.integers(0, N, M)].copy() + rng.normal(0, 0.01, (B, M, 3)).astype(np.float32) * int(rng.integers(0, 2))
        r = float(rng.choice([0.05, 0.2, 0.8, 2.0, 50.0]))
        if op == "ball_query":
            assert np.array_equal(G.ball_query(ext, r, ns, xyz, new_xyz), O.ball_query(r, ns, xyz, new_xyz)), (op, kind, B, N, M, ns, r)
        else:
            rmin = float(rng.choice([0.0, r / 2]))
            assert np.array_equal(G.ball_query(ext, r, ns, xyz, new_xyz, dilated_min=rmin),
                                  O.ball_query_dilated(r, rmin, ns, xyz, new_xyz)), (op, kind, B, N, M, ns, r, rmin)
        done[op] += 1
    elif op == "group":
        B, C, N, M, ns = int(rng.integers(1, 4)), int(rng.choice([1, 3, 17, 64])), int(rng.choice([1, 100, 5000])), int(rng.choice([1, 64, 300])), int(rng.choice([1, 16, 32]))
        pts = rng.normal(size=(B, C, N)).astype(np.float32)
        idx = rng.integers(0, N, (B, M, ns)).astype(np.int32)
        assert np.array_equal(G.group(ext, pts, idx), O.group_points(pts, idx)), (op, B, C, N, M, ns)
        done["group"] += 1
    else:
        B, n, m = int(rng.integers(1, 4)), int(rng.choice([1, 100, 3000])), int(rng.choice([3, 50, 2500]))
        _, unknown = cloud(B, n)
        kind, known = cloud(B, m)
        d_t = torch.empty((B, n, 3), dtype=torch.float32, device=G.DEV)
        i_t = torch.empty((B, n, 3), dtype=torch.int32, device=G.DEV)
        ext.three_nn_wrapper(B, n, m, G.t(unknown), G.t(known), d_t, i_t)
        wd, wi = O.three_nn(unknown, known)          # (both sides: squared distances, the extension's output)
        assert np.array_equal(G.n(i_t), wi) and np.array_equal(G.n(d_t), wd), (op, kind, B, n, m)
        done["three_nn"] += 1
print(f"parity sweep, seed {seed}: {cases} random cases bit-identical to the oracle in {time.time() - t0:.0f} s -- {done}")
