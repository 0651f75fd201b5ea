"""Randomised shapes through the two pieces of host / kernel plumbing added at the end of round 2 (GPU box only):
  * pointnet2_utils.group_with_index with gradients (_GroupConcat) against the op sequence it replaces;
  * OrderedPrefix with its second pass in random pieces against the one-call form and the plain FPS kernel.
usage: python tools/round2_fuzz.py [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from spsnet_amd import pointnet2_utils as U, pointnet2_batch_cuda as ext

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
rng = np.random.default_rng(2026)
worst = 0.0
for case in range(cases):
    B, N = int(rng.integers(1, 5)), int(rng.choice([1, 2, 17, 64, 500, 1024, 4099, 20000]))
    M, ns, C = int(rng.integers(1, 300)), int(rng.choice([1, 2, 5, 16, 32, 64])), int(rng.choice([1, 3, 16, 31, 64, 131]))
    use_xyz, new_grad = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    gen = torch.Generator().manual_seed(case)
    xyz = (torch.rand(B, N, 3, generator=gen) * 5).to(dev)
    idx = torch.randint(0, N, (B, M, ns), generator=gen, dtype=torch.int32).to(dev)
    f0, q0 = torch.randn(B, C, N, generator=gen), torch.rand(B, M, 3, generator=gen) * 5
    w = torch.randn(B, C + (3 if use_xyz else 0), M, ns, generator=gen).to(dev)
    res = {}
    for flag in (True, False):
        U.GROUP_CONCAT_TRAINING = flag
        f = f0.to(dev).requires_grad_(True)
        q = q0.to(dev).requires_grad_(new_grad)
        out = U.group_with_index(xyz, q, f, idx, use_xyz)
        (out * w).sum().backward()
        res[flag] = (out.detach(), f.grad, q.grad)
    assert torch.equal(res[True][0], res[False][0]), ("values", case)
    for a, b in zip(res[True][1:], res[False][1:]):
        assert (a is None) == (b is None), case
        if a is not None:
            err = float((a - b).abs().max()) / max(1.0, float(b.abs().max()))
            worst = max(worst, err)
            assert err <= 2e-5, (case, B, N, M, ns, C, err)
U.GROUP_CONCAT_TRAINING = True
print(f"group_with_index with gradients: {cases} cases, worst relative difference to the op sequence {worst:.2e}")

from spsnet_amd import scenes
ok = 0
for case in range(max(cases // 3, 8)):
    B = int(rng.integers(1, 4))
    N = int(rng.choice([130, 640, 1000, 2048, 4096]))
    m = int(rng.integers(2, N + 1)) if rng.integers(0, 3) == 0 else int(rng.choice([N // 4, N // 2]))
    base, _ = scenes.make_batch("kitti-lidar-v1", B, 4 * N, seed0=1000 + case, dup_fraction=0.02 if case % 4 == 0 else 0.0)
    bx = torch.from_numpy(base).to(dev)
    pick = U.furthest_point_sample(bx, N)
    x1 = torch.gather(bx, 1, pick.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    if case % 3 == 0 and B > 1:
        x1[B - 1] = x1[B - 1][torch.randperm(N, device=dev)]
    want = U.furthest_point_sample(x1, m)
    one, flags1, temp1 = ext.fps_ordered_prefix(x1, m, return_flags=True)
    assert torch.equal(one, want), ("one call", case)
    op = ext.OrderedPrefix(x1, m)
    op.begin()
    for c in sorted(int(v) for v in rng.integers(0, N + 64, size=int(rng.integers(0, 5)))):
        op.check_upto(c)
    got = op.finish()
    assert torch.equal(got, want), ("pieces", case, N, m)
    assert torch.equal(op.flags, flags1) and torch.equal(op.temp, temp1), ("flags / temp", case)
    ok += 1
print(f"OrderedPrefix in pieces: {ok} cases identical to the one-call form and the plain kernel")
