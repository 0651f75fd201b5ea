"""Large-scene FPS spread over K workgroups (fps_pruned_cluster.hip): every (K, T) against the one-workgroup kernel
(indices and final running distances bit for bit) and its time per launch (HIP events, best of 3).
usage: python tools/fps_cluster_probe.py [N] [m] [scenes]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spsnet_amd import pointnet2_batch_cuda as ext  # noqa: E402
from spsnet_amd import scenes  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 180000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
xyz = torch.from_numpy(scenes.make_batch("kitti-lidar-v1", B, N, seed0=0)[0]).to(dev)


def run(tag):
    os.environ["SPS_FPS_CLUSTER"] = tag
    best, out = 1e9, None
    for _ in range(3):
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=dev)
        idx = torch.empty((B, m), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ext.farthest_point_sampling_wrapper(B, N, m, xyz, temp, idx)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
        out = (idx.cpu().numpy(), temp.cpu().numpy())
    return best, out


t1, ref = run("1")
print(f"N={N} m={m} scenes={B}: one workgroup {t1:.3f} ms", flush=True)
for tag in (sys.argv[4].split(";") if len(sys.argv) > 4 else ("2,8", "4,6", "4,8", "8,4", "8,3", "6,5")):
    t, got = run(tag)
    same = np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    print(f"  K,T = {tag:5s}: {t:.3f} ms  ({t / t1:.2f}x)  identical={same}", flush=True)
    if not same:
        bad = np.nonzero(got[0] != ref[0])
        tb = np.nonzero(got[1] != ref[1])
        print(f"      idx differs at {bad[0].size} places, first pick {bad[1][:5] if bad[0].size else '-'}; temp differs at {tb[0].size} points"
              f"; picks are a permutation of the reference's: {np.array_equal(np.sort(got[0]), np.sort(ref[0]))}", flush=True)
