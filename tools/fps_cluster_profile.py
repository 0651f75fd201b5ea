"""Phase times of the clustered large-scene FPS (fps_pruned_cluster.hip built with -DSPS_PC_PROFILE: per-wave s_memtime sums
left in the tail of the workspace histogram area).  Prints, per phase, the mean / max over the waves per round.
usage: python tools/fps_cluster_profile.py [N] [m] [K,T] [scenes]
needs the diagnostic build: hipcc ... -DSPS_PC_PROFILE -c fps_pruned_cluster.hip, relinked into a library of its own and named by
SPS_LIBSPSNET_SA (tools/r5f.sh shows the three commands)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from spsnet_amd import _lib, scenes

N = int(sys.argv[1]) if len(sys.argv) > 1 else 180000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
shape = sys.argv[3] if len(sys.argv) > 3 else "8,4"
os.environ["SPS_FPS_CLUSTER"] = shape
K = int(shape.split(",")[0])
L = _lib.load()
dev = torch.device("cuda:0")
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1          # (the profile is read from scene 0's exchange area)
if N <= 16384:
    os.environ["SPS_FPS_CLUSTER_SMALL"] = "1"
xyz = torch.from_numpy(scenes.make_batch("kitti-lidar-v1", B, N, seed0=0)[0]).to(dev)
wf = int(L.sps_fps_workspace_floats(N))
work = torch.zeros((B * wf,), dtype=torch.float32, device=dev)
temp = torch.full((B, N), 1e10, dtype=torch.float32, device=dev)
idx = torch.empty((B, m), dtype=torch.int32, device=dev)
_lib.check(L.sps_fps_with_workspace(B, N, m, xyz.data_ptr(), temp.data_ptr(), idx.data_ptr(), work.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream), "fps")
torch.cuda.synchronize()
npad = (N + 63) // 64 * 64
PC_MAXK, PF_BINS, PC_MAXR = 16, 4096, 64      # (csrc/fps_sort_split.h)
granules = 8 + 2 * PC_MAXR * 6 + 2 * PC_MAXK + 8
area = work[5 * npad:wf].view(torch.int64).cpu().numpy()          # the exchange area as 8-byte words
tail_end = granules + PC_MAXK * PF_BINS // 2
nw = 8 * K
rec = area[tail_end - 8 * nw:tail_end].reshape(nw, 8)
rounds = rec[0, 6]
names = ["apply", "records", "barrier 1", "rank + publish", "poll + barrier 2", "accept"]
print(f"N={N} m={m} K,T={shape}: {rounds} rounds, {m / max(rounds, 1):.2f} picks per round; per round and wave, in s_memtime ticks (= shader cycles on gfx950, ~2.4 GHz):")
tot = 0.0
for i, nm in enumerate(names):
    cyc = rec[:, i] / rounds
    tot += cyc.mean()
    print(f"  {nm:18s} mean {cyc.mean():7.0f}   max over waves {cyc.max():7.0f}   min {cyc.min():7.0f}")
print(f"  sum of means {tot:.0f} cycles per round = {tot / 2.4e3:.2f} us at 2.4 GHz")
