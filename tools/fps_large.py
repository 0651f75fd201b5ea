"""FPS at the Waymo-shaped sizes (BASELINE config 5 / SURVEY 8 Waymo row): where the large-N path stands."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spsnet_amd import scenes
import spsnet_amd.pointnet2_batch_cuda as ext
for B, N, m in ((8, 65536, 16384), (8, 180000, 16384), (8, 16384, 4096)):
    xyz = torch.from_numpy(scenes.make_batch("kitti-lidar-v1", B, N, seed0=0)[0]).cuda()
    idx = torch.zeros((B, m), dtype=torch.int32, device="cuda")
    ts = []
    for _ in range(2):
        temp = torch.full((B, N), 1e10, device="cuda")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ext.farthest_point_sampling_wrapper(B, N, m, xyz, temp, idx)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"B={B} N={N} m={m}: {min(ts)*1e3:9.2f} ms")
