"""Time the FPS kernels at several sample counts (setup cost = the m=2 row)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spsnet_amd import _lib, scenes
import spsnet_amd.pointnet2_batch_cuda as ext
L = _lib.load()
for N, ms in ((16384, (2, 512, 1024, 4096)), (4096, (2, 256, 1024))):
    xyz = torch.from_numpy(scenes.make_batch("kitti-lidar-v1", 8, N, seed0=0)[0]).cuda()
    for mode in (0, 1):
        L.sps_set_fps_mode(mode)
        for m in ms:
            idx = torch.zeros((8, m), dtype=torch.int32, device="cuda")
            ts = []
            for _ in range(5):
                temp = torch.full((8, N), 1e10, device="cuda")
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); ext.farthest_point_sampling_wrapper(8, N, m, xyz, temp, idx); e.record()
                torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
            print(f"N={N} m={m:5d} mode={'pruned' if mode == 0 else 'brute '}: {min(ts)*1e3:8.1f} us")
    L.sps_set_fps_mode(0)
