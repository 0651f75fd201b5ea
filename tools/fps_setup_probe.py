"""Set-up cost of the register-resident FPS (8 x 16 384 points): launch time for m = 2, 8, 64, 512, 4096 picks through the plain
wrapper (HIP events, best of 5).  m = 2 is essentially the sort + bucket loads + first maxima (~75 us of the 1.79 ms launch).
usage: python tools/fps_setup_probe.py   (SPS_FPS_PRESORT has no effect here: the plain wrapper never uses the pre-pass)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spsnet_amd import pointnet2_batch_cuda as ext, scenes
dev = torch.device("cuda:0")
B, N = 8, 16384
xyz = torch.from_numpy(scenes.make_batch("kitti-lidar-v1", B, N, seed0=0)[0]).to(dev)
for m in (2, 8, 64, 512, 4096):
    best = 1e9
    for _ in range(5):
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=dev)
        idx = torch.empty((B, m), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ext.farthest_point_sampling_wrapper(B, N, m, xyz, temp, idx); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"m={m}: {best*1e3:.1f} us")
