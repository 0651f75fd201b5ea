"""A/B of the eight-wave (fps_pruned.hip) and four-wave (fps_pruned4.hip, SPS_FPS_WAVES=4) register-resident FPS kernels behind
the sorting pre-pass: 8 x 16 384 -> 4096, the publishing launch the streamed layer uses, HIP events, oracle check of the picks.
usage: [SPS_FPS_WAVES=4] python tools/fps_waves_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spsnet_amd import scenes
import spsnet_amd.pointnet2_batch_cuda as ext
B, N, M = 8, 16384, 4096
xyz_np = scenes.make_batch("kitti-lidar-v1", B, N, seed0=0)[0]
xyz = torch.from_numpy(xyz_np).cuda()
idx = torch.zeros((B, M), dtype=torch.int32, device="cuda")
progress = torch.zeros((B,), dtype=torch.int32, device="cuda")
ts = []
for rep in range(8):
    progress.zero_(); idx.zero_()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); work = ext.fps_publish(xyz, None, idx, progress); e.record()
    torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
print(f"waves={os.environ.get('SPS_FPS_WAVES', '8')}: publish launch (pre-pass + kernel) min {min(ts)*1e3:.1f} us, median {np.median(ts)*1e3:.1f} us")
if "--check" in sys.argv:
    from oracle import oracle as O
    want = O.fps(xyz_np[:2], M)
    print("picks identical to the oracle (2 scenes):", bool(np.array_equal(idx[:2].cpu().numpy(), want)), "progress", progress.tolist())
