"""Ball-query kernel timings at the IA-SSD layer-0 shape (8 x 4096 centroids over 16384 points)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spsnet_amd import scenes
import spsnet_amd.pointnet2_batch_cuda as ext
from spsnet_amd import pointnet2_utils as U
xyz = torch.from_numpy(scenes.make_batch("kitti-lidar-v1", 8, 16384, seed0=0)[0]).cuda()
idx = U.furthest_point_sample(xyz, 4096)
new_xyz = ext.gather_xyz(xyz, idx)
def t(fn, n=5):
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) * 1e3)
    return min(ts)
for ra, rb in ((0.2, 0.8), (1e-4, 2e-4), (0.2, 0.2), (0.8, 0.8), (3.0, 3.0)):
    print(f"radii ({ra},{rb}) dual grouped  : {t(lambda: ext.ball_query_full2(ra, 16, rb, 32, xyz, new_xyz, True)):8.1f} us")
    print(f"radii ({ra},{rb}) dual ungrouped: {t(lambda: ext.ball_query_full2(ra, 16, rb, 32, xyz, new_xyz, False)):8.1f} us")
    print(f"radii ({ra},{rb}) dual wave/ctr : {t(lambda: ext.ball_query_full2(ra, 16, rb, 32, xyz, new_xyz, False, True)):8.1f} us")
for r, ns in ((0.2, 16), (0.8, 32)):
    print(f"single r={r} ns={ns}: {t(lambda: ext.ball_query_full(r, ns, xyz, new_xyz)):8.1f} us")
