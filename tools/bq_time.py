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

# small launches: lane-per-centroid (scalar / vector loads) vs wave-per-centroid
def small(label, pts, ctr, ra, na, rb, nb, j0, cnt):
    ia = torch.zeros((pts.shape[0], ctr.shape[1], na), dtype=torch.int32, device="cuda")
    ib = torch.zeros((pts.shape[0], ctr.shape[1], nb), dtype=torch.int32, device="cuda")
    for name, env in (("lane/centroid", {"SPS_BQ_WAVE": "0"}), ("wave/centroid", {"SPS_BQ_WAVE": "1"})):
        os.environ.update(env)
        print(f"{label:28s} {name:14s}: {t(lambda: ext.ball_query_full2_range(ra, rb, pts, ctr, ia, ib, j0, cnt)):8.1f} us")
        for k in env: os.environ.pop(k)
for cnt in (256, 512, 1024, 2048, 4096):
    small(f"L0 chunk of {cnt}", xyz, new_xyz, 0.2, 16, 0.8, 32, 4096 - cnt, cnt)
idx1 = U.furthest_point_sample(new_xyz, 1024); xyz1 = ext.gather_xyz(new_xyz, idx1)
small("L1 8x1024 over 4096", new_xyz, xyz1, 0.8, 16, 1.6, 32, 0, 1024)
xyz2 = xyz1[:, :512].contiguous()
small("L2 8x512 over 1024", xyz1, xyz2, 1.6, 16, 4.8, 32, 0, 512)
