"""A/B at the HEADLINE shape (8 x 16 384 -> 4096): the register-resident FPS kernel (one CU per scene, fps_pruned.hip) against the
clustered kernel (K workgroups per scene exchanging T records per round, fps_pruned_cluster.hip; SPS_FPS_CLUSTER_SMALL=1).
usage: python tools/fps_cluster_small.py [B] [N] [m] ["K,T;K,T;..."]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spsnet_amd import _lib, scenes
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
m = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
shapes = (sys.argv[4] if len(sys.argv) > 4 else "8,8;8,4;4,8;8,6;4,16").split(";")
L = _lib.load()
dev = torch.device("cuda:0")
_lib.ensure_init(dev)
x = torch.from_numpy(scenes.make_batch("kitti-lidar-v1", B, N, seed0=0)[0]).to(dev)
work = torch.empty((B * int(L.sps_fps_workspace_floats(N)),), dtype=torch.float32, device=dev)


def run():
    best, out = 1e9, None
    for _ in range(4):
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=dev)
        idx = torch.empty((B, m), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(L.sps_fps_with_workspace(B, N, m, x.data_ptr(), temp.data_ptr(), idx.data_ptr(), work.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream), "fps")
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
        out = (idx.cpu().numpy(), temp.cpu().numpy())
    return best, out


os.environ.pop("SPS_FPS_CLUSTER_SMALL", None)
t0, ref = run()
print(f"{B} x {N} -> {m}: register-resident kernel behind the sorting pre-pass {t0 * 1e3:.0f} us", flush=True)
os.environ["SPS_FPS_CLUSTER_SMALL"] = "1"
for tag in shapes:
    os.environ["SPS_FPS_CLUSTER"] = tag
    t, got = run()
    print(f"  clustered K,T = {tag:5s}: {t * 1e3:.0f} us ({t / t0:.2f}x)  identical={np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])}", flush=True)
