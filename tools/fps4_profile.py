"""Where does a round of the FOUR-wave FPS kernel (csrc/fps_pruned4.hip) spend its cycles?  (diagnostic build, s_memtime stamps)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spsnet_amd
from spsnet_amd import _lib, scenes
L = _lib.load()
B, N, M = 8, 16384, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
spsnet_amd.init("cuda:0")
xyz, _ = scenes.make_batch("kitti-lidar-v1", B, N, seed0=0)
x = torch.from_numpy(xyz).cuda()
temp = torch.full((B, N), 1e10, device="cuda")
idx = torch.zeros((B, M), dtype=torch.int32, device="cuda")
work = torch.empty((B * int(L.sps_fps_workspace_floats(N)),), dtype=torch.float32, device="cuda")
dbg = torch.zeros((B, 4, 12), dtype=torch.int64, device="cuda")
_lib.check(L.sps_debug_fps4_profile(B, N, M, x.data_ptr(), temp.data_ptr(), idx.data_ptr(), work.data_ptr(), dbg.data_ptr(), 0), "profile")
torch.cuda.synchronize()
d = dbg.cpu().numpy().astype(np.float64)
names = ["apply", "refresh", "candidate", "barrier", "accept"]
it = M - 1
print(f"N={N} m={M}: per-ROUND cycles per wave (mean over scenes; s_memtime ticks), stamp cost included in each segment")
for w in range(4):
    row = d[:, w].mean(0)
    rounds = row[5]
    print(f"wave {w}: " + "  ".join(f"{n}={row[i]/rounds:7.1f}" for i, n in enumerate(names)) +
          f"  total/round={row[:5].sum()/rounds:7.1f}  picks/round={it/rounds:.2f}  ticks/pick={row[:5].sum()/it:7.1f}  "
          f"touched/pick={row[6]/it:.2f} refreshed/pick={row[7]/it:.2f}  whole loop {row[9]-row[8]:.0f} ticks")
