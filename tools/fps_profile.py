"""Where does an iteration of the pruned FPS kernel spend its cycles?  (diagnostic build, s_memtime stamps)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spsnet_amd import _lib, scenes
L = _lib.load()
B, N, M = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 16384, int(sys.argv[2]) if len(sys.argv) > 2 else 4096
DATASET = sys.argv[3] if len(sys.argv) > 3 else "kitti-lidar-v1"
xyz, _ = scenes.make_batch(DATASET, B, N, seed0=0)
x = torch.from_numpy(xyz).cuda()
temp = torch.full((B, N), 1e10, device="cuda")
idx = torch.zeros((B, M), dtype=torch.int32, device="cuda")
dbg = torch.zeros((B, 8, 12), dtype=torch.int64, device="cuda")
_lib.check(L.sps_debug_fps_profile(B, N, M, x.data_ptr(), temp.data_ptr(), idx.data_ptr(), dbg.data_ptr(), 0), "profile")
torch.cuda.synchronize()
d = dbg.cpu().numpy().astype(np.float64)
names = ["apply", "candidate", "publish", "barrier", "accept"]
it = M - 1
print(f"{DATASET} N={N} m={M}: per-ROUND cycles per wave (mean over scenes), stamp cost ~40 included in each segment")
for w in range(8):
    row = d[:, w].mean(0)
    rounds = row[5]
    print(f"wave {w}: " + "  ".join(f"{n}={row[i]/rounds:7.1f}" for i, n in enumerate(names)) + f"  total/round={row[:5].sum()/rounds:7.1f}  picks/round={it/rounds:.2f}  cycles/pick={row[:5].sum()/it:7.1f}  touched/pick={row[6]/it:.2f} tie-path/pick={row[7]/it:.3f}")
w = d[:, 0, 8:12].sum(0)
rounds = w[:3].sum()
print(f"rounds/scene {rounds / B:.0f}, picks/round {w[3] / rounds:.2f}; the accepted prefix ended because the next record was lowered by an "
      f"earlier one: {w[0] / rounds:.1%}, hidden behind an unpublished point of an earlier record's wave: {w[1] / rounds:.1%}, nothing "
      f"rejected: {w[2] / rounds:.1%}")
