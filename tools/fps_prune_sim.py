"""how many buckets survive the prune test per pick: 64-point boxes vs two 32-point sub-boxes vs four 16-point ones"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spsnet_amd import scenes
N, M = 16384, 4096
xyz = scenes.make_batch("kitti-lidar-v1", 1, N, seed0=0)[0][0].astype(np.float64)
lo, hi = xyz.min(0), xyz.max(0)
ext = hi - lo
c = ext.copy(); axis = []; nb = [0, 0, 0]
for i in range(12):
    a = int(np.argmax(c)); axis.append(a); nb[a] += 1; c[a] *= .5
q = [np.minimum(((xyz[:, a] - lo[a]) * ((1 << nb[a]) / ext[a])).astype(int), (1 << nb[a]) - 1) for a in range(3)]
used = [0, 0, 0]; key = np.zeros(N, int)
for i in range(12):
    a = axis[i]; used[a] += 1; sh = nb[a] - used[a]
    key = (key << 1) | ((q[a] >> sh) & 1)
order = np.argsort(key, kind='stable')
P = xyz[order]
t = np.full(N, 1e10)
def boxes(sz):
    B = P.reshape(-1, sz, 3)
    return B.min(1), B.max(1)
bx = {sz: boxes(sz) for sz in (64, 32, 16)}
first = int(np.where(order == 0)[0][0])
cur = first
surv = {64: 0, 32: 0, 16: 0}; truly = 0; changed_pts = 0
for j in range(1, M):
    pc = P[cur]
    for sz in (64, 32, 16):
        blo, bhi = bx[sz]
        qq = np.clip(pc, blo, bhi)
        lb = ((qq - pc) ** 2).sum(1)
        mx = t.reshape(-1, sz).max(1)
        s = (lb < mx)
        if sz == 64: surv[64] += s.sum()
        else: surv[sz] += s.reshape(-1, 64 // sz).any(1).sum()     # a 64-bucket survives if any of its sub-boxes does
    d = ((P - pc) ** 2).sum(1)
    ch = d < t
    truly += len(np.unique(np.where(ch)[0] // 64)); changed_pts += ch.sum()
    t = np.minimum(t, d)
    cur = int(t.argmax())
print({k: v / (M - 1) for k, v in surv.items()}, "buckets with a changed point per pick:", truly / (M - 1), "changed points per pick:", changed_pts / (M - 1))
