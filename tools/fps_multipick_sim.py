"""CPU simulation of the multi-pick FPS rounds of fps_pruned.hip (acceptance statistics only, float64): how many
picks does a round accept when every wave publishes its TOPB best buckets?  usage: [W=8] python fps_multipick_sim.py TOPB KMAX [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spsnet_amd import scenes
N, M, W = 16384, 4096, int(os.environ.get("W", 8))
TOPB = int(sys.argv[1]) if len(sys.argv) > 1 else 1   # published buckets per wave
KMAX = int(sys.argv[2]) if len(sys.argv) > 2 else 8
xyz = scenes.make_batch("kitti-lidar-v1", 1, N, seed0=int(sys.argv[3]) if len(sys.argv) > 3 else 0)[0][0].astype(np.float64)
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
NB = N // 64
bucket = np.arange(N) // 64
wave = bucket % W
t = np.full(N, 1e10)
# first pick: original index 0
first = int(np.where(order == 0)[0][0])
def upd(p):
    global t
    d = ((P - P[p]) ** 2).sum(1)
    ch = d < t
    t = np.minimum(t, d)
    return np.unique(bucket[ch])
upd(first)
picks = 1; rounds = 0; hist = np.zeros(KMAX + 1, int); crit = []; crit1 = []
rej = {"a": 0, "b": 0, "c": 0, "end": 0}
while picks < M:
    tb = t.reshape(NB, 64)
    srt = np.sort(tb, axis=1)
    b1 = srt[:, -1]; b2 = srt[:, -2]; arg = tb.argmax(1)
    # per wave: TOPB best buckets + hidden bound
    recs = []; hidden = {}
    for w in range(W):
        bs = np.arange(w, NB, W)
        o = bs[np.argsort(-b1[bs])]
        for g in o[:TOPB]: recs.append((b1[g], g, w))
        hidden[w] = b1[o[TOPB]]
    recs.sort(key=lambda r: -r[0])
    acc = []; bound = -1.0; touched_by = []
    for (val, g, w) in recs:
        if len(acc) >= KMAX or picks + len(acc) >= M: rej["end"] += 1; break
        p = g * 64 + arg[g]
        if acc:
            if any(((P[p] - P[a]) ** 2).sum() < t[p] for a in acc): rej["a"] += 1; break
            if not val > bound: rej["b"] += 1; break
        acc.append(p)
        bound = max(bound, b2[g])
        # hidden bound of wave w applies once all its TOPB published buckets are consumed... conservative: after the last published one
        nw = sum(1 for a in acc if wave[a] == w)
        if nw >= TOPB: bound = max(bound, hidden[w])
    cnt = np.zeros(W, int)
    for a in acc:
        tb_ = upd(a)
        for g in tb_: cnt[g % W] += 1
    crit.append(cnt.max()); hist[len(acc)] += 1
    picks += len(acc); rounds += 1
print(f"TOPB={TOPB} KMAX={KMAX}: rounds={rounds} picks/round={picks/rounds:.2f} mean crit buckets/round={np.mean(crit):.2f} per pick={np.sum(crit)/picks:.2f} hist={hist.tolist()} rej={rej}")
