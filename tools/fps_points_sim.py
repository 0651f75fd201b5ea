"""picks per round when every wave publishes its T best POINTS (not the maxima of its T best buckets)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spsnet_amd import scenes
N, M, W = 16384, 4096, int(os.environ.get("W", 8))
T = int(sys.argv[1]); KMAX = int(sys.argv[2]); MODE = sys.argv[3] if len(sys.argv) > 3 else "points"
xyz = scenes.make_batch("kitti-lidar-v1", 1, N, seed0=int(sys.argv[4]) if len(sys.argv) > 4 else 0)[0][0].astype(np.float64)
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
widx = [np.where(wave == w)[0] for w in range(W)]
t = np.full(N, 1e10)
first = int(np.where(order == 0)[0][0])
def upd(p):
    global t
    d = ((P - P[p]) ** 2).sum(1)
    t = np.minimum(t, d)
upd(first)
picks = 1; rounds = 0; rej = {"lowered": 0, "hidden": 0, "end": 0}
while picks < M:
    recs = []; hidden = np.zeros(W)
    for w in range(W):
        ids = widx[w]; tw = t[ids]
        if MODE == "points":
            o = np.argsort(-tw)[:T + 1]
            for k in o[:T]: recs.append((tw[k], ids[k], w))
            hidden[w] = tw[o[T]] if os.environ.get("LOOSE", "0") == "0" else tw[o[T - 1]]
        else:   # maxima of the T best buckets (what the kernel does now): hidden = max(rest of those buckets, next bucket)
            tb = tw.reshape(-1, 64); b1 = tb.max(1); ob = np.argsort(-b1)
            rest = []
            for g in ob[:T]:
                k = int(tb[g].argmax()); recs.append((tb[g][k], ids[g * 64 + k], w))
                rest.append(np.partition(tb[g], -2)[-2])
            hidden[w] = max(max(rest), b1[ob[T]])   # (upper bound once ALL the wave's records are gone; per-record bounds below)
    recs.sort(key=lambda r: -r[0])
    acc = []; taken = np.zeros(W, int); bound = -1.0
    perrec = {}
    if MODE != "points":
        # per-record exposure: after record of bucket g is picked, the rest of bucket g is exposed
        for w in range(W):
            ids = widx[w]; tw = t[ids]; tb = tw.reshape(-1, 64); b1 = tb.max(1); ob = np.argsort(-b1)
            for g in ob[:T]:
                k = int(tb[g].argmax()); perrec[ids[g * 64 + k]] = np.partition(tb[g], -2)[-2]
    if os.environ.get("RERANK", "0") == "1":
        vals = {p: val for (val, p, w) in recs}; wv = {p: w for (val, p, w) in recs}
        while vals and len(acc) < KMAX and picks + len(acc) < M:
            p = max(vals, key=lambda k: vals[k]); val = vals[p]
            if acc and not val > bound: rej["hidden"] += 1; break
            del vals[p]; acc.append(p); w = wv[p]; taken[w] += 1
            if MODE != "points": bound = max(bound, perrec[p])
            if taken[w] >= T: bound = max(bound, hidden[w])
            for k in vals:
                d = ((P[k] - P[p]) ** 2).sum()
                if d < vals[k]:
                    vals[k] = d
                    # a lowered record no longer shields its wave's unpublished points: they may now exceed it
                    if os.environ.get("EXPOSE", "1") == "1" and d < hidden[wv[k]]: bound = max(bound, hidden[wv[k]])
        else:
            rej["end"] += 1
        recs = []
    for (val, p, w) in recs:
        if len(acc) >= KMAX or picks + len(acc) >= M: rej["end"] += 1; break
        if acc:
            if any(((P[p] - P[a]) ** 2).sum() < t[p] for a in acc): rej["lowered"] += 1; break
            if not val > bound: rej["hidden"] += 1; break
        acc.append(p); taken[w] += 1
        if MODE != "points": bound = max(bound, perrec[p])
        if taken[w] >= T: bound = max(bound, hidden[w])
    else:
        rej["end"] += 1
    for a in acc: upd(a)
    picks += len(acc); rounds += 1
print(f"MODE={MODE} T={T} KMAX={KMAX}: rounds={rounds} picks/round={picks/rounds:.2f} rej={rej}")
