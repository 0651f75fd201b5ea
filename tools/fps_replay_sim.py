"""CPU simulation (float64, acceptance statistics only) of a REPLAY acceptance for the multi-pick FPS rounds of fps_pruned.hip:
instead of stopping at the first published record that an earlier pick of the round lowers, the records' running distances are
updated pick by pick and the exact FPS is replayed AMONG THE RECORDS for as long as the best one provably beats everything that
is hidden (bucket runner-ups of picked or lowered records; the unpublished buckets of a wave whose weakest published record
was picked or lowered).  usage: [W=8] python tools/fps_replay_sim.py TOPB [seed]   -> picks per round for both schemes"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spsnet_amd import scenes
N, M, W = 16384, 4096, int(os.environ.get("W", 8))
TOPB = int(sys.argv[1]) if len(sys.argv) > 1 else 2
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
xyz = scenes.make_batch("kitti-lidar-v1", 1, N, seed0=seed)[0][0].astype(np.float64)
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


def run(replay):
    t = np.full(N, 1e10)
    first = int(np.where(order == 0)[0][0])
    t = np.minimum(t, ((P - P[first]) ** 2).sum(1))
    picks, rounds, steps = 1, 0, []
    seq = [first]
    while picks < M:
        tb = t.reshape(NB, 64)
        srt = np.sort(tb, axis=1)
        b1 = srt[:, -1]; b2 = srt[:, -2]; arg = tb.argmax(1)
        recs = []          # (value, point, bucket, wave, is_last_of_wave)
        hidden = np.zeros(W)
        for w in range(W):
            bs = np.arange(w, NB, W)
            o = bs[np.argsort(-b1[bs], kind='stable')]
            for k, g in enumerate(o[:TOPB]):
                recs.append([b1[g], g * 64 + arg[g], g, w, k == TOPB - 1])
            hidden[w] = b1[o[TOPB]]
        R = len(recs)
        cur = np.array([r[0] for r in recs]); orig = cur.copy()
        pts = np.array([r[1] for r in recs])
        picked = np.zeros(R, bool)
        acc = []
        while picks + len(acc) < M and not picked.all():
            cand = np.where(~picked)[0]
            c_ = cand[np.argmax(cur[cand])]
            gone = picked | (cur < orig)                       # picked or lowered: their hidden points need their bounds
            bound = -1.0
            for r in np.where(gone)[0]:
                bound = max(bound, b2[recs[r][2]])
                if recs[r][4]: bound = max(bound, hidden[recs[r][3]])
            if acc and not cur[c_] > bound:
                break
            if not replay and acc and cur[c_] < orig[c_]:
                break                                            # the shipped scheme: a lowered record ends the round
            acc.append(pts[c_]); picked[c_] = True
            d = ((P[pts] - P[pts[c_]]) ** 2).sum(1)
            cur = np.minimum(cur, d)
        for a in acc:
            t = np.minimum(t, ((P - P[a]) ** 2).sum(1))
        seq += acc
        picks += len(acc); rounds += 1; steps.append(len(acc))
    return picks / rounds, np.bincount(steps, minlength=R + 1).tolist(), seq


base, hb, s0 = run(False)
rep, hr, s1 = run(True)
assert s0[:M] == s1[:M], "the two schemes must pick the same sequence"
print(f"TOPB={TOPB} W={W}: shipped acceptance {base:.2f} picks/round {hb}")
print(f"TOPB={TOPB} W={W}: replay acceptance  {rep:.2f} picks/round {hr}")
