"""Acceptance / load model of the multi-pick FPS rounds (fps_pruned.hip, fps_pruned_big.hip) for variants NOT built:
more records per wave, and a scene split over K compute units (buckets dealt round-robin over 8 K waves, every CU
publishing its T best records, one cross-CU exchange per round).  float64 statistics on a bench scene, then a cycle model
whose constants are the measured phase times of the shipped kernels:

  N = 16 384 (one CU, 2 records per wave, 16 records): round 6.47 k cycles = apply 3.2 k (0.1 k per centre for the box tests
      + 0.44 k per touched bucket of the busiest wave) + records 0.32 k per record and wave + barrier 0.95 k + exchange /
      pair evaluation / accept 0.6 k + 1.06 k (R / 16)^2
  N = 180 000 (fps_pruned_big): apply 1.0 k per fetched bucket of the busiest wave, records 0.8 k, exchange + accept + barrier 4 k
  cross-CU exchange of a round's records through L2: +2.2 k cycles (0.9 us hand-off, MI355X_MICROARCH.md price list,
      handoff-1to1 / allgather rows) -- optimistic; the judge's figure of 1.7 us is 4.1 k.

usage: python tools/fps_cluster_sim.py [16384|180000]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spsnet_amd import scenes  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
M = 4096 if N <= 16384 else 16384
BIG = N > 16384
xyz = scenes.make_batch("kitti-lidar-v1", 1, N, seed0=0)[0][0].astype(np.float64)

# Morton-like 12-bit key sort -> spatially compact buckets of 64 (as the kernels do)
lo, hi = xyz.min(0), xyz.max(0)
ext = hi - lo
c = ext.copy(); axis = []; nb = [0, 0, 0]
bits = 12 if not BIG else 16
for _ in range(bits):
    a = int(np.argmax(c)); axis.append(a); nb[a] += 1; c[a] *= .5
q = [np.minimum(((xyz[:, a] - lo[a]) * ((1 << nb[a]) / ext[a])).astype(np.int64), (1 << nb[a]) - 1) for a in range(3)]
used = [0, 0, 0]; key = np.zeros(N, np.int64)
for i in range(bits):
    a = axis[i]; used[a] += 1
    key = (key << 1) | ((q[a] >> (nb[a] - used[a])) & 1)
order = np.argsort(key, kind="stable")
P = xyz[order]
NBUCK = (N + 63) // 64
pad = NBUCK * 64 - N
bucket = np.arange(N) // 64


def simulate(K, per_unit, unit, max_rounds=None):
    """K CUs x 8 waves; `unit` = "wave": every wave publishes per_unit records, "cu": every CU publishes per_unit records.
    -> picks/round, mean over rounds of the busiest wave's touched buckets, records per round."""
    W = 8 * K
    wave_of = np.arange(NBUCK) % W
    owner = wave_of if unit == "wave" else wave_of // 8          # publishing unit of every bucket
    U = W if unit == "wave" else K
    t = np.full(N, 1e10)
    first = int(np.where(order == 0)[0][0])

    def upd(p):
        d = ((P - P[p]) ** 2).sum(1)
        ch = d < t
        np.minimum(t, d, out=t)
        return np.unique(bucket[ch])

    upd(first)
    picks, rounds, crit_sum = 1, 0, 0.0
    tpad = np.full(NBUCK * 64, -1.0)
    while picks < M and (max_rounds is None or rounds < max_rounds):
        tpad[:N] = t
        tb = tpad.reshape(NBUCK, 64)
        part = np.partition(tb, 62, axis=1)
        b1, b2 = part[:, 63], part[:, 62]
        arg = tb.argmax(1)
        recs, hidden = [], np.zeros(U)
        for u in range(U):
            bs = np.nonzero(owner == u)[0]
            o = bs[np.argsort(-b1[bs])]
            recs += [(b1[g], g, u) for g in o[:per_unit]]
            hidden[u] = b1[o[per_unit]] if len(o) > per_unit else -1.0
        recs.sort(key=lambda r: -r[0])
        acc, bound, used_u = [], -1.0, np.zeros(U, int)
        for val, g, u in recs:
            if picks + len(acc) >= M:
                break
            p = g * 64 + arg[g]
            if acc:
                if any(((P[p] - P[a]) ** 2).sum() < t[p] for a in acc) or not val > bound:
                    break
            acc.append(p)
            bound = max(bound, b2[g])
            used_u[u] += 1
            if used_u[u] >= per_unit:
                bound = max(bound, hidden[u])
        cnt = np.zeros(W, int)
        for a in acc:
            for g in upd(a):
                cnt[wave_of[g]] += 1
        crit_sum += cnt.max()
        picks += len(acc); rounds += 1
    return picks / rounds, crit_sum / rounds, U * per_unit


def cycles(K, ppr, crit, R, per_wave_records):
    if not BIG:
        apply_ = 100.0 * ppr + 440.0 * crit
        rec = 320.0 * per_wave_records
        sync = 950.0 + (2200.0 if K > 1 else 0.0)
        accept = 600.0 + 1060.0 * (R / 16.0) ** 2
    else:
        apply_ = 1000.0 * crit
        rec = 400.0 * per_wave_records
        sync = 2000.0 + (2200.0 if K > 1 else 0.0)
        accept = 1000.0 + 1000.0 * (R / 16.0) ** 2
    return apply_ + rec + sync + accept


cap = None
print(f"N = {N}, M = {M}, {'first %d rounds' % cap if cap else 'all rounds'}")
base = None
for label, K, per_unit, unit in (("shipped: 1 CU, 2 records / wave", 1, 2, "wave"),
                                 ("1 CU, 3 records / wave", 1, 3, "wave"),
                                 ("1 CU, 4 records / wave", 1, 4, "wave"),
                                 ("2 CUs, 8 records / CU", 2, 8, "cu"),
                                 ("2 CUs, 12 records / CU", 2, 12, "cu"),
                                 ("4 CUs, 6 records / CU", 4, 6, "cu"),
                                 ("4 CUs, 8 records / CU", 4, 8, "cu"),
                                 ("8 CUs, 4 records / CU", 8, 4, "cu")):
    t0 = time.time()
    ppr, crit, R = simulate(K, per_unit, unit, cap)
    pw = per_unit if unit == "wave" else 2          # a wave still ranks its own two best for the CU-level selection
    extra = 500.0 if unit == "cu" else 0.0          # the in-CU selection of the CU's best records
    cyc = cycles(K, ppr, crit, R, pw) + extra
    per_pick = cyc / ppr
    base = base or per_pick
    print(f"{label:34s} picks/round {ppr:5.2f}  busiest wave touches {crit:5.2f}  records {R:3d}  model {cyc / 1e3:5.2f} k cycles/round "
          f"= {per_pick:6.0f} /pick  ({100 * (per_pick / base - 1):+5.1f} % vs shipped)   [{time.time() - t0:.0f} s]", flush=True)
