"""Per-kernel totals of ONE steady-state step from a rocprofv3 kernel trace (CSV): the dispatches between the last two
launches of a marker kernel (e.g. the layer-0 FPS kernel, which runs once per step).
usage: python tools/prof_step.py <kernel_trace.csv> <marker substring> [top]"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
seg = rows[marks[-2]:marks[-1]]
agg = defaultdict(lambda: [0, 0.0])
for r in seg:
    a = agg[r["Kernel_Name"][:100]]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
busy = sum(v[1] for v in agg.values())
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
print(f"{len(seg)} dispatches, sum of kernel times {busy:.0f} us, span {span:.0f} us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{v[1]:9.1f} us  n={v[0]:4d}  {k}")
