"""Kernel timeline of the LAST pass in a rocprofv3 kernel-trace csv of tools/tail_events.py (or bench.py): every kernel that ends
after the pass's FPS producer started its last quarter, relative to the producer's end -- what runs behind the last pick.
usage: python tools/tail_trace.py <kernel_trace.csv>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
prod = [r for r in rows if ("fps_pruned_kernel" in r["Kernel_Name"] or "fps_pruned_cluster_kernel" in r["Kernel_Name"]) and (r["e"] - r["s"]) > 1_000_000]
last = prod[-1]
t0 = last["e"]
nxt = [r for r in rows if r["s"] > last["s"]]
print(f"producer {last['Kernel_Name'][:60]} {(last['e'] - last['s']) / 1e3:.1f} us")
prev_end = None
for r in nxt:
    if r["e"] < t0 - 150_000:
        continue
    name = r["Kernel_Name"].replace("void sps::", "").split("(")[0][:70]
    gap = "" if prev_end is None else f"gap {(r['s'] - prev_end) / 1e3:6.1f}"
    print(f"{(r['s'] - t0) / 1e3:8.1f} -> {(r['e'] - t0) / 1e3:8.1f}  ({(r['e'] - r['s']) / 1e3:6.1f} us) {gap:12s} {name}")
    prev_end = max(prev_end or 0, r["e"])
