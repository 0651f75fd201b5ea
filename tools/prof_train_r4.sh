# kernel statistics of the LAST training step of IASSD_Backbone (MIOpen's find pass in the warm-up steps pollutes whole-run stats)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4f
rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --output-format csv --kernel-trace -d $O/kt -o kt -- python3 tools/backbone_train_time.py 8 16384 4 IASSD > $O/train_profiled.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r4f"
f = glob.glob(O + "/kt/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "fps_pruned_kernel" in r["Kernel_Name"]]
i0 = starts[-1]
last = rows[i0:]
t0, t1 = int(last[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in last)
agg = collections.defaultdict(lambda: [0, 0])
for r in last:
    a = agg[r["Kernel_Name"][:110]]
    a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
with open(O + "/last_step_kernels.txt", "w") as out:
    out.write(f"last step: {(t1 - t0) / 1e6:.3f} ms wall from the FPS launch to the last kernel's end, {len(last)} kernels, "
              f"{sum(v[1] for v in agg.values()) / 1e6:.3f} ms summed kernel time\n")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        out.write(f"{v[1] / 1e3:9.1f} us {v[0]:4d} x  {k}\n")
PY
find $O/kt -type f -delete
head -40 $O/last_step_kernels.txt
