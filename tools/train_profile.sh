# kernel statistics + an in-order kernel timeline of ONE training step (SA layers 0-2, forward + backward)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trainprof
rm -rf $O; mkdir -p $O
cd $R
python3 tools/train_step_time.py 8 16384 10 > $O/train_plain.log 2>&1 &&
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python3 tools/train_step_time.py 8 16384 5 > $O/train_profiled.log 2>&1
python3 - <<'PY'
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/trainprof"
f = glob.glob(O + "/kt/*kernel_trace.csv")
if f:
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last step: from the last fps_pruned launch on
    starts = [i for i, r in enumerate(rows) if "fps_pruned_kernel" in r["Kernel_Name"]]
    i0 = starts[-1]
    t0 = int(rows[i0]["Start_Timestamp"])
    with open(O + "/last_step_timeline.txt", "w") as out:
        for r in rows[i0:]:
            s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
            out.write(f"{s/1e3:10.1f} {e/1e3:10.1f} {(e-s)/1e3:8.1f} q{r.get('Queue_Id','?')} {r['Kernel_Name'][:100]}\n")
PY
find $O -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" -delete
find $O -name "*kernel_trace.csv" -delete
tail -1 $O/train_plain.log
