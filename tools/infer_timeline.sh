# in-order kernel timeline of the LAST timed pass of the headline bench (what runs beside and behind layer 0's FPS)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/inferprof
rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --output-format csv --kernel-trace -d $O/kt -o kt -- python3 bench.py --steps 5 --warmup 3 --no-fp32-leg --no-training-leg --no-cpu-baseline --no-validate > $O/bench_profiled.log 2>&1
python3 - <<'PY'
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/inferprof"
f = glob.glob(O + "/kt/*kernel_trace.csv")
if f:
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "fps_presort_kernel" in r["Kernel_Name"]]
    i0, i1 = starts[7], starts[8] if len(starts) > 8 else len(rows)      # the last timed pass (3 warm-up + 5 timed)
    t0 = int(rows[i0]["Start_Timestamp"])
    with open(O + "/last_pass_timeline.txt", "w") as out:
        for r in rows[i0:i1]:
            s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
            out.write(f"{s/1e3:10.1f} {e/1e3:10.1f} {(e-s)/1e3:8.1f} q{r.get('Queue_Id','?')} {r['Kernel_Name'][:110]}\n")
PY
find $O -type f ! -name "*.log" ! -name "*.txt" -delete
