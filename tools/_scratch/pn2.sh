cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pn2prof
rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python3 tools/backbone_time.py 8 16384 10 ${1:-PointNet2} > $O/log.txt 2>&1
python3 - <<'PY'
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pn2prof"
f = glob.glob(O + "/kt/*kernel_trace.csv")
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "fps_presort_kernel" in r["Kernel_Name"]]
i0, i1 = starts[-2], starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
with open(O + "/timeline.txt", "w") as out:
    for r in rows[i0:i1]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        out.write(f"{s/1e3:10.1f} {e/1e3:10.1f} {(e-s)/1e3:8.1f} q{r.get('Queue_Id','?')} {r['Kernel_Name'][:100]}\n")
PY
find $O -type f ! -name "*.txt" ! -name "*stats.csv" -delete
