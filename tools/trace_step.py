"""Print the kernel timeline of the last bench step from a rocprofv3 --kernel-trace sqlite file (rocpd)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, queue_id from kernels order by start"))
fi = [i for i, r in enumerate(rows) if 'fps_pruned' in r[0]]
i0 = fi[-1]
t0 = rows[i0][1]
for r in rows[i0:]:
    nm = re.sub(r'\(.*', '', r[0]).replace('void ', '')[:70]
    print(f"{(r[1]-t0)/1e3:9.1f} {(r[2]-t0)/1e3:9.1f} {(r[2]-r[1])/1e3:8.1f} q{r[3]} {nm}")
