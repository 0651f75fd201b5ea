"""Summarise a rocprofv3 results database: per-kernel count / average / total.  usage: python tools/prof_db.py file.db [top]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = cur.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                   f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc limit {top}").fetchall()
for r in rows:
    print(f"{r[0][:64]:64s} n={r[1]:5d} avg={r[2]/1e3:9.1f}us min={r[4]/1e3:8.1f} max={r[5]/1e3:8.1f} tot={r[3]/1e6:8.2f}ms")
