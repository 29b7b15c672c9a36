"""Prints per-kernel call counts and average durations from a rocprofv3 results database (debug aid)."""
import sqlite3
import sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if 'kernel_dispatch' in t][0]
sym = [t for t in tabs if 'kernel_symbol' in t][0]
q = f"select s.kernel_name, count(*), avg(d.end-d.start) from {kt} d join {sym} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"
for r in list(c.execute(q))[:int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
    print(f"{r[0][:90]:90s} {r[1]:6d} {r[2]/1e3:9.1f} us")
