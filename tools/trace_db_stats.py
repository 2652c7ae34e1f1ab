"""Per-kernel totals of a rocprofv3 --kernel-trace results .db:  python tools/trace_db_stats.py <results.db> [fits]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); fits = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
q = f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"
tot = 0
for r in c.execute(q):
    tot += r[2]
    print("%-100s launches/fit %7.1f  ms/fit %8.3f  avg us %8.1f" % (r[0][:100], r[1] / fits, r[2] / fits, r[3]))
t0, t1 = c.execute(f"select min(start), max(end) from {kd}").fetchone()
print("sum of kernel time / fit %.3f ms; first start -> last end %.3f ms" % (tot / fits, (t1 - t0) / 1e6))
