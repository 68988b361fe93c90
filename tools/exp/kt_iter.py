"""Per-iteration GPU statistics of a kernel trace: iterations end at the kernel whose name contains `marker`.
usage: kt_iter.py results.db marker [last_n=10]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); marker = sys.argv[2]; n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
rows = db.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (
    T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"))).fetchall()
ends = [r[2] for r in rows if marker in r[0]]
t0, t1 = ends[-n - 1], ends[-1]
sel = [r for r in rows if r[1] >= t0 and r[2] <= t1]
busy = 0; cur_end = t0
for _, a, b in sel:
    if b > cur_end:
        busy += b - max(a, cur_end); cur_end = b
print("per iteration: wall %.3f ms, GPU busy %.3f ms, %d kernels, sum of durations %.3f ms" % (
    (t1 - t0) / n / 1e6, busy / n / 1e6, len(sel) // n, sum(b - a for _, a, b in sel) / n / 1e6))
agg = {}
for nme, a, b in sel:
    k = nme.split("(")[0][:70]; agg[k] = agg.get(k, [0, 0]); agg[k][0] += 1; agg[k][1] += b - a
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%6.1f us/iter %4d x  %s" % (t / n / 1e3, c // n, k))
