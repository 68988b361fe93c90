"""Timeline statistics of the last `steps` training steps of a rocprofv3 kernel trace (rocpd db): wall time per step, time with
0 / 1 / >=2 kernels in flight, and per-kernel-class busy time. usage: timeline.py results.db [steps=5]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
skip_last = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
rows = db.execute("select s.kernel_name, d.start, d.end, d.queue_id from %s d join %s s on d.kernel_id = s.id order by d.start" % (
    T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"))).fetchall()
sol = [r[2] for r in rows if "solver_kernel" in r[0]]
if skip_last:
    sol = sol[:-skip_last]
t0, t1 = sol[-steps - 1], sol[-1]
sel = [r for r in rows if r[1] >= t0 and r[2] <= t1]
ev = []
for n, a, b, q in sel:
    ev.append((a, 1)); ev.append((b, -1))
ev.sort()
busy = {0: 0, 1: 0, 2: 0}
cur, last = 0, t0
for t, d in ev:
    busy[min(cur, 2)] += t - last
    last = t
    cur += d
busy[min(cur, 2)] += t1 - last
wall = (t1 - t0) / steps / 1e6
print("wall %.3f ms/step | idle %.3f | one kernel %.3f | >=2 kernels %.3f | sum of kernel durations %.3f" % (
    wall, busy[0] / steps / 1e6, busy[1] / steps / 1e6, busy[2] / steps / 1e6, sum(b - a for _, a, b, _ in sel) / steps / 1e6))
queues = {}
for n, a, b, q in sel:
    queues[q] = queues.get(q, 0) + (b - a)
print("kernel time per queue (ms/step):", {q: round(v / steps / 1e6, 2) for q, v in queues.items()})
