"""One training step of a rocprofv3 kernel trace (rocpd db) as a table: start (us from the step's first kernel), duration,
hardware queue, grid, kernel. usage: step_trace.py results.db [which_step_from_the_end=3]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
rows = db.execute("select s.kernel_name, d.start, d.end, d.queue_id, d.grid_size_x, d.grid_size_y, d.workgroup_size_x from %s d join %s s "
                  "on d.kernel_id = s.id order by d.start" % (T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"))).fetchall()
sol = [r[2] for r in rows if "solver_kernel" in r[0]]
t0, t1 = sol[-back - 1], sol[-back]
sel = [r for r in rows if r[1] >= t0 and r[2] <= t1]
base = sel[0][1]
qs = sorted(set(r[3] for r in sel))
print("step wall %.3f ms, %d kernels, queues %s" % ((t1 - t0) / 1e6, len(sel), qs))
prev_end = {}
for n, a, b, q, gx, gy, wx in sel:
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
    gap = (a - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = b
    print("%9.1f %8.1f  q%d gap %7.1f  wg %6d x %-3d %s" % ((a - base) / 1e3, (b - a) / 1e3, qs.index(q), gap, gx // max(wx, 1), gy, n[:70]))
