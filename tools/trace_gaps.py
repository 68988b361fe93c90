"""GPU idle gaps inside one training step of a rocprofv3 kernel trace (rocpd sqlite)."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute("select d.start, d.end, s.kernel_name from %s d join %s s on d.kernel_id=s.id order by d.start" % (disp, sym)).fetchall()
def short(n):
    m = re.search(r"N_1\d+([a-z_0-9]+?)(I|E)", n)
    return m.group(1) if m else n[:40]
sol = [i for i, r in enumerate(rows) if "solver_kernel" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(sol) // 2
a, b = sol[k], sol[k + 1]
step = rows[a:b + 1]
wall = (step[-1][1] - step[0][1]) / 1e6
kern = sum(r[1] - r[0] for r in step[1:]) / 1e6
print("step %d: wall %.2f ms, kernels %.2f ms, idle %.2f ms, %d dispatches" % (k, wall, kern, wall - kern, len(step) - 1))
gaps = sorted(((step[i][0] - step[i - 1][1]) / 1e3, short(step[i - 1][2]), short(step[i][2])) for i in range(1, len(step)))[::-1]
for g in gaps[:12]:
    print("%9.1f us  after %-28s before %s" % g)
print("gaps < 20us: total %.2f ms" % (sum(g[0] for g in gaps if g[0] < 20) / 1e3))
agg = {}
for r in step[1:]:
    n = short(r[2]); agg[n] = agg.get(n, 0) + (r[1] - r[0]) / 1e6
for n, v in sorted(agg.items(), key=lambda kv: -kv[1])[:16]:
    print("  %-28s %7.2f ms" % (n, v))
