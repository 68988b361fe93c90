"""Summarises a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel table: the `--stats` view.
usage: python tools/rocpd_stats.py <results.db> [steps] [--skip-steps K] > profiles/xxx.md
--skip-steps K drops everything up to the K-th solver_kernel (the warm-up steps, where the launch configurations are
being measured: candidate kernels run several times each and would distort the per-kernel averages);
--last-steps L keeps only the last L steps (bench.py's live roofline leg: its 5 instrumented steps run every kernel alone
on one stream, whereas the timed steps overlap two kernel chains and stretch the individual durations)"""
import sqlite3, subprocess, sys

db = sqlite3.connect(sys.argv[1])
args = [a for a in sys.argv[2:] if not a.startswith("--")]
steps = int(args[0]) if args else None
skip = int(sys.argv[sys.argv.index("--skip-steps") + 1]) if "--skip-steps" in sys.argv else 0
last = int(sys.argv[sys.argv.index("--last-steps") + 1]) if "--last-steps" in sys.argv else 0
args = [a for i, a in enumerate(sys.argv[2:]) if not a.startswith("--") and not sys.argv[2:][i - 1].startswith("--")] \
    if len(sys.argv) > 2 else []
steps = int(args[0]) if args else None
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
t_min = 0
if skip or last:
    sol = db.execute("select d.end from %s d join %s s on d.kernel_id = s.id where s.kernel_name like '%%solver_kernel%%' "
                     "order by d.start" % (disp, sym)).fetchall()
    t_min = sol[skip - 1][0] if skip else sol[len(sol) - last - 1][0]
rows = db.execute("select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
                  "max(d.end-d.start) from %s d join %s s on d.kernel_id = s.id where d.start >= %d group by s.kernel_name "
                  "order by 3 desc" % (disp, sym, t_min)).fetchall()
tot = sum(r[2] for r in rows)


def demangle(n):
    n = n[:-3] if n.endswith(".kd") else n
    try:
        d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
        d = d.replace("(anonymous namespace)::", "")
        if d.startswith("void "):
            d = d[5:]
        return d.split("(")[0]
    except Exception:
        return n


print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---:|---:|---:|---:|---:|---:|")
for name, calls, total, avg, mn, mx in rows:
    print("| %s | %d | %.2f | %.1f | %.1f | %.1f | %.1f |" % (demangle(name), calls,
                                                             total / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * total / tot))
print("\ntotal kernel time %.2f ms over %d dispatches" % (tot / 1e6, sum(r[1] for r in rows))
      + (" = %.2f ms per step (%d steps)" % (tot / 1e6 / steps, steps) if steps else ""))
