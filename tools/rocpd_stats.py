"""Summarises a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel table: the `--stats` view.
usage: python tools/rocpd_stats.py <results.db> [steps] > profiles/xxx.md"""
import sqlite3, subprocess, sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute("select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
                  "max(d.end-d.start) from %s d join %s s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"
                  % (disp, sym)).fetchall()
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
