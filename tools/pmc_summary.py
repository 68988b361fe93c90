"""Prints per-kernel sums of the PMC counters found in a rocprofv3 rocpd database."""
import sqlite3, sys, subprocess
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
def T(p): return [t for t in tabs if t.startswith(p)][0]
q = """select s.kernel_name, p.symbol, count(*), sum(e.value) from %s e join %s p on e.pmc_id = p.id
       join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id group by s.kernel_name, p.symbol""" % (
    T("rocpd_pmc_event"), T("rocpd_info_pmc"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"))
try:
    rows = db.execute(q).fetchall()
except Exception as ex:
    print("query failed", ex)
    for t in ("rocpd_pmc_event", "rocpd_info_pmc"):
        print(t, [r[1] for r in db.execute("pragma table_info(%s)" % T(t))])
    sys.exit(1)
for r in rows:
    n = r[0]
    if "igemm" in n or len(sys.argv) > 2:
        print("%-70s %-28s n=%d sum=%.4g per=%.4g" % (n[20:90], r[1], r[2], r[3], r[3] / r[2]))
