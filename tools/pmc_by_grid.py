"""Per (kernel, grid) sums of the PMC counters of a rocprofv3 rocpd database: one line per distinct launch shape, counters
summed over the counter instances of a dispatch and averaged over the dispatches. Usage: pmc_by_grid.py results.db [substr]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else "igemm"
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]


def T(p):
    return [t for t in tabs if t.startswith(p)][0]


q = """select s.kernel_name, d.grid_size_x, d.grid_size_y, d.dispatch_id, d.end - d.start, p.symbol, sum(e.value)
       from %s e join %s p on e.pmc_id = p.id join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id
       group by d.dispatch_id, p.symbol""" % (T("rocpd_pmc_event"), T("rocpd_info_pmc"), T("rocpd_kernel_dispatch"),
                                               T("rocpd_info_kernel_symbol"))
agg = collections.OrderedDict()
for name, gx, gy, did, dur, sym, val in db.execute(q):
    if sub not in name:
        continue
    key = (name.split("(")[0][-60:], gx, gy)
    a = agg.setdefault(key, {"n": set(), "dur": {}, "c": collections.defaultdict(float)})
    a["n"].add(did)
    a["dur"][did] = dur
    a["c"][sym] += val
for (name, gx, gy), a in agg.items():
    n = len(a["n"])
    dur = sum(a["dur"].values()) / n / 1e3
    print("%s grid %dx%d  n=%d  %.1f us" % (name, gx // 256, gy, n, dur))
    print("    " + "  ".join("%s=%.4g" % (k, v / n) for k, v in sorted(a["c"].items())))
