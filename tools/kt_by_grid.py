"""Per (kernel, grid) average duration from a rocprofv3 rocpd kernel trace. usage: kt_by_grid.py results.db [substr]"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
q = "select s.kernel_name, d.grid_size_x, d.grid_size_y, d.workgroup_size_x, count(*), avg(d.end-d.start), min(d.end-d.start) from %s d join %s s on d.kernel_id = s.id group by 1,2,3 order by d.start" % (
    T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"))
for name, gx, gy, wx, n, avg, mn in db.execute(q):
    if sub in name:
        print("%-64s grid %6dx%-3d n=%4d avg %8.1f us min %8.1f us" % (name.split("(")[0][-64:], gx // max(wx, 1), gy, n, avg / 1e3, mn / 1e3))
