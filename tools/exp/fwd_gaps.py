"""Gaps between consecutive kernels of the compute stream's hardware queue in the timed steps of a rocprofv3 kernel trace:
how much of a step is spent between dependent launches. usage: fwd_gaps.py results.db [steps=8]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
rows = db.execute("select s.kernel_name, d.start, d.end, d.queue_id from %s d join %s s on d.kernel_id = s.id order by d.start" % (
    T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"))).fetchall()
sol = [r[2] for r in rows if "solver_kernel" in r[0]]
t0, t1 = sol[-steps - 1], sol[-1]
sel = [r for r in rows if r[1] >= t0 and r[2] <= t1]
byq = collections.defaultdict(list)
for r in sel:
    byq[r[3]].append(r)
main = max(byq, key=lambda q: sum(b - a for _, a, b, _ in byq[q]))
ks = byq[main]
hist = collections.Counter()
tot = 0.0
small = 0.0
n = 0
for (n0, a0, b0, _), (n1, a1, b1, _) in zip(ks[:-1], ks[1:]):
    g = (a1 - b0) / 1e3
    if g <= 0:
        continue
    tot += g
    if g < 30:
        small += g
        n += 1
        hist[int(g)] += 1
print("main queue %s: %d kernels/step, busy %.2f ms/step, gaps %.2f ms/step of which < 30 us: %.2f ms/step in %d gaps/step (mean %.1f us)" % (
    main, len(ks) // steps, sum(b - a for _, a, b, _ in ks) / steps / 1e6, tot / steps / 1e3, small / steps / 1e3, n // steps, small / max(n, 1)))
print("histogram of small gaps (us: count/step):", " ".join("%d:%d" % (k, v // steps) for k, v in sorted(hist.items())))
# ---- global idle: intervals with no kernel in flight on any queue
ev = sorted(sel, key=lambda r: r[1])
cur_end, last_name = ev[0][2], ev[0][0]
idle = []
for name, a, b, q in ev[1:]:
    if a > cur_end:
        idle.append(((a - cur_end) / 1e3, last_name.split("(")[0][-40:], name.split("(")[0][-40:]))
    if b > cur_end:
        cur_end, last_name = b, name
idle.sort(reverse=True)
print("global idle %.2f ms/step in %d intervals/step; by what follows:" % (sum(g for g, _, _ in idle) / steps / 1e3, len(idle) // steps))
agg = collections.defaultdict(lambda: [0.0, 0])
for g, a, b in idle:
    agg[(a, b)][0] += g
    agg[(a, b)][1] += 1
for (a, b), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  %7.1f us/step  x%-3d after %-40s before %s" % (g / steps, c // steps if c >= steps else c, a, b))
# ---- the kernels around the longest idle interval of the last step
ev2 = [r for r in ev if r[1] >= sol[-2]]
cur_end = ev2[0][2]
best = (0, 0)
for name, a, b, q in ev2[1:]:
    if a > cur_end and a - cur_end > best[0]:
        best = (a - cur_end, cur_end)
    cur_end = max(cur_end, b)
g, at = best
print("longest idle of the last step: %.1f us; kernels from 1.5 ms before to 0.3 ms after it (us relative to its start):" % (g / 1e3))
for name, a, b, q in ev2:
    if b > at - 1500e3 and a < at + g + 300e3:
        print("  q%d %9.1f .. %9.1f  %s" % (q, (a - at) / 1e3, (b - at) / 1e3, name.split("(")[0][-50:]))
