"""MFMA utilisation per kernel from one rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE).
  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter
  is summed over the 8 XCDs). Calibration: a v_mfma_f32_32x32x2_f32 occupies its SIMD for 64 cycles, so BUSY = 64 x
  (executed FLOP / 4096) - checked on the component GEMMs (9.66 GFLOP -> 1.51e8).
usage: python tools/pmc_mfma.py <pmc.db> > profiles/rNN_pmc_mfma.json"""
import collections
import json
import sqlite3
import subprocess
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
q = """select s.kernel_name, d.dispatch_id, d.end - d.start, p.symbol, sum(e.value) from %s e join %s p on e.pmc_id = p.id
       join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id group by d.dispatch_id, p.symbol""" % (
    T("rocpd_pmc_event"), T("rocpd_info_pmc"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"))
agg = {}
for name, did, dur, sym, val in db.execute(q):
    a = agg.setdefault(name, {"ids": set(), "dur": 0.0, "c": collections.defaultdict(float)})
    if did not in a["ids"]:
        a["ids"].add(did)
        a["dur"] += dur
    a["c"][sym] += val


def demangle(n):
    n = n[:-3] if n.endswith(".kd") else n
    d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
    return (d[5:] if d.startswith("void ") else d).split("(")[0]


out = {}
for name, a in agg.items():
    c, n = a["c"], len(a["ids"])
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if busy <= 0 or cyc <= 0:
        continue
    wave = c.get("SQ_WAVE_CYCLES", 0.0)
    out[demangle(name)] = {
        "launches": n, "avg_us_in_this_pass": round(a["dur"] / n / 1e3, 2),
        "mfma_busy_cycles_per_launch": round(busy / n), "kernel_cycles_per_launch": round(cyc / n),
        "clock_ghz_in_this_pass": round(cyc / a["dur"], 3),
        "mfma_util": round(busy / (1024.0 * cyc), 4),
        "executed_tflops_at_this_clock": round(busy / 64.0 * 4096.0 / a["dur"] / 1e3, 2),
        "wave_time_split": {k: round(c.get(s, 0.0) / wave, 3) for k, s in (("waiting_for_issue", "SQ_WAIT_INST_ANY"),
                                                                            ("parked_waitcnt_barrier", "SQ_WAIT_ANY"),
                                                                            ("issuing", "SQ_ACTIVE_INST_ANY"))} if wave else None}
print(json.dumps({"note": "one --pmc pass (kernels serialised by the profiler); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs "
                          "x GRBM_GUI_ACTIVE / 8)", "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"] * kv[1]["launches"]))}, indent=1))
