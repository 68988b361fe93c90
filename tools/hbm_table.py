"""HBM-bound kernels of the training step as GB/s against the 8 TB/s peak (SURVEY section 8d asks for them separately from the
matrix kernels): joins the alone-leg durations of <tag>_kernel_stats.md with the bytes per launch of <tag>_pmc_traffic.json.
usage: python tools/hbm_table.py profiles/r04_c > profiles/r04_c_hbm_kernels.md"""
import json, re, sys
tag = sys.argv[1]
PEAK = 8000.0   # GB/s, /opt/skills/guides/MI355X_MICROARCH.md
MATRIX = ("igemm_kernel", "wino4f_kernel", "wino4t_kernel", "wino4g_kernel", "wino2f_ws_kernel", "wino2f_wgrad_kernel", "stem_fwd_kernel",
          "stem_wgrad_kernel", "gemm3b")
rows, leg = {}, False
for line in open(tag + "_kernel_stats.md"):
    if line.startswith("## "):
        leg = "roofline leg" in line
    m = re.match(r"\| (.+?) \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", line)
    if leg and m:
        rows[m.group(1)] = (int(m.group(2)), float(m.group(3)), float(m.group(4)))
traffic = json.load(open(tag + "_pmc_traffic.json"))["kernels"]
print("# HBM-bound kernels of one training step (DeNet-34 skip, batch 32), %s\n" % tag.split("/")[-1])
print("Durations: the 5 steps of bench.py's roofline leg (every kernel alone on one stream, `%s_kernel_stats.md`); bytes: FETCH_SIZE +\n"
      "WRITE_SIZE per launch from two separate `--pmc` passes of the same command (`%s_pmc_traffic.json`, gfx950 corrections as the\n"
      "guide prescribes). Peak 8 000 GB/s (spec); ~6 300 GB/s is what a pure copy sustains on this chip. Matrix kernels are in\n"
      "`%s_pmc_mfma.json`.\n" % ((tag.split("/")[-1],) * 3))
print("| kernel | launches / step | us / launch | MB / launch (fetch + write) | GB/s | of 8 TB/s | ms / step |")
print("|---|---:|---:|---:|---:|---:|---:|")
out = []
for name, (calls, total_ms, avg_us) in rows.items():
    t = traffic.get(name)
    if any(name.startswith(m) for m in MATRIX) or not t or t.get("write_bytes_per_launch") is None:
        continue
    b = t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"]
    if b < 16e6 or total_ms / 5 < 0.02:
        continue                       # latency-bound launches of a few microseconds: not a bandwidth statement
    out.append((total_ms / 5, name, calls / 5, avg_us, b / 1e6, b / avg_us / 1e3))
tot_ms = tot_b = 0.0
for ms, name, n, us, mb, gbs in sorted(out, reverse=True):
    print("| `%s` | %g | %.1f | %.1f | %.0f | %.2f | %.3f |" % (name, n, us, mb, gbs, gbs / PEAK, ms))
    tot_ms += ms; tot_b += mb * n
print("| **all of the above** | | | %.0f per step | %.0f | %.2f | %.3f |" % (tot_b, tot_b / tot_ms, tot_b / tot_ms / PEAK, tot_ms))
