"""prints the main fields of a bench.py JSON line: python tools/exp/show_bench.py <file>"""
import json
import sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "roofline.frac", d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"])
for k, v in d.items():
    if k not in ("config", "roofline", "cpu_baseline"):
        print(k, json.dumps(v)[:260])
for k, v in d["config"].items():
    print("config." + k, json.dumps(v)[:200])
for k, v in d["roofline"].items():
    print("roofline." + k, json.dumps(v)[:200])
