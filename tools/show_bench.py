"""Prints the numbers of one bench.py JSON line that are looked at while tuning.  usage: python tools/show_bench.py <file.json>"""
import json, sys
d = json.load(open(sys.argv[1]))
print("headline", round(d["value"]), "f/s", round(d["ms_per_step"], 4), "ms  frac", round(d["roofline"]["frac"], 4), "verified", d.get("verified"))
print(" ", d["roofline"].get("kernels"), {k: (round(v["ms"], 4) if isinstance(v, dict) else round(v, 4)) for k, v in d["roofline"]["per_kernel"].items()})
for k, v in d.get("configs", {}).items():
    print(f"  cfg {k:22s} {round(v['value']):8d}  frac {v['roofline']['frac']:.3f}  {v.get('kernels', '')}  {v.get('us_per_frame', '')}", v["roofline"].get("per_kernel_ms", ""), round(v["at_batch_1024"]["value"]) if "at_batch_1024" in v else "")
for k, v in d.get("next_rows", {}).items():
    print(f"  {k:28s} {round(v['value']):8d} {v['unit']:10s} frac {v['roofline']['frac']:.3f}")
if "cpu_baseline" in d: print("  cpu", round(d["cpu_baseline"]["value"], 1), d["cpu_baseline"]["cores"])
