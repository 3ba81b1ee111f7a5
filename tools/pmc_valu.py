"""VALU instruction counts and issue cycles per bench step from one rocprofv3 --pmc pass
(SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES ..., collected by tools/collect_profiles.sh: counters in
their own pass, nothing but --pmc).

  python tools/pmc_valu.py <sq1 counter_collection.csv> [out.json]

Writes profiles/valu_latest.json: per kernel of the 1024-frame step, the mean per dispatch of SQ_INSTS_VALU (wave-level VALU
instructions) and SQ_ACTIVE_INST_VALU (quad-cycles in which a SIMD issues VALU; x4 = cycles).  bench.py reports them as
roofline.valu: issue time = active x 4 / (1024 SIMDs x 2.4 GHz) against the live kernel time."""
import collections, csv, json, os, sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if "dcmt::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"kernels": {}, "units": "mean per dispatch; SQ_ACTIVE_INST_VALU in quad-cycles summed over all SIMDs"}
for k, cs in sorted(acc.items()):
    if not ("k_pre_s" in k or "k_pre_p" in k or "k_fp_s" in k or "k_fp_q" in k or "k_fp_p" in k):
        continue        # the redo launches return at once
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    out["kernels"][k] = {"insts_valu": m.get("SQ_INSTS_VALU"), "active_inst_valu": m.get("SQ_ACTIVE_INST_VALU"),
                         "wave_cycles": m.get("SQ_WAVE_CYCLES"), "busy_cycles": m.get("SQ_BUSY_CYCLES"),
                         "wait_inst_any": m.get("SQ_WAIT_INST_ANY"), "wait_any": m.get("SQ_WAIT_ANY"), "dispatches": len(next(iter(cs.values())))}
path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "valu_latest.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
