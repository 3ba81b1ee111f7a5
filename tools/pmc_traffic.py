"""HBM traffic per bench step from two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE), collected
exactly as /opt/skills/guides/MI355X_MICROARCH.md prescribes: separate passes (FETCH_SIZE takes 3
of the 4 TCC slots, WRITE_SIZE 2), units of KiB, and on gfx950 FETCH_SIZE counts 128-B read
requests at 64 B, i.e. exactly half the bytes of a wide coalesced stream -> doubled here.

  python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps+warmup> [out.json]

Writes profiles/traffic_latest.json ({"hbm_bytes_per_step": ...}) which bench.py reports as roofline.traffic."""
import collections, csv, json, os, sys

def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter and "dcmt::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return acc

fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
# launches per bench step: k_fp_s runs exactly once per step, so its dispatch count is the number of steps the pass saw;
# every other kernel's launches per step = its dispatches / that -- except k_pre_s, which bench.py also runs alone for its
# live per-kernel split (per_kernel_ms): one launch per step by construction.  (argv[3], the step count, is only checked.)
steps_seen = max(len(v) for k, v in fetch.items() if "k_fp_s" in k)
if int(sys.argv[3]) != steps_seen:
    print(f"note: {steps_seen} k_fp_s dispatches seen, {sys.argv[3]} steps + warmup given", file=sys.stderr)
out = {"kernels": {}, "fetch_correction": "x2 (gfx950: FETCH_SIZE = TCC_EA0_RDREQ x 64 B for 128-B requests)"}
tot = 0.0
for k in sorted(set(fetch) | set(write)):
    fv, wv = fetch.get(k, [0.0]), write.get(k, [0.0])
    per_step = 1.0 if "k_pre_s" in k else len(fv) / steps_seen
    fb = 2.0 * 1024.0 * (sum(fv) / len(fv)) * per_step
    wb = 1024.0 * (sum(wv) / len(wv)) * per_step
    out["kernels"][k] = {"read_bytes_per_step": fb, "write_bytes_per_step": wb, "launches_per_step": per_step}
    tot += fb + wb
out["hbm_bytes_per_step"] = tot
path = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic_latest.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
