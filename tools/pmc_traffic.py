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
nsteps = int(sys.argv[3])
out = {"kernels": {}, "fetch_correction": "x2 (gfx950: FETCH_SIZE = TCC_EA0_RDREQ x 64 B for 128-B requests)"}
tot = 0.0
for k in sorted(set(fetch) | set(write)):
    fb = 2.0 * 1024.0 * sum(fetch.get(k, [0.0])) / nsteps
    wb = 1024.0 * sum(write.get(k, [0.0])) / nsteps
    out["kernels"][k] = {"read_bytes_per_step": fb, "write_bytes_per_step": wb}
    tot += fb + wb
out["hbm_bytes_per_step"] = tot
path = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic_latest.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
