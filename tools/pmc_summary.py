"""Summarises rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch, per kernel.
usage: python tools/pmc_summary.py <counter_collection.csv> [...]"""
import csv, sys, collections
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            short = name.split("(")[0].replace("void ", "")[:60]
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", path)
    for k, cs in acc.items():
        if "dcmt" not in k:
            continue
        print(k, " dispatches", len(next(iter(cs.values()))))
        for c, v in cs.items():
            print(f"    {c:24s} mean {sum(v)/len(v):16.1f}")
