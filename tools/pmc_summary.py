"""Aggregates rocprofv3 counter_collection CSVs (one directory per --pmc pass) per kernel:
mean counter value per dispatch."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(os.path.join(root, "pass*", "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "anonymous namespace" in name:
            name = name.split("::")[2].split("(")[0]
        else:
            name = name.split("(")[0]
        a = acc[name][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
out = {}
for k, cs in acc.items():
    out[k] = {c: v[0] / v[1] for c, v in cs.items()}
    out[k]["dispatches_seen"] = max(v[1] for v in cs.values())
json.dump(out, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
for k in sorted(out):
    if not (k.startswith("k_synth") or k.startswith("k_fft") or k.startswith("k_sort") or k.startswith("k_sel") or k.startswith("k_recomb")):
        continue
    print(k)
    for c, v in sorted(out[k].items()):
        print(f"   {c:28s} {v:16.1f}")
