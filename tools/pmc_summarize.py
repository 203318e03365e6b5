"""Summarise rocprofv3 --pmc CSV output directories written by tools/pmc_profile.sh."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(out, "*", "*", "*counter_collection.csv")):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            if "qlpc_wave4096" not in k:
                continue
            agg[k.split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
summary = {}
for k, ctrs in agg.items():
    summary[k] = {c: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for c, v in sorted(ctrs.items())}
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
for k, ctrs in summary.items():
    print(k)
    for c, v in ctrs.items():
        print(f"  {c:28s} {v['mean_per_dispatch']:18.1f}  (n={v['dispatches']})")
