"""Mean of each PMC counter per kernel from rocprofv3 --pmc counter_collection CSVs under a directory tree.

    python tools/pmc_summary.py gpurun_out/pmc_x [substring filters...]  -> JSON on stdout
"""
import csv
import glob
import json
import sys
from collections import defaultdict

root, filters = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if filters and not any(s in name for s in filters):
            continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v), 1) for c, v in d.items()} | {"dispatches": max(len(v) for v in d.values())}
       for k, d in acc.items()}
print(json.dumps(out, indent=1, sort_keys=True))
