"""Summarises rocprofv3 --pmc output: mean counter value per kernel.  python tools/parse_pmc.py DIR [substr]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name", "")
            if sub and sub not in k:
                continue
            key = (k[:90], row.get("Counter_Name"))
            acc[key][0] += float(row.get("Counter_Value", 0))
            acc[key][1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print(f"{c:24s} mean {s / n:16.1f} over {n:4d} dispatches  {k}")
