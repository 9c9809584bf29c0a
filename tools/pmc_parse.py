#!/usr/bin/env python3
"""Merge rocprofv3 --pmc counter_collection CSVs (one directory per counter group) into {kernel: {counter: {n, mean}}}."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

work, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(f"{work}/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
        a = acc[name][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
res = {k: {c: {"n": n, "mean": s / n} for c, (n, s) in v.items()} for k, v in acc.items()}
json.dump(res, open(out, "w"), indent=1)
print(f"{len(res)} kernels -> {out}")
