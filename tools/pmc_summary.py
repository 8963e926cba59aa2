#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv passes: mean counter value per dispatch of each strip kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        names = [n for n in ("rgb_walk", "rgb_strip", "rgb_generic", "rgb_unscaled", "rgbin_", "strip_", "column_lds", "generic_kernel") if n in k]
        if not names:
            continue
        k = k[k.index(names[0]):][:44]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = []
for k, cs in acc.items():
    lines.append("kernel %s" % k)
    for c in sorted(cs):
        v = cs[c]
        lines.append("  %-44s %16.0f   (n=%d)" % (c, sum(v) / len(v), len(v)))
txt = "\n".join(lines)
open(os.path.join(out, "summary.txt"), "w").write(txt + "\n")
print(txt)
