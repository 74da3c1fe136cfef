#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean of each counter over dispatches."""
import csv, glob, sys, collections, re
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        m = re.search(r"(k_[a-z_]+(?:<[^>]*>)?)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    if "k_conv" not in k: continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
