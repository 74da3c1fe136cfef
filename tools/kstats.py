#!/usr/bin/env python3
"""Print a rocprofv3 --stats kernel_stats.csv as ms per step.  usage: kstats.py <dir> <steps_profiled>"""
import csv, glob, re, sys
rows = list(csv.DictReader(open(sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[0])))
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
for r in rows:
    tot += float(r["TotalDurationNs"])
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 14]:
    m = re.search(r"(k_[a-z_]+(?:<[^>]*>)?)", r["Name"]); name = m.group(1) if m else r["Name"][:50]
    print(f"{name:36s} calls {r['Calls']:>5s} {float(r['TotalDurationNs'])/n/1e6:8.3f} ms/step  avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):5.2f}%")
print(f"all kernels: {tot/n/1e6:.3f} ms/step")
