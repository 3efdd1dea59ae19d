#!/usr/bin/env python3
"""Condenses rocprofv3 csv output of tools/prof.sh: kernel stats + per-dispatch mean of each counter for the count kernel."""
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats", os.path.relpath(f, out))
    for i, row in enumerate(csv.reader(open(f))):
        if i < 8: print("  ", ",".join(row))
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(d): continue
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "lsq_count_kernel" not in row.get("Kernel_Name", ""): continue
            a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    print("== %s (mean per lsq_count_kernel dispatch)" % os.path.basename(d))
    for k, (v, n) in sorted(acc.items()):
        print("   %-28s %18.1f  (n=%d)" % (k, v / max(n, 1), n))
