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
            kn = row.get("Kernel_Name", "")
            if "lsq_" not in kn: continue
            kn = kn.split("lsq_")[1].split("(")[0]
            a = acc[(kn, row["Counter_Name"])]; a[0] += float(row["Counter_Value"]); a[1] += 1
    print("== %s (mean per dispatch)" % os.path.basename(d))
    for (kn, k), (v, n) in sorted(acc.items()):
        print("   %-22s %-28s %18.1f  (n=%d)" % (kn, k, v / max(n, 1), n))
