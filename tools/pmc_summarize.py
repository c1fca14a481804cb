#!/usr/bin/env python3
"""Mean counter value per dispatch and kernel from tools/pmc_collect.sh output.
usage: pmc_summarize.py <dir> [kernel-substring]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if sub and sub not in r["Kernel_Name"]:
            continue
        k = (r["Kernel_Name"][:90], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
for (k, c), (s, n) in sorted(acc.items()):
    w.writerow([k, c, n, "%.3f" % (s / n)])
