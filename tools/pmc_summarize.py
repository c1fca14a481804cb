#!/usr/bin/env python3
"""Mean counter value per dispatch and kernel from tools/pmc_collect.sh output; for every kernel also over its LAST 20
dispatches alone (a registration's converged passes: what bench.py's `value` times).
usage: pmc_summarize.py <dir> [kernel-substring]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
LAST = 20
rows = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if sub and sub not in r["Kernel_Name"]:
            continue
        rows[(f, r["Kernel_Name"][:90], r["Counter_Name"])].append((int(r.get("Dispatch_Id", 0) or 0), float(r["Counter_Value"])))
acc = collections.defaultdict(lambda: [0.0, 0])
for (f, k, c), v in rows.items():
    v.sort()
    for _, x in v:
        acc[(k, c)][0] += x; acc[(k, c)][1] += 1
    if len(v) > LAST:
        for _, x in v[-LAST:]:
            acc[(k + " [last %d dispatches]" % LAST, c)][0] += x; acc[(k + " [last %d dispatches]" % LAST, c)][1] += 1
    if "persist" in k:
        # a persistent launch runs a whole host loop: the LAST dispatch is the loop of the last 20 passes
        # (tools/run_c4_iterations.py), the regime `value` is timed in
        acc[(k + " [last dispatch = one host loop]", c)][0] += v[-1][1]; acc[(k + " [last dispatch = one host loop]", c)][1] += 1
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
for (k, c), (s, n) in sorted(acc.items()):
    w.writerow([k, c, n, "%.3f" % (s / n)])
