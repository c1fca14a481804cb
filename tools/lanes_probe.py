#!/usr/bin/env python3
"""GPU probe: (G lanes per query, U loads in flight) of the exact grid search at a few sizes.
    python tools/lanes_probe.py [ns nt]..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import tile_probe  # noqa: E402

sizes = [(5000, 20000), (65536, 1048576), (262144, 4194304)]
if len(sys.argv) > 2:
    a = [int(x) for x in sys.argv[1:]]
    sizes = list(zip(a[0::2], a[1::2]))
cfgs = [("default", {}, "exact")] + [
    ("G%d-U%d" % (g, u), {"VISMA_ICP_GRID_LANES": str(g + 100 * u)}, "exact")
    for g, u in ((1, 4), (1, 8), (1, 12), (2, 4), (2, 8), (4, 4), (4, 8), (8, 4))]
for ns, nt in sizes:
    tile_probe.timing(ns, nt, 30, cfgs)
