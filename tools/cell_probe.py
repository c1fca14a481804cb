#!/usr/bin/env python3
"""GPU probe: cell edge of the radius-cell grid as a multiple of the search radius.
    python tools/cell_probe.py [ns nt]..."""
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
for ns, nt in sizes:
    for f in ("1.0", "1.2", "1.4", "1.6", "2.0", "2.5", "3.0"):
        os.environ["VISMA_ICP_GRID_CELL"] = f          # read when the grid is planned (at the cloud upload)
        tile_probe.timing(ns, nt, 30, [("cell x%s" % f, {}, "exact")])
