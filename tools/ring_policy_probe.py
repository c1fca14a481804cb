"""Where the ring search (grid_ring.hip) should take over: registrations and yaw sweeps at the annotation tool's sizes and a
few others, radius cells (mode 0) against rings (mode 1), with the occupancy of the radius-sized cells that the default rule
looks at.  python tools/ring_policy_probe.py"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from visma_amd import _lib, synth  # noqa: E402

CASES = [(5000, 20000, 0.075), (5000, 20000, 0.15), (5000, 20000, 0.05), (16384, 65536, 0.15), (16384, 65536, 0.075),
         (40000, 20000, 0.075), (65536, 262144, 0.05), (65536, 262144, 0.1), (65536, 1048576, 0.03), (262144, 4194304, 0.02)]


def timed(f, reps=5):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


def main():
    for ns, nt, r in CASES:
        src, tgt, T_gt, _ = synth.make_pair(ns, nt, motion="fixed")
        row = {"ns": ns, "nt": nt, "radius": r}
        for mode in (0, 1):
            c = _lib.Context(0)
            c.set_ring_search(mode)
            c.set_nn_mode(_lib.NN_GRID)
            c.set_clouds_f64(src, tgt)
            res = c.run(None, r, 30)
            if mode == 1:
                row["grid"] = c.ring_search()
            row["run30_ms_mode%d" % mode] = round(timed(lambda: c.run(None, r, 30)), 3)
            row["iters_mode%d" % mode] = res.iterations
            if ns <= 65536:
                row["sweep24_ms_mode%d" % mode] = round(timed(lambda: c.run_yaw_sweep(24, r, 30), 3), 3)
            c.close()
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
