"""What a caller pays for a registration in the large-radius regime (grid_ring.hip): clouds in, transform out, on a context
whose buffers exist.  python tools/ring_e2e.py [ns nt radius iters]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from visma_amd import _lib, synth  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 4194304
r = float(sys.argv[3]) if len(sys.argv) > 3 else 0.15
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 30
src, tgt, T_gt, _ = synth.make_pair(ns, nt, motion="fixed")
c = _lib.Context(0)
c.set_nn_mode(_lib.NN_GRID)
c.set_clouds_f64(src, tgt)
c.run(None, r, iters, 0.0, 0.0)
up, first, run = [], [], []
for _ in range(5):
    t0 = time.perf_counter()
    c.set_clouds_f64(src, tgt)
    t1 = time.perf_counter()
    c.iterate(None, r, 1)
    t2 = time.perf_counter()
    res = c.run(None, r, iters, 0.0, 0.0)
    t3 = time.perf_counter()
    up.append(t1 - t0); first.append(t2 - t1); run.append(t3 - t2)
print(json.dumps({"ns": ns, "nt": nt, "radius": r, "grid": c.ring_search(), "kernel": c.search_kernel_used(),
                  "upload_ms": round(float(np.median(up)) * 1e3, 3), "grid_build_and_first_pass_ms": round(float(np.median(first)) * 1e3, 3),
                  "run_%d_iterations_ms" % iters: round(float(np.median(run)) * 1e3, 3), "fitness": res.fitness_,
                  "err_vs_T_gt": float(np.abs(res.transformation_ - T_gt).max())}))
