"""the 24-yaw sweep of config 2 (5 k -> 20 k, r = 0.075, <= 30 iterations: feh::RegisterModelToScene's call), clouds resident:
ms per sweep, median of 9 -- run under VISMA_ICP_LIB / VISMA_ICP_SOLVE_IN_FOLD to compare builds"""
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from visma_amd import _lib, synth
src, tgt, _, _ = synth.make_pair(5000, 20000)
c = _lib.Context(0)
c.set_clouds_f64(src, tgt)
c.run_yaw_sweep(24, 0.075, 30, 1e-6, 1e-6)
ts = []
for _ in range(9):
    t0 = time.perf_counter(); best, which, per = c.run_yaw_sweep(24, 0.075, 30, 1e-6, 1e-6); ts.append(time.perf_counter() - t0)
print("lib=%s sweep_persist=%s solve_in_fold=%s sweep %.3f ms (min %.3f) iterations %d best %d K %d" % (
    os.path.basename(os.environ.get("VISMA_ICP_LIB", "product")), os.environ.get("VISMA_ICP_SWEEP_PERSIST", "1"), os.environ.get("VISMA_ICP_SOLVE_IN_FOLD", "0"),
    np.median(ts) * 1e3, min(ts) * 1e3, sum(p.iterations for p in per), which, best.num_correspondences))
