"""us per pass of the 24-yaw sweep (5 k -> 20 k, fixed iterations): slope between 10 and 40 iterations"""
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from visma_amd import _lib, synth
src, tgt, _, _ = synth.make_pair(5000, 20000)
c = _lib.Context(0)
c.set_clouds_f64(src, tgt)
def t(k):
    c.run_yaw_sweep(24, 0.075, k, 0.0, 0.0)
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); c.run_yaw_sweep(24, 0.075, k, 0.0, 0.0); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6
a, b = t(10), t(40)
print("sweep_persist=%s: 10 iterations %.0f us, 40 iterations %.0f us, slope %.2f us per pass" % (os.environ.get("VISMA_ICP_SWEEP_PERSIST", "1"), a, b, (b - a) / 30))
