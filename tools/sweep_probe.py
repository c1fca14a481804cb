"""Time the 24-yaw orientation-constrained sweep: batched on-device vs sequential."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visma_amd import _lib, synth

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "yaw_sweep.npz"))
ctx = _lib.Context(0)
cases = [("chair 4k->8k r=0.02", g["model"].astype(np.float64), g["scene"].astype(np.float64), 0.02)]
src, tgt, _, _ = synth.make_pair(40000, 20000)
cases.append(("S-surf 40k->20k r=0.02", src, tgt, 0.02))
for name, s, t, r in cases:
    ctx.set_clouds_f64(s, t)
    for label, dev in (("batched device loop", None), ("sequential host loop", False)):
        ctx.set_device_loop(dev)
        ctx.run_yaw_sweep(24, r)
        t0 = time.time()
        best, level, per = ctx.run_yaw_sweep(24, r)
        dt = time.time() - t0
        its = sum(p.iterations for p in per)
        print(json.dumps(dict(case=name, mode=label, ms=dt * 1e3, best_level=level, K=best.num_correspondences,
                              total_iterations=its, icp_iterations_per_s=its / dt)))
    ctx.set_device_loop(None)
