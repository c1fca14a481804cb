import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from visma_amd import _lib, synth
for ns, nt in ((262144, 4194304), (65536, 1048576), (5000, 20000)):
    src, tgt, T_gt, r = synth.make_pair(ns, nt, motion="radius")
    for name, env in (("serial-first", {}), ("coop-cold", {"VISMA_ICP_GRID_LANES": "9901"})):
        for k, v in env.items(): os.environ[k] = v
        c = _lib.Context(0)
        for k in env: os.environ.pop(k)
        c.set_clouds_f64(src, tgt); c.set_nn_mode(_lib.NN_GRID)
        c.iterate(np.eye(4), r, 2)
        c.set_profiling(1)
        for T0 in (np.eye(4), T_gt):
            c.get_timing(reset=True)
            for _ in range(10):
                c.forget_winners()
                c.iterate(T0, r, 1)     # 2 passes? iterate(steps=1) = 1 pass + solve
            tm = c.get_timing(reset=True)
            print(ns, name, "identity" if T0 is not T_gt else "converged", "nn_us/launch %.1f" % (tm["nn_ms"] / max(tm["nn_launches"], 1) * 1e3), tm["nn_launches"], c.search_kernel_used(), flush=True)
        c.close()
