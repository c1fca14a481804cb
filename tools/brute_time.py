#!/usr/bin/env python3
"""GPU probe: kernel times of the brute-force path at C4 (NN kernel, exact reduction)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visma_amd import _lib, synth
src, tgt, T_gt, r = synth.make_pair(262144, 4194304, motion="radius")
c = _lib.Context(0); c.set_nn_mode(_lib.NN_BRUTE); c.set_clouds_f64(src, tgt)
c.set_profiling(1)
T, _ = c.iterate(np.eye(4), r, 1); c.get_timing(reset=True)
T, last = c.iterate(T, r, 2)
tm = c.get_timing(reset=True)
print(c.search_mode_used(), "nn_ms", tm["nn_ms"]/max(tm["nn_launches"],1), "reduce_ms", tm["reduce_ms"]/max(tm["reduce_launches"],1), "K", last.num_correspondences)
