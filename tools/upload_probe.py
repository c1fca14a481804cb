#!/usr/bin/env python3
"""GPU probe: visma_icp_set_clouds_f64 on a warm context at a few sizes (VISMA_ICP_RAW_UPLOAD_MIN=huge: host packing)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visma_amd import _lib, synth
for ns, nt in ((5000, 20000), (20000, 100000), (65536, 262144), (65536, 1048576), (262144, 4194304)):
    src, tgt, _, r = synth.make_pair(ns, nt, motion="radius")
    c = _lib.Context(0)
    for _ in range(3):
        c.set_clouds_f64(src, tgt)
    t = []
    for _ in range(8):
        t0 = time.perf_counter(); c.set_clouds_f64(src, tgt); t.append(time.perf_counter() - t0)
    print(ns, nt, "set_clouds_f64 ms: median %.3f min %.3f" % (np.median(t) * 1e3, min(t) * 1e3), flush=True)
    c.close()
