#!/usr/bin/env python3
"""GPU probe (measurement build -DVISMA_COOP_LB_PROBE): what limits the lower bound LB a search of the certificate kernel
leaves behind -- the runner-up among the examined candidates, or the cells that were not listed -- pass by pass along a C4
registration from the identity: share of the searched queries whose LB is candidate-limited, and by how much the
geometric bound exceeds it there (the room a runner-up-aware certificate could use: DESIGN.md 4.1e).
    python tools/lb_probe.py [ns nt]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visma_amd import build  # noqa: E402

side = os.path.join(build.LIB_DIR, "libvisma_icp_lbprobe.so")
if not os.path.exists(side) or build.is_stale_against(side):
    build.build_lib(force=True, defines=("VISMA_COOP_LB_PROBE",), out=side)
os.environ["VISMA_ICP_LIB"] = side
os.environ["VISMA_ICP_PERSIST"] = "0"           # (one launch per pass: the counters are read between passes)
from visma_amd import _lib, synth  # noqa: E402

a = [int(x) for x in sys.argv[1:] if x.isdigit()]
ns, nt = a[:2] if len(a) >= 2 else (262144, 4194304)
src, tgt, T_gt, r = synth.make_pair(ns, nt, motion="radius")
c = _lib.Context(0)
c.set_nn_mode(_lib.NN_GRID)
c.set_clouds_f64(src, tgt)
c.set_device_loop(False)
c.set_profiling(1)
T = np.eye(4)
c.iterate(T, r, 1)
c.forget_winners()
c.get_timing(reset=True)
print("ns=%d nt=%d radius %.4f mm" % (ns, nt, r * 1e3))
print(" pass | certified | searched with a winner, LB candidate-limited | mean (geometric - candidate bound) there, mm")
for p in range(24):
    T, res = c.iterate(T, r, 1)
    tm = c.get_timing(reset=True)
    nl = max(tm["nn_launches"], 1)
    searched = ns - tm["grid_certified"] / nl
    lim = tm["grid_candidates_27cell"] / nl
    gap = tm["grid_candidates"] / nl / max(lim, 1) * 1e-3
    print(" %4d | %8.4f | %8.4f of the searched | %.3f   (%s)" % (p, tm["grid_certified"] / nl / ns, lim / max(searched, 1), gap, c.search_kernel_used()))
c.close()
