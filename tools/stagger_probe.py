#!/usr/bin/env python3
"""GPU probe: the warm-started search with part of its workgroups starting late (measurement build
-DVISMA_COOP_STAGGER, built to a side library): does taking the waves out of lock-step shorten the launch?
   python tools/stagger_probe.py [ns nt]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visma_amd import build  # noqa: E402

SIDE = os.path.join(ROOT, "visma_amd", "lib", "libvisma_icp_stagger.so")
if not os.path.exists(SIDE) or "--rebuild" in sys.argv:
    build.build_lib(force=True, defines=("VISMA_COOP_STAGGER",), out=SIDE)
os.environ["VISMA_ICP_LIB"] = SIDE
from visma_amd import _lib, synth  # noqa: E402


def main():
    a = [int(x) for x in sys.argv[1:] if x.isdigit()]
    ns, nt = (a + [262144, 4194304])[:2] if len(a) >= 2 else (262144, 4194304)
    src, tgt, T_gt, r = synth.make_pair(ns, nt, motion="radius")
    c = _lib.Context(0)
    c.set_clouds_f64(src, tgt)
    c.set_nn_mode(_lib.NN_GRID)
    L = _lib.load()
    c.iterate(np.eye(4), r, 6)
    for mode, n in [(0, 0)] + [(m, n) for m in (1, 2, 3, 4, 5, 6) for n in (2, 4, 8)] + [(0, 0)]:
        L.visma_debug_coop_stagger(mode, n)
        c.set_profiling(1)
        c.iterate(np.eye(4), r, 3)
        c.get_timing(reset=True)
        c.iterate(np.eye(4), r, 30)
        tm = c.get_timing(reset=True)
        c.set_profiling(0)
        print(json.dumps({"mode": mode, "sleep_x_8128_cycles": n, "nn_us": tm["nn_ms"] / max(tm["nn_launches"], 1) * 1e3}), flush=True)


if __name__ == "__main__":
    main()
