#!/usr/bin/env python3
"""GPU probe: where a wave of the warm-started search spends its cycles (measurement build
-DVISMA_COOP_DEBUG_PHASES, built to a side library).   python tools/coop_phases.py [ns nt]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visma_amd import build  # noqa: E402

SIDE = os.path.join(ROOT, "visma_amd", "lib", "libvisma_icp_phases.so")
if not os.path.exists(SIDE) or "--rebuild" in sys.argv:
    build.build_lib(force=True, defines=("VISMA_COOP_DEBUG_PHASES",), out=SIDE)
os.environ["VISMA_ICP_LIB"] = SIDE
from visma_amd import _lib, synth  # noqa: E402

NAMES = ["src+slot", "bounds+prev, prune", "list written", "chunks", "merge", "f64 winner", "out+moments", "partial row", "fold"]


def main():
    a = [int(x) for x in sys.argv[1:] if x.isdigit()]
    ns, nt = (a + [262144, 4194304])[:2] if len(a) >= 2 else (262144, 4194304)
    src, tgt, T_gt, r = synth.make_pair(ns, nt, motion="radius")
    c = _lib.Context(0)
    c.set_clouds_f64(src, tgt)
    c.set_nn_mode(_lib.NN_GRID)
    c.iterate(np.eye(4), r, 6)
    L = _lib.load()
    out = (C.c_ulonglong * 16)()
    L.visma_debug_coop_phases(out, 1)
    steps = 10
    c.iterate(np.eye(4), r, steps)
    L.visma_debug_coop_phases(out, 0)
    nwg = (ns + 255) // 256 * steps
    tot = sum(out[k] for k in range(9))
    print("ns=%d nt=%d: cycles per workgroup (wave 0) and share; 100 MHz counter" % (ns, nt))
    for k in range(9):
        print("  %-22s %9.1f ticks = %6.2f us  %5.1f%%" % (NAMES[k], out[k] / nwg, out[k] / nwg / 100.0, 100.0 * out[k] / max(tot, 1)))
    print("  total %.2f us" % (tot / nwg / 100.0))


if __name__ == "__main__":
    main()
