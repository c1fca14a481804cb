#!/usr/bin/env python3
"""GPU probe: ms per ICP iteration of single registrations of several source sizes against the 4 M-point target (the
source shards of 1 / 2 / 4 / 8 ranks and saturated launches), iterations 1..20 from the identity and 41..60 -- run it
under different libraries / environments (VISMA_ICP_LIB, VISMA_ICP_COOP_KERNEL) to compare search kernels.
    python tools/ab_probe.py [nt] [ns ...] [--partial]      (--partial: the whole model against a scan of half of its surface)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visma_amd import _lib, synth  # noqa: E402


def main():
    a = [int(x) for x in sys.argv[1:] if x.isdigit()]
    nt = a[0] if a else 4194304
    sizes = a[1:] or [32768, 65536, 131072, 262144, 1048576]
    tgt = None
    for ns in sizes:
        if "--partial" in sys.argv:
            src, tgt, T_gt, r = synth.make_partial_pair(ns, nt, overlap=0.5)
        else:
            src, t, T_gt, r = synth.make_pair(ns, nt, motion="radius")
            tgt = t if tgt is None else tgt
        c = _lib.Context(0)
        c.set_clouds_f64(src, tgt)
        c.set_nn_mode(_lib.NN_GRID)
        c.iterate(np.eye(4), r, 3)
        first, cont = [], []
        for _ in range(5):
            c.forget_winners()
            t0 = time.perf_counter()
            T, _ = c.iterate(np.eye(4), r, 20)
            first.append(time.perf_counter() - t0)
            T, _ = c.iterate(T, r, 20)
            t0 = time.perf_counter()
            T, _ = c.iterate(T, r, 20)
            cont.append(time.perf_counter() - t0)
        print(json.dumps({"ns": ns, "nt": nt, "lib": os.path.basename(os.environ.get("VISMA_ICP_LIB", "product")),
                          "kernel": c.search_kernel_used(), "env": os.environ.get("VISMA_ICP_COOP_KERNEL", "") + ("cold_in_launch=" + os.environ["VISMA_ICP_COLD_IN_LAUNCH"] if "VISMA_ICP_COLD_IN_LAUNCH" in os.environ else "") + (" partial" if "--partial" in sys.argv else ""),
                          "us_per_iteration_1_20": round(float(np.median(first)) / 20 * 1e6, 2),
                          "us_per_iteration_41_60": round(float(np.median(cont)) / 20 * 1e6, 2)}), flush=True)
        c.close()


if __name__ == "__main__":
    main()
