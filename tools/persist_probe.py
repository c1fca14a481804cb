#!/usr/bin/env python3
"""GPU probe: the PERSISTENT launch of the certificate kernel (one launch per host loop, the next transform handed over
through mapped host memory) against one launch per pass: the same registration on two contexts, transforms and
correspondences compared bit for bit, microseconds per iteration of both.
    python tools/persist_probe.py [nt] [ns ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visma_amd import _lib, synth  # noqa: E402


def context(persist, src, tgt):
    os.environ["VISMA_ICP_PERSIST"] = "1" if persist else "0"      # (read when the context is created)
    c = _lib.Context(0)
    c.set_clouds_f64(src, tgt)
    c.set_nn_mode(_lib.NN_GRID)
    return c


def main():
    a = [int(x) for x in sys.argv[1:] if x.isdigit()]
    nt = a[0] if a else 4194304
    sizes = a[1:] or [5000, 65536, 262144]
    for ns in sizes:
        src, tgt, T_gt, r = synth.make_pair(ns, nt, motion="radius")
        out = {"ns": ns, "nt": nt}
        res = {}
        for persist in (0, 1):
            c = context(persist, src, tgt)
            c.iterate(np.eye(4), r, 3)
            c.forget_winners()
            Ts = []
            T = np.eye(4)
            first, cont = [], []
            t0 = time.perf_counter()
            T, _ = c.iterate(T, r, 20)
            first.append(time.perf_counter() - t0)
            Ts.append(T.copy())
            for _ in range(7):
                t0 = time.perf_counter()
                T, last = c.iterate(T, r, 20)
                cont.append(time.perf_counter() - t0)
                Ts.append(T.copy())
            si, ti, d2 = c.get_correspondences()
            # a complete registration through visma_icp_run (stop test on the host)
            c.forget_winners()
            t0 = time.perf_counter()
            rr = c.run(np.eye(4), r, 30)
            t_run = time.perf_counter() - t0
            c.set_profiling(1)
            c.get_timing(reset=True)
            T2, _ = c.iterate(T, r, 20)
            tm = c.get_timing(reset=True)
            c.set_profiling(0)
            res[persist] = (Ts, si, ti, d2, rr)
            out["persist" if persist else "per_pass"] = {
                "us_per_iteration_1_20": round(first[0] / 20 * 1e6, 2),
                "us_per_iteration_cont_median": round(float(np.median(cont)) / 20 * 1e6, 2),
                "us_per_iteration_cont_min": round(float(np.min(cont)) / 20 * 1e6, 2),
                "run30_ms": round(t_run * 1e3, 3), "run_iterations": rr.iterations,
                "nn_ms_per_pass": tm["nn_ms"] / max(tm["nn_launches"], 1),
                "persist_launches": tm["persist_launches"], "persist_passes": tm["persist_passes"],
                "certified": tm["grid_certified"] / max(tm["nn_launches"], 1) / ns}
            c.close()
        same_T = all(np.array_equal(x, y) for x, y in zip(res[0][0], res[1][0]))
        same_c = all(np.array_equal(res[0][k], res[1][k]) for k in (1, 2, 3))
        same_run = np.array_equal(res[0][4].transformation_, res[1][4].transformation_) and \
            res[0][4].iterations == res[1][4].iterations
        out.update(transforms_identical=bool(same_T), correspondences_identical=bool(same_c), run_identical=bool(same_run))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
