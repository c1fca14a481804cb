"""Large-radius regime (grid_ring.hip): us per ICP iteration of a registration from the identity under SURVEY 8d's literal
ground truth, ring search vs radius cells, and what a pass examines.  python tools/ring_probe.py [ns nt radius iters]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from visma_amd import _lib, synth  # noqa: E402


def main():
    ns = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    nt = int(sys.argv[2]) if len(sys.argv) > 2 else 4194304
    r = float(sys.argv[3]) if len(sys.argv) > 3 else 0.15
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    modes = [int(m) for m in (sys.argv[5].split(",") if len(sys.argv) > 5 else ["1", "0"])]
    lib = _lib
    src, tgt, T_gt, _ = synth.make_pair(ns, nt, motion="fixed")
    for mode in modes:
        c = _lib.Context(0)
        c.set_ring_search(mode)
        c.set_nn_mode(lib.NN_GRID)
        t0 = time.perf_counter()
        c.set_clouds_f64(src, tgt)
        T, res = c.iterate(None, r, 1)                      # (grid build + first pass)
        t_first = time.perf_counter() - t0
        info = c.ring_search()
        # per-pass times, profiled
        c.set_profiling(1)
        per, cand, rows, rmse = [], [], [], []
        T = np.eye(4)
        for k in range(iters):
            c.get_timing(True)
            T, res = c.iterate(T, r, 1)
            tm = c.get_timing(True)
            per.append(round(tm.get("nn_ms", 0.0) * 1e3, 1))
            cand.append(round(tm.get("grid_candidates", 0.0) / ns, 1))
            rows.append(round(tm.get("grid_candidates_27cell", 0.0) / ns, 1))
            rmse.append(round(res.inlier_rmse_ / max(info["cell"], 1e-30), 3))
        c.set_profiling(0)
        reps = []
        for _ in range(3):
            t0 = time.perf_counter()
            T2, res2 = c.iterate(None, r, iters)
            reps.append((time.perf_counter() - t0) / iters * 1e6)
        print(json.dumps({"ns": ns, "nt": nt, "radius": r, "ring_mode": mode, "grid": info, "kernel": c.search_kernel_used(),
                          "setup_and_first_pass_ms": round(t_first * 1e3, 2), "us_per_iteration": [round(x, 1) for x in reps],
                          "kernel_us_by_pass": per, "candidates_per_query_by_pass": cand, "rows_per_query_by_pass": rows, "rmse_in_cells_by_pass": rmse, "fitness": res2.fitness_, "rmse": res2.inlier_rmse_,
                          "err_vs_T_gt": float(np.abs(T2 - T_gt).max())}), flush=True)
        c.close()


if __name__ == "__main__":
    main()
