#!/usr/bin/env python3
"""GPU probe of the wave-cooperative exact search (grid_coop.hip, lanes code 9901): (1) correspondences and
the 38 statistics bit-identical to the all-f64 search on random passes (several sizes, dense / degenerate /
offset clouds, duplicated points); (2) timing against the lane-serial kernel.
    python tools/coop_probe.py [--quick] [--timing-only]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import tile_probe  # noqa: E402
from tile_probe import ctx_with, rand_T  # noqa: E402
from visma_amd import synth  # noqa: E402

VARIANTS = [("auto", {}),                                        # first pass lane-serial, then warm-started
            ("coop-always", {"VISMA_ICP_GRID_LANES": "9901"}),  # also the cold pass (radius pruning)
            ("serial", {"VISMA_ICP_COOP": "0", "VISMA_ICP_GRID_LANES": "801"})]


def check(ns, nt, npass, seed, radius=None, offset=None, label="", dup=0):
    rng = np.random.default_rng(seed)
    src, tgt, T_gt, r = synth.make_pair(ns, nt, seed_t=seed, seed_s=seed + 7, offset=offset, motion="radius")
    if radius is not None:
        r = radius
    if dup:                                            # every point `dup` more times: exact ties everywhere
        tgt = np.concatenate([tgt] + [tgt[rng.permutation(len(tgt))[: len(tgt) // 2]] for _ in range(dup)])
    ref = ctx_with({"VISMA_ICP_COOP": "0", "VISMA_ICP_GRID_LANES": "801"}, "f64", src, tgt)   # same summation tree
    cs = [(n, ctx_with(e, "exact", src, tgt)) for n, e in VARIANTS]
    bad = 0
    for p in range(npass):
        T = T_gt @ rand_T(rng, r * 0.8, r * 0.5) if p else np.eye(4)
        ref.nn_pass(T, r)
        st0 = ref.reduce()
        i0 = ref.correspondence_index()
        d0 = ref.get_correspondences()[2]
        for n, c in cs:
            c.nn_pass(T, r)
            st = c.reduce()
            i1 = c.correspondence_index()
            d1 = c.get_correspondences()[2]
            nd = int((i0 != i1).sum())
            same = np.array_equal(st.view(np.uint64), st0.view(np.uint64)) and np.array_equal(d0.view(np.uint32), d1.view(np.uint32))
            if n == "auto" and p == 0 and not same:
                # the first pass of `auto` runs the lane-serial kernel with the lanes-per-query policy of the cloud size
                # (several lanes per query for small clouds): another summation tree than the reference context's 801
                # -- same correspondences and distances, statistics equal to rounding
                same = np.array_equal(d0.view(np.uint32), d1.view(np.uint32)) and bool(np.all(np.abs(st - st0) <= 1e-11 * (np.abs(st0) + np.abs(st0).max())))
            if nd or not same:
                bad += 1
                rel = float(np.max(np.abs(st - st0) / (np.abs(st0) + 1e-300 + 1e-12 * np.abs(st0).max())))
                print("MISMATCH %s %s pass %d: idx diff %d, K %d vs %d, stats rel %.3e" % (label, n, p, nd, st[0], st0[0], rel), flush=True)
    for n, c in cs:
        assert c.search_mode_used() == "exact", c.search_mode_used()
        c.close()
    ref.close()
    print("check %s ns=%d nt=%d r=%.4g passes=%d K=%d: %s" % (label, ns, len(tgt), r, npass, int(st0[0]), "OK" if bad == 0 else "BAD(%d)" % bad), flush=True)
    return bad


def main():
    bad = 0
    if "--timing-only" not in sys.argv:
        bad += check(5000, 20000, 4, 11, label="5k-20k")
        bad += check(3000, 8000, 4, 12, radius=0.075, label="3k-8k big radius")
        bad += check(2000, 500, 3, 13, radius=0.2, label="2k-500 degenerate")
        bad += check(20000, 100000, 3, 14, offset=[3.0, -2.0, 1.0], label="offset 3m")
        bad += check(4000, 30000, 3, 17, dup=3, label="duplicated points")
        bad += check(300, 7, 2, 18, radius=0.5, label="7 targets")
        bad += check(70000, 60000, 2, 19, radius=0.05, label="dense rows (window overflow)")
        if "--quick" not in sys.argv:
            bad += check(65536, 1048576, 2, 15, label="64k-1M")
            bad += check(262144, 4194304, 2, 16, label="C4")
            bad += check(1000000, 4194304, 1, 20, label="1M-4M (several queries per lane)")
    cfgs = [(n, e, "exact") for n, e in VARIANTS]
    for ns, nt, steps in ((5000, 20000, 40), (65536, 1048576, 30), (262144, 4194304, 30), (1048576, 4194304, 10)):
        tile_probe.timing(ns, nt, steps, cfgs)
    print("TOTAL MISMATCHES", bad)


if __name__ == "__main__":
    main()
