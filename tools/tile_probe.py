#!/usr/bin/env python3
"""GPU probe of the streamed (LDS-tile) search: (1) identical correspondences / statistics to the
f64 search of round 1 on random passes, tile and fallback paths; (2) timing at C4 per tile config.
    python tools/tile_probe.py [--quick]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visma_amd import _lib, synth  # noqa: E402


def ctx_with(env, prec, src, tgt):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        c = _lib.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    c.set_search_precision(prec)
    c.set_clouds_f64(src, tgt)
    c.set_nn_mode(_lib.NN_GRID)
    return c


def rand_T(rng, ang, tr):
    a = rng.normal(size=3)
    a /= np.linalg.norm(a)
    th = rng.uniform(0, ang)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = rng.normal(size=3) * tr
    return T


def check(ns, nt, npass, seed, radius=None, offset=None, label=""):
    rng = np.random.default_rng(seed)
    src, tgt, T_gt, r = synth.make_pair(ns, nt, seed_t=seed, seed_s=seed + 7, offset=offset, motion="radius")
    if radius is not None:
        r = radius
    ref = ctx_with({}, "f64", src, tgt)
    res = {}
    bad = 0
    variants = [("exact", {}), ("exact-2launch", {"VISMA_ICP_FUSED_FOLD": "0"}),
                ("exact-g4", {"VISMA_ICP_GRID_LANES": "804"}), ("exact-g2", {"VISMA_ICP_GRID_LANES": "802"})]
    if os.environ.get("PROBE_TILE"):
        variants = [("tile%d" % k, {"VISMA_ICP_TILE": "1", "VISMA_ICP_TILE_CONFIG": str(k)}) for k in (0, 1, 2, 3, 4)]
    cs = [(n, ctx_with(e, "exact", src, tgt)) for n, e in variants]
    for p in range(npass):
        T = T_gt @ rand_T(rng, r * 0.8, r * 0.5) if p else np.eye(4)
        ref.nn_pass(T, r)
        st0 = ref.reduce()
        i0 = ref.correspondence_index()
        for n, c in cs:
            c.nn_pass(T, r)
            st = c.reduce()
            i1 = c.correspondence_index()
            nd = int((i0 != i1).sum())
            rel = float(np.max(np.abs(st - st0) / (np.abs(st0) + 1e-300 + 1e-12 * np.abs(st0).max())))
            ok = nd == 0 and rel < 1e-10 and st[0] == st0[0]
            if not ok:
                bad += 1
                print("MISMATCH %s %s pass %d: idx diff %d, K %d vs %d, stats rel %.3e" % (label, n, p, nd, st[0], st0[0], rel), flush=True)
            res.setdefault(n, []).append((nd, rel))
    for n, c in cs:
        assert c.search_mode_used() == "exact", c.search_mode_used()
    assert ref.search_mode_used() == "f64"
    print("check %s ns=%d nt=%d r=%.4g passes=%d K=%d: %s" % (
        label, ns, nt, r, npass, int(st0[0]), "OK" if bad == 0 else "BAD(%d)" % bad), flush=True)
    return bad


def timing(ns, nt, steps, cfgs):
    src, tgt, T_gt, r = synth.make_pair(ns, nt, motion="radius")
    out = []
    for name, env, prec in cfgs:
        c = ctx_with(env, prec, src, tgt)
        c.set_profiling(1)
        T, _ = c.iterate(np.eye(4), r, 3)
        c.get_timing(reset=True)
        T, last = c.iterate(np.eye(4), r, steps)
        tm = c.get_timing(reset=True)
        c.set_profiling(0)
        c.iterate(np.eye(4), r, 3)
        t0 = time.perf_counter()
        T2, last = c.iterate(np.eye(4), r, steps)
        dt = time.perf_counter() - t0
        nl = max(tm["nn_launches"], 1)
        row = {"cfg": name, "ns": ns, "nt": nt, "it_per_s": steps / dt, "us_per_iter": dt / steps * 1e6,
               "nn_us": tm["nn_ms"] / nl * 1e3, "fold_us": tm["reduce_ms"] / max(tm["reduce_launches"], 1) * 1e3,
               "cand_per_q": tm["grid_candidates"] / nl / ns, "tile_wg": tm["tile_workgroups"] / nl,
               "fallback_wg": tm["tile_fallback_workgroups"] / nl, "tile_pts_per_wg": tm["tile_points"] / max(tm["tile_workgroups"], 1),
               "tile_points": tm["tile_points"] / nl,
               "rows_per_wg": tm["tile_rows"] / max(tm["tile_workgroups"], 1), "reranks": tm["f64_reranks"] / nl,
               "parts_per_wg": tm["tile_parts"] / max(tm["tile_workgroups"], 1),
               "mode": c.search_mode_used(), "K": last.num_correspondences,
               "phase_cyc_per_wg": [round(x / max(tm["tile_workgroups"], 1)) for x in tm["tile_phase_cycles"]],
               "err_vs_gt": synth.rel_frobenius(T2, T_gt)}
        print(json.dumps(row), flush=True)
        out.append(row)
        c.close()
    return out


def main():
    quick = "--quick" in sys.argv
    bad = 0
    if "--tile" in sys.argv:
        os.environ["PROBE_TILE"] = "1"
        bad += check(5000, 20000, 4, 11, label="5k-20k")
        bad += check(3000, 8000, 4, 12, radius=0.075, label="3k-8k big radius")
        bad += check(2000, 500, 3, 13, radius=0.2, label="2k-500 degenerate")
        bad += check(20000, 100000, 3, 14, offset=[3.0, -2.0, 1.0], label="offset 3m")
        bad += check(65536, 1048576, 2, 15, label="64k-1M")
        bad += check(262144, 4194304, 2, 16, label="C4")
        cfgs = [("exact-gather", {}, "exact")] + [
            ("tile%d" % k, {"VISMA_ICP_TILE": "1", "VISMA_ICP_TILE_CONFIG": str(k)}, "exact") for k in (0, 1, 2, 3, 4, 10)]
        timing(5000, 20000, 40, cfgs)
        timing(65536, 1048576, 30, cfgs)
        timing(262144, 4194304, 30, cfgs)
        print("TOTAL MISMATCHES", bad)
        return
    if "--hyb" in sys.argv:
        bad += check(5000, 20000, 4, 11, label="5k-20k")
        bad += check(3000, 8000, 4, 12, radius=0.075, label="3k-8k big radius")
        bad += check(2000, 500, 3, 13, radius=0.2, label="2k-500 degenerate")
        bad += check(20000, 100000, 3, 14, offset=[3.0, -2.0, 1.0], label="offset 3m")
        bad += check(65536, 1048576, 2, 15, label="64k-1M")
        bad += check(262144, 4194304, 2, 16, label="C4")
        cfgs = [("legacy-f32", {}, "f32"), ("legacy-f64", {}, "f64"), ("exact", {}, "exact"),
                ("exact-2launch", {"VISMA_ICP_FUSED_FOLD": "0"}, "exact"),
                ("f32-2launch", {"VISMA_ICP_FUSED_FOLD": "0"}, "f32")]
        timing(5000, 20000, 40, cfgs)
        timing(65536, 1048576, 30, cfgs)
        timing(262144, 4194304, 30, cfgs)
        print("TOTAL MISMATCHES", bad)
        return
    if "--time-only" in sys.argv:
        cfgs = [("legacy-f32", {"VISMA_ICP_TILE": "0"}, "f32")] + [
            ("tile%d" % k, {"VISMA_ICP_TILE_CONFIG": str(k)}, "exact") for k in (0, 1, 2, 3, 8, 9, 10, 11)] + [
            ("tile0-nofold", {"VISMA_ICP_TILE_CONFIG": "0", "VISMA_ICP_TILE_FOLD": "0"}, "exact"),
            ("fallback", {"VISMA_ICP_TILE_FALLBACK": "1"}, "exact")]
        timing(5000, 20000, 40, cfgs)
        timing(262144, 4194304, 30, cfgs)
        return
    bad += check(5000, 20000, 4, 11, label="5k-20k")
    bad += check(3000, 8000, 4, 12, radius=0.075, label="3k-8k big radius")
    bad += check(2000, 500, 3, 13, radius=0.2, label="2k-500 degenerate")
    bad += check(20000, 100000, 3, 14, offset=[3.0, -2.0, 1.0], label="offset 3m")
    bad += check(65536, 1048576, 2, 15, label="64k-1M")
    if not quick:
        bad += check(262144, 4194304, 2, 16, label="C4")
    cfgs = [("legacy-f32", {"VISMA_ICP_TILE": "0"}, "f32"), ("legacy-f64", {"VISMA_ICP_TILE": "0"}, "f64"),
            ] + [("tile%d" % k, {"VISMA_ICP_TILE_CONFIG": str(k)}, "exact") for k in range(5)] + [
            ("fallback", {"VISMA_ICP_TILE_FALLBACK": "1"}, "exact")]
    timing(5000, 20000, 40, cfgs)
    timing(65536, 1048576, 30, cfgs)
    timing(262144, 4194304, 30, cfgs)
    print("TOTAL MISMATCHES", bad)


if __name__ == "__main__":
    main()
