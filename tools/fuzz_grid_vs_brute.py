"""Randomised search for a disagreement between the grid search and the brute-force kernel
(which is pinned to the oracle): clouds on lattices (points exactly on cell faces), duplicates,
planes, lines, extreme radii, elongated boxes, big offsets.  usage: fuzz_grid_vs_brute.py [N] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visma_amd import _lib, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = _lib.Context(0)
ctx.set_search_precision("f32")          # the brute-force kernel is fp32: compare like with like
bad = 0
for it in range(N):
    kind = rng.integers(0, 7)
    nt = int(rng.integers(1, 60000)); ns = int(rng.integers(1, 20000))
    scale = 10.0 ** rng.uniform(-2, 2)
    if kind == 0:      # lattice: many points exactly on cell boundaries of a radius-sized grid
        r = scale * 0.05
        tgt = rng.integers(-40, 40, (nt, 3)) * r * rng.choice([0.5, 1.0, 1.001, 2.0])
        src = rng.integers(-40, 40, (ns, 3)) * r * 0.5
    elif kind == 1:    # duplicates
        base = rng.standard_normal((max(nt // 8, 1), 3)) * scale
        tgt = base[rng.integers(0, len(base), nt)]
        src = base[rng.integers(0, len(base), ns)] + rng.standard_normal((ns, 3)) * scale * 1e-3
        r = scale * 10.0 ** rng.uniform(-3, 0)
    elif kind == 2:    # plane / line
        tgt = rng.standard_normal((nt, 3)) * scale; tgt[:, rng.integers(0, 3)] = 0.0
        if rng.random() < 0.5: tgt[:, rng.integers(0, 3)] = 1.0
        src = rng.standard_normal((ns, 3)) * scale; src[:, 2] *= 1e-3
        r = scale * 10.0 ** rng.uniform(-2.5, 0.5)
    elif kind == 3:    # elongated box (fp32 cell coordinate far from the origin of the grid)
        tgt = rng.random((nt, 3)) * scale * np.array([1000.0, 1.0, 1.0])
        src = rng.random((ns, 3)) * scale * np.array([1000.0, 1.0, 1.0])
        r = scale * 10.0 ** rng.uniform(-2, 0)
    elif kind == 4:    # far from the origin (centring must absorb it)
        off = rng.standard_normal(3) * scale * 1e4
        tgt = rng.standard_normal((nt, 3)) * scale + off
        src = rng.standard_normal((ns, 3)) * scale + off
        r = scale * 10.0 ** rng.uniform(-2, 0)
    elif kind == 5:    # tiny / huge radius
        tgt = rng.standard_normal((nt, 3)) * scale; src = rng.standard_normal((ns, 3)) * scale
        r = scale * 10.0 ** rng.choice([-6, -4, 1, 3])
    else:              # the bench's surface
        src, tgt, _, r = synth.make_pair(ns, max(nt, 8), seed_t=int(rng.integers(1 << 30)), seed_s=int(rng.integers(1 << 30)), motion="radius")
        r *= 10.0 ** rng.uniform(-0.5, 0.5)
    T = synth.make_T(synth.rot_y(rng.uniform(-0.2, 0.2)), rng.standard_normal(3) * r * 0.5)
    ctx.set_clouds_f64(src, tgt)
    res = {}
    for name, mode in (("grid", _lib.NN_GRID), ("brute", _lib.NN_BRUTE)):
        ctx.set_nn_mode(mode)
        ctx.nn_pass(T, r)
        st = ctx.reduce()
        res[name] = (ctx.correspondence_index(), ctx.get_correspondences()[2].view(np.uint32), st, ctx.nn_mode_used())
    same = np.array_equal(res["grid"][0], res["brute"][0]) and np.array_equal(res["grid"][1], res["brute"][1])
    if not same:
        bad += 1
        d = np.flatnonzero(res["grid"][0] != res["brute"][0])
        print("MISMATCH it=%d kind=%d ns=%d nt=%d r=%g: %d indices differ (first %s)" % (it, kind, ns, nt, r, len(d), d[:5]))
print("done: %d configurations, %d mismatches" % (N, bad))
sys.exit(1 if bad else 0)
