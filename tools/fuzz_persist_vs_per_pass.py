#!/usr/bin/env python3
"""Fuzz: the PERSISTENT launch of a host loop against one launch per pass on the hostile configurations of
fuzz_warm_vs_serial.py (lattices on cell faces, duplicates, planes, elongated boxes, far offsets, extreme radii,
disjoint clouds).  Per configuration, on two contexts: a fixed-iteration loop of 9 passes from a random pose near
the truth, a second loop of 4 that carries on, and a whole registration with the stop test (early STOP to the
launch) -- transformation (bit for bit), K, iteration count, per-query winners and squared distances must be equal,
and the persistent launches must really have run.
    python tools/fuzz_persist_vs_per_pass.py [N] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visma_amd import _lib, synth  # noqa: E402


def make_case(rng):
    """the hostile configurations of fuzz_warm_vs_serial.py -> (src, tgt, start pose, radius, kind)"""
    kind = int(rng.integers(0, 7))
    nt = int(rng.integers(1, 60000))
    ns = int(rng.integers(1, 20000))
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
        tgt = rng.standard_normal((nt, 3)) * scale
        tgt[:, rng.integers(0, 3)] = 0.0
        if rng.random() < 0.5:
            tgt[:, rng.integers(0, 3)] = 1.0
        src = rng.standard_normal((ns, 3)) * scale
        src[:, 2] *= 1e-3
        r = scale * 10.0 ** rng.uniform(-2.5, 0.5)
    elif kind == 3:    # elongated box
        tgt = rng.random((nt, 3)) * scale * np.array([1000.0, 1.0, 1.0])
        src = rng.random((ns, 3)) * scale * np.array([1000.0, 1.0, 1.0])
        r = scale * 10.0 ** rng.uniform(-2, 0)
    elif kind == 4:    # far from the origin
        off = rng.standard_normal(3) * scale * 1e4
        tgt = rng.standard_normal((nt, 3)) * scale + off
        src = rng.standard_normal((ns, 3)) * scale + off
        r = scale * 10.0 ** rng.uniform(-2, 0)
    elif kind == 5:    # tiny / huge radius
        tgt = rng.standard_normal((nt, 3)) * scale
        src = rng.standard_normal((ns, 3)) * scale
        r = scale * 10.0 ** rng.choice([-6, -4, 1, 3])             # (a huge radius: every target point a candidate --
                                                                   #  passes of a second each: the launch must wait them out)
        if r > scale:
            ns, src = min(ns, 2000), src[:2000]                    # (... kept short)
    else:              # the bench's surface
        src, tgt, _, r = synth.make_pair(ns, max(nt, 8), seed_t=int(rng.integers(1 << 30)), seed_s=int(rng.integers(1 << 30)),
                                         motion="radius")
        r *= 10.0 ** rng.uniform(-0.5, 0.5)
    T = synth.make_T(synth.rot_y(rng.uniform(-0.2, 0.2)), rng.standard_normal(3) * r * 0.5)
    return src, tgt, T, r, kind


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    a, b = _lib.Context(0), _lib.Context(0)
    a.set_persistent(False)
    b.set_persistent(True)
    b.set_profiling(1)
    bad = loops = launches = passes = 0
    t0 = time.time()
    only = int(os.environ.get("FUZZ_ONLY", "-1"))        # (debugging: run this case alone, the others only drawn)
    for case in range(n):
        src, tgt, T, r, kind = make_case(rng)
        if only >= 0 and case != only:
            continue
        tc = time.time()
        for c in (a, b):
            c.set_nn_mode(_lib.NN_GRID)
            c.set_clouds_f64(src, tgt)
        ok = True
        Ta, Tb = T.copy(), T.copy()
        for steps in (9, 4):
            Ta, ra = a.iterate(Ta, r, steps)
            Tb, rb = b.iterate(Tb, r, steps)
            ok = ok and np.array_equal(Ta, Tb) and ra.num_correspondences == rb.num_correspondences
            for x, y in zip(a.get_correspondences(), b.get_correspondences()):
                ok = ok and np.array_equal(x.view(np.uint32), y.view(np.uint32))
            loops += 1
        for c in (a, b):
            c.forget_winners()
        ra, rb = a.run(T, r, 25, 1e-7, 1e-7), b.run(T, r, 25, 1e-7, 1e-7)
        ok = ok and np.array_equal(ra.transformation_, rb.transformation_) and ra.iterations == rb.iterations \
            and ra.num_correspondences == rb.num_correspondences
        loops += 1
        tm = b.get_timing(reset=True)
        launches += int(tm["persist_launches"])
        passes += int(tm["persist_passes"])
        if tm["persist_aborts"]:
            print("case %d (kind %d): a persistent launch gave up" % (case, kind), flush=True)
            b.set_persistent(True)
        if only >= 0:
            print("case %d kind %d ns=%d nt=%d r=%g: %.2f s, timing %s" % (case, kind, len(src), len(tgt), r, time.time() - tc,
                                                                        {k: v for k, v in tm.items() if k.startswith("persist") or k.startswith("nn_")}))
        if not ok:
            bad += 1
            print("MISMATCH case %d (kind %d): ns=%d nt=%d r=%g" % (case, kind, len(src), len(tgt), r), flush=True)
    if only < 0:
        # passes that last about a second each (a radius that makes every target point a candidate of every query): the
        # launch waits them out -- its workgroups for the publication, the host for the statistics -- however long they take
        r2 = np.random.default_rng(seed + 1)
        tgt = r2.standard_normal((50000, 3))
        src = r2.standard_normal((10000, 3))
        for c in (a, b):
            c.set_clouds_f64(src, tgt)
        tl = time.time()
        Ta, _ = a.iterate(np.eye(4), 300.0, 4)
        Tb, _ = b.iterate(np.eye(4), 300.0, 4)
        tm = b.get_timing(reset=True)
        okl = np.array_equal(Ta, Tb) and tm["persist_aborts"] == 0 and tm["persist_passes"] == 3
        print("long passes: 2 x 4 passes in %.1f s, persistent launch ran %d passes, gave up %d times, equal %s"
              % (time.time() - tl, tm["persist_passes"], tm["persist_aborts"], bool(np.array_equal(Ta, Tb))), flush=True)
        bad += 0 if okl else 1
    print("fuzz_persist_vs_per_pass: %d configurations (seed %d), %d host loops, %d persistent launches running %d passes, "
          "%d mismatches, %.0f s" % (n, seed, loops, launches, passes, bad, time.time() - t0), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
