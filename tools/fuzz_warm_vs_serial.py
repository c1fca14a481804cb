"""Randomised search for a disagreement between the warm-started wave-cooperative search (grid_coop.hip: every pass
after the first) and the lane-serial kernel, both exact: clouds on lattices (points exactly on cell faces),
duplicates (exact ties everywhere: the whole-wave re-scan), planes, lines, extreme radii, elongated boxes, big
offsets; passes after no motion at all, small and decaying motions (the certificate: previous winner provably
unchanged, no search) and jumps larger than the radius (previous winner useless).
Indices, distances and the 38 statistics must be equal BIT for bit.   usage: fuzz_warm_vs_serial.py [N] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
from visma_amd import _lib, synth  # noqa: E402

os.environ.pop("VISMA_ICP_COOP", None)
os.environ.pop("VISMA_ICP_GRID_LANES", None)
warm = _lib.Context(0)                                   # default policy: lane-serial first pass, then warm-started
os.environ["VISMA_ICP_COOP"] = "0"
os.environ["VISMA_ICP_GRID_LANES"] = "801"
serial = _lib.Context(0)                                 # the lane-serial kernel, one query per lane, every pass
os.environ.pop("VISMA_ICP_COOP", None)
os.environ.pop("VISMA_ICP_GRID_LANES", None)
ring = _lib.Context(0)                                   # the ring search over cells smaller than the radius (grid_ring.hip),
ring.set_ring_search(1)                                  # wherever a finer table than the radius-sized one exists
for c in (warm, serial, ring):
    c.set_nn_mode(_lib.NN_GRID)
for c in (warm, serial):
    c.set_ring_search(0)                                 # (these two are about the radius-cell kernels)
warm.set_profiling(1)
queries = 0
bad = 0
ring_passes = ring_bad = 0
used = {}
for it in range(N):
    kind = int(rng.integers(0, 7))
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
        if rng.random() < 0.5:
            tgt[:, rng.integers(0, 3)] = 1.0
        src = rng.standard_normal((ns, 3)) * scale; src[:, 2] *= 1e-3
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
        tgt = rng.standard_normal((nt, 3)) * scale; src = rng.standard_normal((ns, 3)) * scale
        r = scale * 10.0 ** rng.choice([-6, -4, 1, 3])
    else:              # the bench's surface
        src, tgt, _, r = synth.make_pair(ns, max(nt, 8), seed_t=int(rng.integers(1 << 30)), seed_s=int(rng.integers(1 << 30)), motion="radius")
        r *= 10.0 ** rng.uniform(-0.5, 0.5)
    for c in (warm, serial, ring):
        c.set_clouds_f64(src, tgt)
    T = synth.make_T(synth.rot_y(rng.uniform(-0.2, 0.2)), rng.standard_normal(3) * r * 0.5)
    # in radii; pass 0 is the first (lane-serial on both); the decaying tail is what ICP does -- where the certificate
    # of grid_coop.hip (previous winner provably unchanged: no search) decides most queries
    motions = [0.0, 0.02, 0.3, 0.0, 3.0, 0.1, 0.03, 0.01, 0.003, 0.0, 1e-4, 0.05]
    for p, m in enumerate([None] + motions):
        if m is not None:
            T = synth.make_T(synth.rot_y(rng.uniform(-1, 1) * min(m, 1.0) * 0.05), rng.standard_normal(3) * r * m) @ T
        out = []
        for c in (warm, serial):
            c.nn_pass(T, r)
            st = c.reduce()
            out.append((c.correspondence_index(), c.get_correspondences()[2].view(np.uint32), st.view(np.uint64)))
        # the ring search against the lane-serial kernel: indices and distances bit for bit, statistics to rounding
        # (another summation order)
        ring.nn_pass(T, r)
        rst = ring.reduce()
        ridx, rd2 = ring.correspondence_index(), ring.get_correspondences()[2].view(np.uint32)
        if ring.search_kernel_used() == "ring":
            ring_passes += 1
            sst = out[1][2].view(np.float64)
            tol = 1e-10 * np.maximum(np.abs(sst), np.abs(sst).max() * 1e-3 + 1e-300)
            if not (np.array_equal(ridx, out[1][0]) and np.array_equal(rd2, out[1][1]) and rst[0] == sst[0] and
                    np.all(np.abs(rst - sst) <= tol)):
                ring_bad += 1
                d = np.flatnonzero(ridx != out[1][0])
                print("RING MISMATCH it=%d kind=%d pass=%d ns=%d nt=%d r=%g: %d indices differ (first %s), grid %s" %
                      (it, kind, p, ns, nt, r, len(d), d[:5], ring.ring_search()), flush=True)
        queries += ns if p > 0 else 0
        k = warm.search_kernel_used()
        used[k] = used.get(k, 0) + 1
        same = np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
        if p > 0:
            same = same and np.array_equal(out[0][2], out[1][2])   # (pass 0 of `warm` may use several lanes per query)
            if k != "warm":
                print("NOTE it=%d pass %d ran kernel %s" % (it, p, k))
        if not same:
            bad += 1
            d = np.flatnonzero(out[0][0] != out[1][0])
            print("MISMATCH it=%d kind=%d pass=%d ns=%d nt=%d r=%g: %d indices differ (first %s), stats equal %s" %
                  (it, kind, p, ns, nt, r, len(d), d[:5], np.array_equal(out[0][2], out[1][2])), flush=True)
cert = warm.get_timing()["grid_certified"]
print("done: %d configurations x 13 passes, kernels used %s, %.1f %% of the warm passes' queries certified (no search), %d mismatches"
      % (N, used, 100.0 * cert / max(queries, 1), bad))
print("ring search (grid_ring.hip, forced wherever a finer table exists): %d passes, %d ring mismatches" % (ring_passes, ring_bad))
sys.exit(1 if bad or ring_bad else 0)
