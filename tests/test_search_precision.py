"""Double-precision search (include/visma_icp.h: visma_icp_set_search_precision).

The fp32 search decides near-ties and radius-boundary cases differently from the reference's
f64 KD-tree about once in 1e5 queries; for clouds of a few thousand points ONE flipped pair
moves the final transform past the 1e-5 tolerance (tools/fuzz_icp_vs_oracle.py: 5 of 300 random
registrations, worst 6.1e-5).  In f64 mode the correspondences must BE the f64 oracle's
(the restatement of Registration.cpp:41-96 on FLANN's f64 L2), pass after pass.
"""
import os

import numpy as np
import pytest

from visma_amd import _lib, synth

pytestmark = pytest.mark.gpu


def hard_case(seed):
    """Configurations like the ones the fuzz tool flags: small cloud, large radius, off-origin."""
    rng = np.random.default_rng(seed)
    ns, nt = int(rng.integers(2000, 7000)), int(rng.integers(15000, 40000))
    src, tgt, T_gt, r = synth.make_pair(ns, nt, seed_t=int(rng.integers(1 << 30)), seed_s=int(rng.integers(1 << 30)),
                                        noise=10.0 ** rng.uniform(-4, -2.5), motion="radius")
    r *= rng.uniform(1.5, 3.0)
    off = rng.standard_normal(3) * 10.0
    init = synth.make_T(synth.rot_y(rng.uniform(-0.02, 0.02)), rng.standard_normal(3) * r * 0.3)
    return src + off, tgt + off, init, r


@pytest.mark.parametrize("seed", [3, 4])
def test_f64_search_returns_the_f64_oracles_correspondences(lib, oracle, seed):
    src, tgt, init, r = hard_case(seed)
    ctx = _lib.Context(0)
    ctx.set_search_precision("f64")
    ctx.set_clouds_f64(src, tgt)
    T = init
    for _ in range(4):                                   # a few passes along an ICP trajectory
        ctx.nn_pass(T, r)
        st = ctx.reduce()
        assert ctx.search_is_f64()
        k, idx, d2, e2 = oracle.nn_pass(oracle.transform_points(src, T), tgt, r)
        assert np.array_equal(ctx.correspondence_index(), idx)
        assert int(st[0]) == k
        assert abs(st[1] - e2) <= 1e-12 * max(e2, 1e-30)
        T = np.array(_lib.solve_from_stats(st)) @ T
    # the fp32 search of the same context differs somewhere on such data or not -- but never by much
    ctx.set_search_precision("f32")
    ctx.set_clouds_f64(src, tgt)
    ctx.nn_pass(init, r)
    ctx.reduce()
    assert not ctx.search_is_f64()
    k, idx, _, _ = oracle.nn_pass(oracle.transform_points(src, init), tgt, r)
    assert np.mean(ctx.correspondence_index() == idx) > 0.999


def test_f64_search_closes_the_parity_gap_of_small_clouds(lib, oracle):
    """The three worst configurations of tools/fuzz_icp_vs_oracle.py 300 1 (fp32: 2.0e-5, 1.3e-5, 6.1e-5)."""
    rng = np.random.default_rng(1)
    worst = {}
    for it in range(122):
        ns = int(rng.integers(500, 8000)); nt = int(rng.integers(2000, 40000))
        st, ss = int(rng.integers(1 << 30)), int(rng.integers(1 << 30))
        noise = 10.0 ** rng.uniform(-4, -2.5)
        rmul = rng.uniform(0.7, 3.0)
        off = rng.standard_normal(3) * rng.choice([0.0, 1.0, 10.0])
        ang = rng.uniform(-0.02, 0.02); tv = rng.standard_normal(3)
        iters = int(rng.integers(1, 40))
        if it in (2, 6, 121):
            worst[it] = (ns, nt, st, ss, noise, rmul, off, ang, tv, iters)
    ctx = _lib.Context(0)
    errs = {}
    for it, (ns, nt, st, ss, noise, rmul, off, ang, tv, iters) in worst.items():
        src, tgt, T_gt, r = synth.make_pair(ns, nt, seed_t=st, seed_s=ss, noise=noise, motion="radius")
        r *= rmul
        src, tgt = src + off, tgt + off
        init = synth.make_T(synth.rot_y(ang), tv * r * 0.3)
        want = oracle.registration_icp(src, tgt, r, init=init, max_iter=iters)
        for mode in ("f32", "f64"):
            ctx.set_search_precision(mode)
            ctx.set_clouds_f64(src, tgt)
            got = ctx.run(init, r, iters, 1e-6, 1e-6)
            errs[(it, mode)] = synth.rel_frobenius(got.transformation_, want.T)
            assert got.num_correspondences == want.k
    for it in worst:
        assert errs[(it, "f32")] > 1e-5                  # the gap is real (and inherent to an fp32 search)
        assert errs[(it, "f64")] < 1e-10, errs


def test_exact_search_is_the_default_at_every_size_and_in_sweeps(lib, oracle):
    """No size-keyed policy any more: the default search is the exact one (fp32 ranking, f64
    re-rank of the candidates inside the rounding band) and returns what the f64 search returns."""
    src, tgt, T_gt, r = synth.make_pair(3000, 9000, motion="radius")
    ctx = _lib.Context(0)
    ctx.set_clouds_f64(src, tgt)
    best, lvl, per = ctx.run_yaw_sweep(8, r * 2)           # the batched device loop runs it too
    assert ctx.search_mode_used() == "exact"
    ref = _lib.Context(0)
    ref.set_search_precision("f64")
    ref.set_clouds_f64(src, tgt)
    ref.set_device_loop(False)
    b2, l2, p2 = ref.run_yaw_sweep(8, r * 2)
    assert ref.search_mode_used() == "f64"
    assert lvl == l2
    for a, b in zip(per, p2):
        assert a.num_correspondences == b.num_correspondences and a.iterations == b.iterations
        assert synth.rel_frobenius(a.transformation_, b.transformation_) < 1e-12
    big_s, big_t, _, rr = synth.make_pair(140000, 200000, motion="radius")
    ctx.set_clouds_f64(big_s, big_t)                       # round 1 switched to fp32 above 131,072 sources
    a = ctx.run(None, rr, 3, 0, 0)
    assert ctx.search_mode_used() == "exact"
    ref.set_clouds_f64(big_s, big_t)
    b = ref.run(None, rr, 3, 0, 0)
    assert a.num_correspondences == b.num_correspondences
    assert synth.rel_frobenius(a.transformation_, b.transformation_) < 1e-12
    assert np.array_equal(ctx.correspondence_index(), ref.correspondence_index())
    ctx.set_nn_mode(_lib.NN_BRUTE)                         # the brute-force kernels have an exact flavour too
    ctx.set_clouds_f64(src, tgt)
    ctx.run(None, r, 2, 0, 0)
    assert ctx.search_mode_used() == "exact"
    ctx.set_search_precision("f32")                        # ... and the fp32 specification on request
    ctx.set_clouds_f64(src, tgt)
    ctx.run(None, r, 2, 0, 0)
    assert ctx.search_mode_used() == "f32"


def test_batched_problems_use_the_f64_search_too(lib, oracle):
    """Config 3 (many small ICPs in flight) is exactly where one flipped pair shows."""
    probs, want = [], []
    for seed in (11, 12, 13):
        src, tgt, init, r = hard_case(seed)
        src, tgt = src[:2500], tgt[:12000]
        probs.append((src, tgt, init, r))
        want.append(oracle.registration_icp(src, tgt, r, init=init, max_iter=15))
    ctx = _lib.Context(0)
    got = ctx.run_batch(probs, max_iter=15)
    for g, w in zip(got, want):
        assert g.num_correspondences == w.k
        assert synth.rel_frobenius(g.transformation_, w.T) < 1e-10


def test_small_cloud_with_a_huge_radius_stays_in_f64(lib, oracle):
    """A radius comparable to the cloud gives a degenerate grid (a few cells); for small f64 clouds the
    f64 grid search is kept anyway (the fp32 brute-force kernel would be the AUTO choice otherwise)."""
    src, tgt, T_gt, r = synth.make_pair(1500, 4000, motion="radius")
    r = 0.6                                               # the surface is about 1 m across
    ctx = _lib.Context(0)
    ctx.set_clouds_f64(src, tgt)
    init = synth.make_T(synth.rot_y(0.05), [0.01, 0.0, -0.02])
    ctx.nn_pass(init, r)
    st = ctx.reduce()
    assert ctx.search_is_f64() and ctx.nn_mode_used() == _lib.NN_GRID
    k, idx, d2, e2 = oracle.nn_pass(oracle.transform_points(src, init), tgt, r)
    assert np.array_equal(ctx.correspondence_index(), idx) and int(st[0]) == k
    got = ctx.run(init, r, 10, 0, 0)
    want = oracle.registration_icp(src, tgt, r, init=init, max_iter=10, rel_fitness=0.0, rel_rmse=0.0)
    assert synth.rel_frobenius(got.transformation_, want.T) < 1e-10


def test_reference_fixtures_under_the_default_precision(lib):
    """The committed outputs of the compiled reference (tests/golden/*.npz), with a default context:
    small f64 clouds -> f64 search -> the reference's correspondences and iteration counts, every
    yaw of the orientation sweep included (the fp32 search is allowed a few pairs of slack there)."""
    import os
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ctx = _lib.Context(0)
    g = np.load(os.path.join(G, "chair_5k_20k.npz"))
    ctx.set_clouds_f64(g["src"].astype(np.float64), g["tgt"].astype(np.float64))
    for it in (0, 1, 5, 20):
        r = ctx.run(g["init"], float(g["radius"]), it, 0.0, 0.0)
        row = g["trace"][it]
        assert ctx.search_is_f64()
        assert synth.rel_frobenius(r.transformation_, row[:16].reshape(4, 4)) < 1e-12
        assert r.num_correspondences == row[18] and abs(r.inlier_rmse_ - row[17]) < 1e-13
    assert np.array_equal(ctx.correspondence_index(), g["final_idx"])
    g = np.load(os.path.join(G, "chair_offset3m.npz"))
    ctx.set_clouds_f64(g["src"].astype(np.float64), g["tgt"].astype(np.float64))
    for row, it in zip(g["trace"], g["trace_iters"]):
        r = ctx.run(g["init"], float(g["radius"]), int(it), 0.0, 0.0)
        assert synth.rel_frobenius(r.transformation_, row[:16].reshape(4, 4)) < 1e-11      # 3 m from the origin
        assert r.num_correspondences == row[18]
    g = np.load(os.path.join(G, "yaw_sweep.npz"))
    ctx.set_clouds_f64(g["model"].astype(np.float64), g["scene"].astype(np.float64))
    best, level, per = ctx.run_yaw_sweep(int(g["level"]), float(g["radius"]))
    assert level == int(g["best"])
    for p, T, k in zip(per, g["T"], g["k"]):
        assert p.num_correspondences == k
        if k > 50:                                        # (a yaw that matched almost nothing has no defined rotation)
            assert synth.rel_frobenius(p.transformation_, T) < 1e-9


def test_point_to_plane_under_the_default_precision(lib):
    """O3D's point-to-plane estimator on the reference's fragment pair: f64 search + f64 coordinates in
    the statistics, normals included, against the compiled reference's trace."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fragments.npz"))
    ctx = _lib.Context(0)
    ctx.set_clouds_f64(g["src"].astype(np.float64), g["tgt"].astype(np.float64))
    r = ctx.run(g["init"], float(g["radius"]), 10, 0.0, 0.0)
    assert ctx.search_is_f64()
    assert synth.rel_frobenius(r.transformation_, g["trace_p2p"][10][:16].reshape(4, 4)) < 1e-11
    assert r.num_correspondences == g["trace_p2p"][10][18]
    ctx.set_target_normals_f64(g["tgt_normals"].astype(np.float64))
    for it in (1, 10):
        r = ctx.run_point_to_plane(g["init"], float(g["radius"]), it, 0.0, 0.0)
        row = g["trace_p2plane"][it]
        assert r.num_correspondences == row[18]
        assert synth.rel_frobenius(r.transformation_, row[:16].reshape(4, 4)) < 1e-10    # f64 normals too


def test_point_to_plane_sweep_on_the_device_equals_single_runs(lib):
    """The batched device loop with the plane estimator (visma_icp_run_yaw_sweep_point_to_plane)."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fragments.npz"))
    src, tgt = g["src"].astype(np.float64), g["tgt"].astype(np.float64)
    r = float(g["radius"])
    ctx = _lib.Context(0)
    ctx.set_clouds_f64(src, tgt)
    ctx.set_target_normals_f64(g["tgt_normals"].astype(np.float64))
    best, level, per = ctx.run_yaw_sweep_point_to_plane(6, r, 8)
    one = _lib.Context(0)
    one.set_device_loop(False)
    one.set_clouds_f64(src, tgt)
    one.set_target_normals_f64(g["tgt_normals"].astype(np.float64))
    ks = []
    for i in range(6):
        w = one.run_point_to_plane(synth.make_T(synth.rot_y(2 * np.pi * i / 6), [0, 0, 0]), r, 8)
        assert w.num_correspondences == per[i].num_correspondences and w.iterations == per[i].iterations
        assert synth.rel_frobenius(per[i].transformation_, w.transformation_) < 1e-9
        ks.append(w.num_correspondences)
    assert level == int(np.argmax(ks))
    # the reference trace of the identity start (fragments.npz is an output of oracle/_ref)
    w = ctx.run_point_to_plane(g["init"], r, 10, 0.0, 0.0)
    assert synth.rel_frobenius(w.transformation_, g["trace_p2plane"][10][:16].reshape(4, 4)) < 1e-9
