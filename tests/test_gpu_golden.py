"""GPU: the HIP path against the committed golden fixtures (reference outputs).

These never touch /root/reference: the fixtures are data produced earlier by
tests/golden/gen_golden.py from the compiled reference.  Tolerance: final
SE(3) within 1e-5 relative Frobenius (north_star); K equal; >= 99.99 %
identical correspondences.
"""
import os

import numpy as np
import pytest

from visma_amd import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL_T = 1e-5


def k_matches(ctx, got, want):
    """Correspondence counts: EQUAL with the default (exact) search, which decides near-ties and
    radius cases in f64 like the reference; the fp32-specification kernels may decide a pair at
    the radius differently (a few in 1e5 queries)."""
    if getattr(ctx, "exact", False):
        return int(got) == int(want)
    return abs(int(got) - int(want)) <= max(3, int(0.01 * int(want)))


def tol(ctx):
    """north_star's bar is 1e-5; the exact search reproduces the reference's correspondences, so
    what is left is f64 rounding."""
    return 1e-9 if getattr(ctx, "exact", False) else TOL_T


def load(name):
    return np.load(os.path.join(G, name))


def rel(A, B):
    return synth.rel_frobenius(A, B)


def test_chair_5k_20k_config2(gpu_ctx_any):
    """BASELINE config 2: the 5k -> 20k chair alignment on 1 MI355X vs the CPU reference."""
    g = load("chair_5k_20k.npz")
    gpu_ctx_any.set_clouds_f64(g["src"].astype(np.float64), g["tgt"].astype(np.float64))
    for it in (0, 1, 5, 20):
        r = gpu_ctx_any.run(g["init"], float(g["radius"]), it, 0.0, 0.0)
        row = g["trace"][it]
        assert rel(r.transformation_, row[:16].reshape(4, 4)) < tol(gpu_ctx_any)
        assert r.num_correspondences == row[18]
        assert abs(r.fitness_ - row[16]) < 1e-12 and abs(r.inlier_rmse_ - row[17]) < 1e-6
    assert rel(r.transformation_, g["trace"][20, :16].reshape(4, 4)) < 1e-7   # measured budget ~1e-9
    assert np.mean(gpu_ctx_any.correspondence_index() == g["final_idx"]) >= 0.9999


def test_offset_3m(gpu_ctx_any):
    g = load("chair_offset3m.npz")
    gpu_ctx_any.set_clouds_f64(g["src"].astype(np.float64), g["tgt"].astype(np.float64))
    for row, it in zip(g["trace"], g["trace_iters"]):
        r = gpu_ctx_any.run(g["init"], float(g["radius"]), int(it), 0.0, 0.0)
        assert rel(r.transformation_, row[:16].reshape(4, 4)) < tol(gpu_ctx_any)
        assert r.num_correspondences == row[18]


def test_termination_and_scaling(gpu_ctx_any):
    g = load("chair_5k_20k.npz")
    e = load("estimators.npz")
    gpu_ctx_any.set_clouds_f64(g["src"].astype(np.float64), g["tgt"].astype(np.float64))
    r = gpu_ctx_any.run(None, 0.075, 30, 1e-6, 1e-6)
    assert rel(r.transformation_, e["termination_T"]) < tol(gpu_ctx_any)
    assert k_matches(gpu_ctx_any, r.num_correspondences, e["termination"][2])
    gpu_ctx_any.set_clouds_f64(e["scaled_src"].astype(np.float64), g["tgt"].astype(np.float64))
    r = gpu_ctx_any.run(None, 0.075, 15, 0.0, 0.0, with_scaling=True)
    assert rel(r.transformation_, e["scaled_T"]) < tol(gpu_ctx_any) and r.num_correspondences == e["scaled"][2]


def test_yaw_sweep_orientation_constrained(gpu_ctx_any):
    g = load("yaw_sweep.npz")
    gpu_ctx_any.set_clouds_f64(g["model"].astype(np.float64), g["scene"].astype(np.float64))
    best, level, per = gpu_ctx_any.run_yaw_sweep(int(g["level"]), float(g["radius"]))
    assert level == int(g["best"])
    assert best.num_correspondences == g["k"][level]
    assert rel(best.transformation_, g["T"][level]) < tol(gpu_ctx_any)
    ks = np.array([p.num_correspondences for p in per])
    assert all(k_matches(gpu_ctx_any, a, b) for a, b in zip(ks, g["k"]))


def test_fragments_p2p_and_p2plane(gpu_ctx_any):
    g = load("fragments.npz")
    gpu_ctx_any.set_clouds_f64(g["src"].astype(np.float64), g["tgt"].astype(np.float64))
    r = gpu_ctx_any.run(g["init"], float(g["radius"]), 10, 0.0, 0.0)
    assert rel(r.transformation_, g["trace_p2p"][10][:16].reshape(4, 4)) < tol(gpu_ctx_any)
    assert r.num_correspondences == g["trace_p2p"][10][18]
    gpu_ctx_any.set_target_normals_f64(g["tgt_normals"].astype(np.float64))
    for it in (1, 10):
        r = gpu_ctx_any.run_point_to_plane(g["init"], float(g["radius"]), it, 0.0, 0.0)
        row = g["trace_p2plane"][it]
        assert rel(r.transformation_, row[:16].reshape(4, 4)) < tol(gpu_ctx_any)
        assert k_matches(gpu_ctx_any, r.num_correspondences, row[18])
        # inlier_rmse is the nearest-neighbour distance, not the plane residual (Registration.cpp:65-68,93)
        assert abs(r.inlier_rmse_ - row[17]) < (1e-9 if gpu_ctx_any.search_mode_used() == "exact" else 1e-4) * row[17]


def test_edge_cases(gpu_ctx_any):
    g = load("edge_cases.npz")
    src, tgt, dup = (g[k].astype(np.float64) for k in ("src", "tgt", "tgt_dup"))
    cases = {"none": (src + 50.0, tgt), "tiny_radius": (src, tgt), "dup": (src, dup),
             "one_src": (src[:1], tgt), "one_tgt": (src, tgt[:1]), "huge_radius": (src, tgt),
             "zero_iter": (src, tgt)}
    for name, (s, t) in cases.items():
        r_, m = g[name + "_args"]
        gpu_ctx_any.set_clouds_f64(s, t)
        r = gpu_ctx_any.run(None, float(r_), int(m), 0.0, 0.0)
        assert r.num_correspondences == g[name + "_frk"][2], name
        assert abs(r.fitness_ - g[name + "_frk"][0]) < 1e-12, name
        assert abs(r.inlier_rmse_ - g[name + "_frk"][1]) < 1e-6 * max(1.0, g[name + "_frk"][1]), name
        if name not in ("one_tgt", "one_src"):
            assert rel(r.transformation_, g[name + "_T"]) < tol(gpu_ctx_any), name
    gpu_ctx_any.set_clouds_f64(src, tgt)
    r = gpu_ctx_any.run(g["bad_radius_T"], 0.0, 5)
    assert np.array_equal(r.transformation_, g["bad_radius_T"]) and r.num_correspondences == 0
    r = gpu_ctx_any.run_point_to_plane(g["plane_without_normals_T"], 0.05, 5)   # no normals uploaded
    assert np.array_equal(r.transformation_, g["plane_without_normals_T"])
    # EvaluateRegistration == one NN pass at T
    gpu_ctx_any.nn_pass(g["evaluate_T"], 0.05)
    st = gpu_ctx_any.reduce()
    assert int(round(st[0])) == g["evaluate_frk"][2]
    assert np.array_equal(gpu_ctx_any.correspondence_index(), g["evaluate_idx"])
    assert abs(np.sqrt(st[1] / st[0]) - g["evaluate_frk"][1]) < 1e-6


def test_known_answer_nn_distances(gpu_ctx_any):
    """O3D/UnitTest/Core/Geometry/PointCloud.cpp:1074-1111 through the NN kernel."""
    g = load("open3d_known_answers.npz")
    p = g["rand_points"]
    gpu_ctx_any.set_clouds_f64(p[:50], p[50:100])
    gpu_ctx_any.nn_pass(np.eye(4), 1e4)
    gpu_ctx_any.reduce()
    si, ti, d2 = gpu_ctx_any.get_correspondences()
    assert len(si) == 50
    assert np.abs(np.sqrt(d2.astype(np.float64)) - g["nn_distance_ref"]).max() < 1e-3   # fp32 at |x| ~ 1e3
    q = p[50:100][ti]
    assert np.abs(np.linalg.norm(p[:50] - q, axis=1) - g["nn_distance_ref"]).max() < 1e-6


def test_known_answer_transform(gpu_ctx_any):
    """O3D/UnitTest/Core/Geometry/PointCloud.cpp:172-232: the fused transform (w row ignored)."""
    g = load("open3d_known_answers.npz")
    p = g["rand_points"][:10]
    T = g["transform_T"].copy()
    expected = g["transform_ref_points"]
    # target = the expected transformed points: every source point must find ITS image at ~0
    gpu_ctx_any.set_clouds_f64(p, expected)
    gpu_ctx_any.nn_pass(T, 1.0)
    gpu_ctx_any.reduce()
    idx = gpu_ctx_any.correspondence_index()
    assert np.array_equal(idx, np.arange(10))
    _, _, d2 = gpu_ctx_any.get_correspondences()
    assert np.sqrt(d2.max()) < 1e-3


def test_so3_device_math(lib, oracle):
    """Device rodrigues/invrodrigues (visma_amd/csrc/so3.h) vs the oracle restatement."""
    import ctypes as C
    L = lib.load()
    if not hasattr(L, "visma_icp_selftest_so3"):
        pytest.skip("selftest entry not exported")
    g = load("rodrigues.npz")
    w = np.ascontiguousarray(g["w"], np.float64)
    n = len(w)
    R = np.empty((n, 9)); w2 = np.empty((n, 3))
    dp = C.POINTER(C.c_double)
    rc = L.visma_icp_selftest_so3(w.ctypes.data_as(dp), R.ctypes.data_as(dp), w2.ctypes.data_as(dp), n)
    assert rc == 0
    assert np.abs(R.reshape(n, 3, 3) - g["R"]).max() < 1e-13
    assert np.abs(w2 - g["w_back"]).max() < 1e-9
