"""GPU parity tests for the HIP kernels, through the C ABI (pytest -m gpu).

The checker is the CPU oracle (oracle/icp_oracle.c): correspondences and fp32
distances must match its kernel specification BIT FOR BIT; the f64 statistics
to summation-order accuracy; the final SE(3) within 1e-5 relative Frobenius
of the f64 reference algorithm (north_star tolerance).
"""
import os
import numpy as np
import pytest

from visma_amd import synth

pytestmark = pytest.mark.gpu

TOL_T = 1e-5  # north_star: final SE(3) within 1e-5 relative Frobenius of the CPU reference


def _rand_T(rng, ang=0.2, tr=0.1):
    w = rng.standard_normal(3)
    w *= ang / np.linalg.norm(w)
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K
    return synth.make_T(R, rng.standard_normal(3) * tr)


def _clouds(rng, ns, nt, spread=1.0):
    tgt = (rng.random((nt, 3)) * 2 - 1) * spread
    src = (rng.random((ns, 3)) * 2 - 1) * spread
    return src.astype(np.float32), tgt.astype(np.float32)


@pytest.mark.parametrize("ns,nt,radius", [
    (1000, 777, 0.3),       # ragged, single chunk tail
    (5000, 20000, 0.075),   # C1/C2 shape
    (257, 513, 0.5),        # just past tile / chunk boundaries
    (70000, 3000, 0.1),     # large-source path (8 points per thread)
    (1, 1, 10.0),
    (3, 100000, 0.05),
])
def test_nn_bit_exact(gpu_ctx, oracle, ns, nt, radius):
    rng = np.random.default_rng(ns * 31 + nt)
    src, tgt = _clouds(rng, ns, nt)
    T = _rand_T(rng, 0.1, 0.05)
    gpu_ctx.set_target(tgt)
    gpu_ctx.set_source(src)
    gpu_ctx.nn_pass(T, radius)
    st = gpu_ctx.reduce()
    idx = gpu_ctx.correspondence_index()
    si, ti, d2 = gpu_ctx.get_correspondences()
    T32 = T[:3, :].astype(np.float32)
    r2f = np.float32(radius * radius)
    k, oidx, od2 = oracle.k_nn_pass(src, tgt, T32, r2f, grid=(ns * nt > 5e7))
    assert np.array_equal(idx, oidx)
    assert k == len(si) == int(round(st[0]))
    assert np.array_equal(d2.view(np.uint32), od2[oidx >= 0].view(np.uint32))
    ost = oracle.k_reduce_stats(src, tgt, oidx, T[:3, :])
    scale = np.maximum(np.abs(ost), 1.0)
    assert np.max(np.abs(st - ost) / scale) < 1e-9


def test_nn_ties_lowest_index_and_strict_radius(gpu_ctx, oracle):
    # duplicated targets -> exact ties; a target exactly at distance r -> rejected
    tgt = np.array([[0, 0, 0], [1, 0, 0], [0, 0, 0], [1, 0, 0], [0.5, 0, 0]] * 200, np.float32)
    src = np.array([[0.01, 0, 0], [0.99, 0, 0], [0.25, 0, 0], [5, 5, 5]], np.float32)
    gpu_ctx.set_target(tgt)
    gpu_ctx.set_source(src)
    gpu_ctx.nn_pass(np.eye(4), 0.25)
    gpu_ctx.reduce()
    idx = gpu_ctx.correspondence_index()
    _, oidx, _ = oracle.k_nn_pass(src, tgt, np.eye(4)[:3].astype(np.float32), np.float32(0.0625))
    assert np.array_equal(idx, oidx)
    assert idx[0] == 0 and idx[1] == 1 and idx[3] == -1
    # |0.25 - 0.5|^2 == 0.0625 == r^2 -> strict '<' rejects both neighbours of src[2]
    assert idx[2] == -1


def test_no_correspondences_and_bad_radius(gpu_ctx):
    rng = np.random.default_rng(5)
    src, tgt = _clouds(rng, 500, 800)
    gpu_ctx.set_clouds_f64(src.astype(np.float64) + 100.0, tgt.astype(np.float64))
    r = gpu_ctx.run(None, 0.01, 5, 0, 0)
    assert r.num_correspondences == 0 and r.fitness_ == 0.0 and r.inlier_rmse_ == 0.0
    assert np.allclose(r.transformation_, np.eye(4), atol=1e-12)   # identity updates
    init = _rand_T(rng)
    r = gpu_ctx.run(init, 0.0, 5)                                   # Registration.cpp:148-151
    assert np.array_equal(r.transformation_, init) and r.iterations == 0


def test_empty_clouds(gpu_ctx):
    rng = np.random.default_rng(6)
    src, tgt = _clouds(rng, 10, 10)
    gpu_ctx.set_clouds_f64(src.astype(np.float64), np.zeros((0, 3)))
    r = gpu_ctx.run(None, 0.5, 3, 0, 0)
    assert r.num_correspondences == 0
    gpu_ctx.set_clouds_f64(np.zeros((0, 3)), tgt.astype(np.float64))
    r = gpu_ctx.run(None, 0.5, 3, 0, 0)
    assert r.num_correspondences == 0


@pytest.mark.parametrize("offset", [None, (3.0, -2.0, 1.5)])
def test_run_parity_5k_20k(gpu_ctx, oracle, offset):
    src, tgt, Tgt, _ = synth.make_pair(5000, 20000, offset=offset)
    gpu_ctx.set_clouds_f64(src, tgt)
    r = gpu_ctx.run(None, 0.075, 20, 0, 0)
    o = oracle.registration_icp(src, tgt, 0.075, max_iter=20, rel_fitness=0, rel_rmse=0)
    k = oracle.k_registration_icp(src, tgt, 0.075, max_iter=20, rel_fitness=0, rel_rmse=0)
    assert r.iterations == 20 and r.nn_passes == 21
    assert synth.rel_frobenius(r.transformation_, o.T) < TOL_T
    assert synth.rel_frobenius(r.transformation_, k.T) < 1e-9      # same arithmetic
    assert r.num_correspondences == o.k
    idx = gpu_ctx.correspondence_index()
    assert np.mean(idx == o.idx) >= 0.9999
    assert abs(r.fitness_ - o.fitness) < 1e-12 and abs(r.inlier_rmse_ - o.rmse) < 1e-6


def test_run_parity_64k_256k(gpu_ctx, oracle):
    src, tgt, Tgt, radius = synth.make_pair(65536, 262144)
    gpu_ctx.set_clouds_f64(src, tgt)
    r = gpu_ctx.run(None, radius, 10, 0, 0)
    o = oracle.registration_icp(src, tgt, radius, max_iter=10, rel_fitness=0, rel_rmse=0, grid=True)
    assert synth.rel_frobenius(r.transformation_, o.T) < TOL_T
    assert abs(r.num_correspondences - o.k) <= max(2, int(1e-4 * o.k))


def test_termination_semantics(gpu_ctx, oracle):
    src, tgt, _, _ = synth.make_pair(4000, 8000)
    gpu_ctx.set_clouds_f64(src, tgt)
    r = gpu_ctx.run(None, 0.075, 60, 1e-6, 1e-6)
    o = oracle.registration_icp(src, tgt, 0.075, max_iter=60, rel_fitness=1e-6, rel_rmse=1e-6)
    assert r.iterations < 60
    assert abs(r.iterations - o.iters) <= 1
    assert synth.rel_frobenius(r.transformation_, o.T) < 1e-4


def test_solver_modes_converge_to_same_fixed_point(gpu_ctx):
    src, tgt, _, _ = synth.make_pair(3000, 9000)
    gpu_ctx.set_clouds_f64(src, tgt)
    a = gpu_ctx.run(None, 0.075, 150, 0, 0, solver=0)
    b = gpu_ctx.run(None, 0.075, 150, 0, 0, solver=1)
    c = gpu_ctx.run(None, 0.075, 150, 0, 0, solver=2)
    assert synth.rel_frobenius(b.transformation_, a.transformation_) < 1e-6
    assert synth.rel_frobenius(c.transformation_, a.transformation_) < 1e-6


def test_with_scaling(gpu_ctx, oracle):
    src, tgt, _, _ = synth.make_pair(3000, 9000)
    src = src * 1.02
    gpu_ctx.set_clouds_f64(src, tgt)
    r = gpu_ctx.run(None, 0.1, 15, 0, 0, with_scaling=True)
    o = oracle.registration_icp(src, tgt, 0.1, max_iter=15, rel_fitness=0, rel_rmse=0, with_scaling=True)
    assert synth.rel_frobenius(r.transformation_, o.T) < TOL_T


def test_repeatable_bitwise(gpu_ctx):
    src, tgt, _, _ = synth.make_pair(6000, 15000)
    gpu_ctx.set_clouds_f64(src, tgt)
    a = gpu_ctx.run(None, 0.075, 10, 0, 0)
    b = gpu_ctx.run(None, 0.075, 10, 0, 0)
    assert np.array_equal(a.transformation_, b.transformation_)
    assert a.inlier_rmse_ == b.inlier_rmse_


def test_create_fails_loudly_on_bad_device(lib):
    with pytest.raises(lib.IcpError):
        lib.Context(99)


def test_device_loop_equals_host_loop(gpu_ctx, oracle):
    """The on-device loop (solve + stop test in a kernel epilogue) and the
    synchronous host loop are the same algorithm: same iteration count, same K,
    transforms equal to rounding (device vs host libm)."""
    src, tgt, _, _ = synth.make_pair(6000, 15000)
    gpu_ctx.set_clouds_f64(src, tgt)
    for kw in (dict(max_iter=25, rel_fitness=1e-6, rel_rmse=1e-6),
               dict(max_iter=12, rel_fitness=0.0, rel_rmse=0.0),
               dict(max_iter=0, rel_fitness=0.0, rel_rmse=0.0),
               dict(max_iter=9, rel_fitness=0.0, rel_rmse=0.0, solver=1),
               dict(max_iter=9, rel_fitness=0.0, rel_rmse=0.0, solver=2)):
        try:
            gpu_ctx.set_device_loop(True)
            a = gpu_ctx.run(None, 0.075, **kw)
            ia = gpu_ctx.correspondence_index()
            gpu_ctx.set_device_loop(False)
            b = gpu_ctx.run(None, 0.075, **kw)
            ib = gpu_ctx.correspondence_index()
        finally:
            gpu_ctx.set_device_loop(None)
        assert a.iterations == b.iterations and a.nn_passes == b.nn_passes
        assert a.num_correspondences == b.num_correspondences
        assert synth.rel_frobenius(a.transformation_, b.transformation_) < 1e-12
        assert abs(a.inlier_rmse_ - b.inlier_rmse_) < 1e-12 and a.fitness_ == b.fitness_
        assert np.array_equal(ia, ib)
    gpu_ctx.set_device_loop(True)
    Ta, la = gpu_ctx.iterate(None, 0.075, 7)
    gpu_ctx.set_device_loop(False)
    try:
        Tb, lb = gpu_ctx.iterate(None, 0.075, 7)
    finally:
        gpu_ctx.set_device_loop(None)
    assert synth.rel_frobenius(Ta, Tb) < 1e-12 and la.num_correspondences == lb.num_correspondences


def test_batch_of_different_clouds_in_flight(gpu_ctx_auto, oracle):
    """BASELINE config 3: many small ICPs with their own clouds advanced together
    on the device == the same problems run one at a time."""
    ctx = gpu_ctx_auto
    rng = np.random.default_rng(17)
    probs = []
    for i in range(14):
        ns, nt = int(rng.integers(300, 6000)), int(rng.integers(200, 9000))
        src, tgt, _, _ = synth.make_pair(ns, nt, seed_t=100 + i, seed_s=200 + i)
        init = synth.make_T(synth.rot_y(0.01 * i), [0.002 * i, 0, -0.001 * i])
        probs.append((src + 0.1 * i, tgt + 0.1 * i, init, [0.05, 0.075, 0.02][i % 3]))
    probs.append((probs[0][0] + 30.0, probs[0][1], None, 0.01))      # no correspondences at all
    probs.append((probs[1][0][:1], probs[1][1], None, 0.5))           # a single source point
    batched = ctx.run_batch(probs, max_iter=25, rel_fitness=1e-6, rel_rmse=1e-6)
    ctx.set_device_loop(False)
    try:
        single = ctx.run_batch(probs, max_iter=25, rel_fitness=1e-6, rel_rmse=1e-6)
    finally:
        ctx.set_device_loop(None)
    for i, (a, b) in enumerate(zip(batched, single)):
        assert a.num_correspondences == b.num_correspondences, i
        assert a.iterations == b.iterations and a.nn_passes == b.nn_passes, i
        if i != 15:      # one source point: the rotation is undetermined
            assert synth.rel_frobenius(a.transformation_, b.transformation_) < 1e-9, i
        assert abs(a.inlier_rmse_ - b.inlier_rmse_) < 1e-9 and a.fitness_ == b.fitness_
    s0, t0, i0, r0 = probs[3]
    o = oracle.registration_icp(s0, t0, r0, init=i0, max_iter=25)
    assert synth.rel_frobenius(batched[3].transformation_, o.T) < TOL_T
    assert batched[3].num_correspondences == o.k


@pytest.mark.gpu
def test_batch_shares_uploads_and_grids_between_problems_with_the_same_clouds(gpu_ctx_auto):
    """24 yaw starts of 3 models against ONE scene (the shape of src/annotation.cpp:35-61,103-168):
    problems that pass the same arrays share one packed copy / upload / grid.  Same answers,
    bit for bit, as the batch given private copies of every cloud."""
    ctx = gpu_ctx_auto
    scene = synth.surface_points(30000, 77)
    models = [synth.surface_points(n, 80 + i)[:n] * 0.999 for i, n in enumerate((3000, 5000, 800))]
    shared, private = [], []
    for m in models:
        for k in range(8):
            init = synth.make_T(synth.rot_y(2 * np.pi * k / 8), [0.001 * k, 0, 0])
            r = 0.05 if k % 2 else 0.03                       # two radii -> two grids over one scene
            shared.append((m, scene, init, r))
            private.append((m.copy(), scene.copy(), init, r))
    a = ctx.run_batch(shared, max_iter=20)
    b = ctx.run_batch(private, max_iter=20)
    for x, y in zip(a, b):
        assert np.array_equal(x.transformation_, y.transformation_)
        assert x.num_correspondences == y.num_correspondences and x.iterations == y.iterations
        assert x.inlier_rmse_ == y.inlier_rmse_


@pytest.mark.gpu
def test_half_pitch_rows_option_is_exact(lib, oracle):
    """VISMA_ICP_GRID_SUB=2 (25 half-pitch rows, second ring visited lazily; off by default because it
    measured slower) must return the brute-force answer too -- including queries without any
    neighbour, which walk all 25 rows."""
    os.environ["VISMA_ICP_GRID_SUB"] = "2"
    try:
        ctx = lib.Context(0)
        ctx.set_search_precision("f32")                       # compared with the (fp32) brute-force kernel
    finally:
        del os.environ["VISMA_ICP_GRID_SUB"]
    src, tgt, T_gt, r = synth.make_pair(20000, 120000, motion="radius")
    src = np.concatenate([src, src[:500] + [0.0, 3.0 * r, 0.0], src[:50] + 5.0])     # near misses, far outliers
    ctx.set_clouds_f64(src, tgt)
    out = {}
    for name, mode in (("grid", lib.NN_GRID), ("brute", lib.NN_BRUTE)):
        ctx.set_nn_mode(mode)
        ctx.nn_pass(np.eye(4), r)
        st = ctx.reduce()
        out[name] = (ctx.correspondence_index(), ctx.get_correspondences()[2], st)
    assert np.array_equal(out["grid"][0], out["brute"][0])
    assert np.array_equal(out["grid"][1].view(np.uint32), out["brute"][1].view(np.uint32))
    assert (out["grid"][0] < 0).sum() >= 50
    a = ctx.run(None, r, 10, 0, 0)
    ctx.set_nn_mode(lib.NN_BRUTE)
    b = ctx.run(None, r, 10, 0, 0)
    assert synth.rel_frobenius(a.transformation_, b.transformation_) < 1e-12


@pytest.mark.gpu
def test_randomised_grid_equals_brute_force(lib):
    """tools/fuzz_grid_vs_brute.py: 300 random configurations (lattices on cell faces, duplicates,
    planes, elongated boxes, far-off origins, extreme radii) -- the two searches must agree bit for bit."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_grid_vs_brute.py"), "300", "7"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "300 configurations, 0 mismatches" in p.stdout
