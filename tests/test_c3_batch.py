"""BASELINE config 3 at its real shape: all objects of one scene in flight -- 12 objects x 24 yaw
starts = 288 ICPs in ONE visma_icp_run_batch -- against the oracle for a sampled subset, and
against one-at-a-time runs for every problem (bench.py --workload c3 times this set)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import c3_problems  # noqa: E402
from visma_amd import _lib, synth  # noqa: E402


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_288_problems_in_flight(lib, oracle):
    objs, probs = c3_problems()
    assert len(probs) == 288
    ctx = _lib.Context(0)
    got = ctx.run_batch([p[:4] for p in probs], max_iter=30)
    assert ctx.search_mode_used() == "exact"
    # every problem against a one-at-a-time host-loop run of the library
    one = _lib.Context(0)
    one.set_device_loop(False)
    for oi in range(0, 12, 5):                                   # three objects x 24 starts, one at a time
        src, tgt = objs[oi]
        one.set_clouds_f64(src, tgt)
        for k in range(24):
            i = oi * 24 + k
            w = one.run(probs[i][2], 0.02, 30)
            assert got[i].num_correspondences == w.num_correspondences, (oi, k)
            assert got[i].iterations == w.iterations
            assert synth.rel_frobenius(got[i].transformation_, w.transformation_) < 1e-11
    # a sample against the oracle (the CPU restatement of Open3D's RegistrationICP)
    rng = np.random.default_rng(0)
    for i in rng.choice(288, 6, replace=False):
        src, tgt, init, r, _ = probs[int(i)]
        w = oracle.registration_icp(src, tgt, r, init=init, max_iter=30, grid=True)
        assert got[i].num_correspondences == w.k, i
        if w.k >= 3:
            assert synth.rel_frobenius(got[i].transformation_, w.T) < 1e-9, i


def _ellipsoid(n, seed, axes=(0.5, 0.3, 0.2), centre=(0.3, -0.2, 1.0), noise=0.0):
    """points on an ellipsoid with three different axes (all six degrees of freedom are constrained)
    and their exact unit normals"""
    rng = np.random.default_rng(seed)
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    a = np.asarray(axes)
    p = u * a
    nrm = p / a ** 2
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    p = p + rng.normal(size=p.shape) * noise + np.asarray(centre)
    return p, nrm


def _pp_problems():
    probs = []
    for k, (ns, nt, r) in enumerate([(3000, 9000, 0.03), (5000, 20000, 0.02), (1500, 4000, 0.045)]):
        tgt, nrm = _ellipsoid(nt, 40 + k, noise=2e-4)
        src, _ = _ellipsoid(ns, 50 + k)
        a = 0.02
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
        c = tgt.mean(0)
        src = (src - c) @ R.T + c + np.array([0.004, -0.003, 0.002])
        for y in range(3):                                       # the same clouds from three starts: shared uploads
            b = 0.01 * y
            init = np.eye(4)
            Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
            init[:3, :3] = Ry
            init[:3, 3] = c - Ry @ c
            probs.append((src, tgt, nrm, init, r))
    return probs


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("precision", ["exact", "f32"])
def test_batch_point_to_plane_equals_single_runs_and_oracle(lib, oracle, precision):
    """visma_icp_run_batch_point_to_plane: problems with their own clouds and normals, shared targets
    included, against one-at-a-time runs of the library and the oracle's point-to-plane ICP."""
    from oracle.oracle import EST_POINT_TO_PLANE
    probs = _pp_problems()
    ctx = _lib.Context(0)
    ctx.set_search_precision(precision)
    got = ctx.run_batch_point_to_plane(probs, max_iter=15)
    assert ctx.search_mode_used() == precision
    one = _lib.Context(0)
    one.set_search_precision(precision)
    one.set_device_loop(False)
    for i, (src, tgt, nrm, init, r) in enumerate(probs):
        one.set_clouds_f64(src, tgt)
        one.set_target_normals_f64(nrm)
        w = one.run_point_to_plane(init, r, 15)
        assert w.fitness_ > 0.9, (i, w)                            # the registrations do converge
        assert got[i].num_correspondences == w.num_correspondences, i
        assert got[i].iterations == w.iterations, i
        assert synth.rel_frobenius(got[i].transformation_, w.transformation_) < (1e-10 if precision == "exact" else 1e-6), i
        if precision == "exact" and i % 3 == 0:
            o = oracle.registration_icp(src, tgt, r, init=init, max_iter=15, estimator=EST_POINT_TO_PLANE,
                                        tgt_normals=nrm, grid=True)
            assert got[i].num_correspondences == o.k, i
            assert synth.rel_frobenius(got[i].transformation_, o.T) < 1e-9, i


@pytest.mark.gpu
def test_batch_point_to_plane_without_normals_returns_the_initial_transform(lib):
    tgt, nrm = _ellipsoid(6000, 3)
    src, _ = _ellipsoid(2000, 4)
    r = 0.04
    init = np.eye(4); init[0, 3] = 1e-3
    ctx = _lib.Context(0)
    got = ctx.run_batch_point_to_plane([(src, tgt, nrm, init, r), (src, tgt.copy(), None, init, r)], max_iter=5)
    assert got[0].num_correspondences > 0 and not np.allclose(got[0].transformation_, init)
    assert np.array_equal(got[1].transformation_, init)          # Registration.cpp:152-157


def test_abi_exports_the_multi_context_batch(lib):
    assert hasattr(lib.load(), "visma_icp_run_batch_multi")


@pytest.mark.gpu
def test_batch_over_worker_contexts_equals_one_context(lib):
    """visma_icp_run_batch_multi: the problems dealt (targets kept together) to 1, 2 and 3 contexts on the same GPU,
    shares side by side -- every problem's result is the single-context batch's, bit for bit."""
    from visma_amd import _lib, synth
    rng = np.random.default_rng(5)
    probs = []
    for k, (ns, nt) in enumerate(((3000, 9000), (7000, 5000), (1500, 12000), (5000, 5000), (900, 700))):
        src, tgt, T_gt, _ = synth.make_pair(ns, nt, seed_t=30 + k, seed_s=60 + k, motion="fixed")
        for yaw in (0.0, 0.2, -0.3):
            probs.append((src, tgt, synth.make_T(synth.rot_y(yaw), rng.normal(size=3) * 0.01), 0.05 + 0.01 * k))
    ctxs = [_lib.Context(0) for _ in range(3)]
    want = ctxs[0].run_batch(probs, max_iter=15)
    for W in (1, 2, 3):
        got = _lib.run_batch_multi(ctxs[:W], probs, max_iter=15)
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert a.num_correspondences == b.num_correspondences and a.iterations == b.iterations
            assert np.array_equal(a.transformation_, b.transformation_)
    for c in ctxs:
        c.close()


@pytest.mark.gpu
def test_batch_over_worker_contexts_edge_cases(lib):
    """more contexts than target groups, an empty batch, a failing share (message comes back)"""
    src, tgt, _, _ = synth.make_pair(2000, 4000, seed_t=3, seed_s=4, motion="fixed")
    ctxs = [_lib.Context(0) for _ in range(3)]
    probs = [(src, tgt, synth.make_T(synth.rot_y(0.1 * k), [0, 0, 0]), 0.05) for k in range(4)]   # ONE target group
    want = ctxs[0].run_batch(probs, max_iter=8)
    got = _lib.run_batch_multi(ctxs, probs, max_iter=8)              # only one context gets work
    for a, b in zip(got, want):
        assert np.array_equal(a.transformation_, b.transformation_) and a.num_correspondences == b.num_correspondences
    assert _lib.run_batch_multi(ctxs, [], max_iter=8) == []
    L = _lib.load()
    import ctypes as C
    h = (C.c_void_p * 2)(ctxs[0]._h, None)
    err = C.create_string_buffer(256)
    arr, n, _keep, out = ctxs[0].make_batch(probs)
    assert L.visma_icp_run_batch_multi(h, 2, arr, n, 8, 1e-6, 1e-6, 0, out, err, 256) != 0 and b"NULL context" in err.value
    assert L.visma_icp_run_batch_multi(h, 1, arr, n, 8, 1e-6, 1e-6, 99, out, err, 256) != 0 and len(err.value) > 0   # unknown solver
    for c in ctxs:
        c.close()
