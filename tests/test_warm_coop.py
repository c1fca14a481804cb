"""The warm-started, wave-cooperative exact search (visma_amd/csrc/grid_coop.hip) against the lane-serial
exact kernel it replaces after the first pass of a registration: correspondences, distances and all 38
statistics BIT for bit (same query -> lane map, same summation tree), on random passes, with exact ties
(points given several times), dense rows (chunk lists longer than one LDS window), several queries per
lane, a target of seven points, point-to-plane; and the state it starts from is dropped whenever the
source, the target or the radius changes."""
import os

import numpy as np
import pytest

from visma_amd import _lib, synth

pytestmark = pytest.mark.gpu


def ctx_env(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return _lib.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def rand_T(rng, ang, tr):
    a = rng.normal(size=3)
    a /= np.linalg.norm(a)
    th = rng.uniform(0, ang)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    T = np.eye(4)
    T[:3, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    T[:3, 3] = rng.normal(size=3) * tr
    return T


SERIAL = {"VISMA_ICP_COOP": "0", "VISMA_ICP_GRID_LANES": "801"}        # lane-serial, one lane per query
CASES = [
    # ns, nt, radius (None = default), duplicate the target, passes
    ("5k-20k", 5000, 20000, None, 0, 4),
    ("big radius", 3000, 8000, 0.075, 0, 3),
    ("degenerate grid", 2000, 500, 0.2, 0, 3),
    ("seven targets", 300, 7, 0.5, 0, 2),
    ("points given four times", 4000, 30000, None, 3, 3),
    ("dense rows", 70000, 60000, 0.05, 0, 2),
    ("64k-1M", 65536, 1048576, None, 0, 2),
    ("several queries per lane", 1000000, 2000000, None, 0, 2),
]


@pytest.mark.parametrize("name,ns,nt,radius,dup,passes", CASES, ids=[c[0] for c in CASES])
def test_warm_search_equals_the_lane_serial_search_bit_for_bit(lib, name, ns, nt, radius, dup, passes):
    rng = np.random.default_rng(ns + nt)
    src, tgt, T_gt, r = synth.make_pair(ns, nt, seed_t=ns, seed_s=nt + 7, motion="radius")
    if radius is not None:
        r = radius
    if dup:
        tgt = np.concatenate([tgt] + [tgt[rng.permutation(len(tgt))[: len(tgt) // 2]] for _ in range(dup)])
    ref = ctx_env(SERIAL)
    auto = _lib.Context(0)                                        # first pass lane-serial, then warm
    always = ctx_env({"VISMA_ICP_GRID_LANES": "9901"})            # warm kernel from the first pass (radius pruning)
    for c in (ref, auto, always):
        c.set_nn_mode(lib.NN_GRID)
        c.set_clouds_f64(src, tgt)
    for p in range(passes):
        T = T_gt @ rand_T(rng, r * 0.8, r * 0.5) if p else np.eye(4)
        ref.nn_pass(T, r)
        st0 = ref.reduce()
        i0, d0 = ref.correspondence_index(), ref.get_correspondences()[2]
        for which, c in (("auto", auto), ("always", always)):
            c.nn_pass(T, r)
            st = c.reduce()
            assert c.search_mode_used() == "exact"
            assert np.array_equal(c.correspondence_index(), i0), (name, which, p)
            assert np.array_equal(c.get_correspondences()[2].view(np.uint32), d0.view(np.uint32)), (name, which, p)
            if which == "always" or p > 0:                        # (auto's cold pass may use more lanes per query)
                assert np.array_equal(st.view(np.uint64), st0.view(np.uint64)), (name, which, p)
            else:
                assert st[0] == st0[0] and np.max(np.abs(st - st0) / np.maximum(np.abs(st0), 1.0)) < 1e-10
    for c in (ref, auto, always):
        c.close()


def test_the_warm_state_is_dropped_with_the_clouds_and_the_radius(lib):
    """Winners remembered for another source order, target or grid must never be read as bounds."""
    rng = np.random.default_rng(5)
    a_src, a_tgt, T_gt, r = synth.make_pair(6000, 30000, seed_t=1, seed_s=2, motion="radius")
    b_src, b_tgt, _, _ = synth.make_pair(6000, 30000, seed_t=3, seed_s=4, motion="radius")
    b_tgt = b_tgt + 0.4                                            # far from where a's winners were
    ref = ctx_env(SERIAL)
    c = _lib.Context(0)

    def same(src, tgt, T, rad):
        out = []
        for x in (ref, c):
            x.nn_pass(T, rad)
            st = x.reduce()
            out.append((x.correspondence_index(), st[0]))
        assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]

    for x in (ref, c):
        x.set_nn_mode(lib.NN_GRID)
        x.set_clouds_f64(a_src, a_tgt)
    same(a_src, a_tgt, np.eye(4), r)
    same(a_src, a_tgt, T_gt, r)                                     # warm
    for x in (ref, c):
        x.set_clouds_f64(a_src, b_tgt)                              # new target, same source
    same(a_src, b_tgt, np.eye(4), r)
    same(a_src, b_tgt, T_gt, r)
    for x in (ref, c):
        x.set_clouds_f64(b_src[::-1].copy(), b_tgt)                 # new source
    same(b_src, b_tgt, np.eye(4), r)
    same(b_src, b_tgt, T_gt, r * 2.5)                               # new radius: new grid
    same(b_src, b_tgt, T_gt, r * 2.5)
    same(b_src, b_tgt, rand_T(rng, 0.5, 0.3), r * 2.5)             # a pose far from the previous one
    same(b_src, b_tgt, T_gt, r * 0.5)
    ref.close(); c.close()


def test_runs_sweeps_and_batches_give_the_lane_serial_results(lib):
    """Whole registrations (host loop, device loop, the 24-yaw sweep, a batch with own clouds, point-to-plane):
    the warm kernel serves every pass after the first; results equal the lane-serial library's."""
    src, tgt, T_gt, r = synth.make_pair(20000, 90000, seed_t=8, seed_s=9, motion="radius")
    ref = ctx_env({"VISMA_ICP_COOP": "0"})
    c = _lib.Context(0)
    nrm = c.estimate_normals(tgt, knn=12)
    for x in (ref, c):
        x.set_clouds_f64(src, tgt)
        x.set_target_normals_f64(nrm)
    for loop in (False, True):
        for x in (ref, c):
            x.set_device_loop(loop)
        a, b = ref.run(None, r, 25, 1e-6, 1e-6), c.run(None, r, 25, 1e-6, 1e-6)
        assert a.num_correspondences == b.num_correspondences and a.iterations == b.iterations
        assert synth.rel_frobenius(b.transformation_, a.transformation_) < 1e-13
        assert np.array_equal(ref.correspondence_index(), c.correspondence_index())
        # (the two libraries sum the first pass's moments in different trees: last-bit differences, which a
        # Gauss-Newton iteration carries along)
        a, b = ref.run_point_to_plane(None, r, 12, 0, 0), c.run_point_to_plane(None, r, 12, 0, 0)
        assert a.num_correspondences == b.num_correspondences
        assert synth.rel_frobenius(b.transformation_, a.transformation_) < 1e-10
    for x in (ref, c):
        x.set_device_loop(None)
    ba, la, pa = ref.run_yaw_sweep(8, 3 * r, 15)
    bb, lb, pb = c.run_yaw_sweep(8, 3 * r, 15)
    assert la == lb
    for u, v in zip(pa, pb):
        assert u.num_correspondences == v.num_correspondences and u.iterations == v.iterations
        assert synth.rel_frobenius(v.transformation_, u.transformation_) < 1e-12
    probs = []
    for k in range(5):
        s, t, _, rr = synth.make_pair(3000 + 700 * k, 9000 + 2000 * k, seed_t=20 + k, seed_s=40 + k, motion="radius")
        probs.append((s, t, np.eye(4), rr))
    ga, gb = ref.run_batch(probs, max_iter=15), c.run_batch(probs, max_iter=15)
    for u, v in zip(ga, gb):
        assert u.num_correspondences == v.num_correspondences and u.iterations == v.iterations
        assert synth.rel_frobenius(v.transformation_, u.transformation_) < 1e-12
    ref.close(); c.close()
