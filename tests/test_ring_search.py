"""The ring search over cells SMALLER than the radius (visma_amd/csrc/grid_ring.hip, visma_icp_set_ring_search) against
the search it stands in for when the radius is large against the target's point spacing: the radius-cell kernels
(27-cell neighbourhoods).  Correspondences and distances BIT for bit -- first passes (bounded by the radius), later
passes (bounded by the previous winner under a small or a LARGE motion), queries without a partner, queries far outside
the table, points given several times (ties go to the lowest index), a target of seven points, several queries per lane
octet; the statistics to rounding (another summation order); whole registrations, yaw sweeps and the device-resident loop
to 1e-10; point-to-plane; and SURVEY 8d's literal ground truth on a small pair against the CPU oracle."""
import numpy as np
import pytest

from visma_amd import _lib, synth

pytestmark = pytest.mark.gpu


def rand_T(rng, ang, tr):
    a = rng.normal(size=3)
    a /= np.linalg.norm(a)
    th = rng.uniform(0, ang)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    T = np.eye(4)
    T[:3, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    T[:3, 3] = rng.normal(size=3) * tr
    return T


def pair_of_contexts(lib, src, tgt, target_occupancy=None):
    import os
    ref = _lib.Context(0)
    ref.set_ring_search(0)
    old = os.environ.get("VISMA_ICP_RING_TARGET")
    if target_occupancy is not None:
        os.environ["VISMA_ICP_RING_TARGET"] = str(target_occupancy)
    try:
        ring = _lib.Context(0)
    finally:
        if target_occupancy is not None:
            if old is None:
                os.environ.pop("VISMA_ICP_RING_TARGET", None)
            else:
                os.environ["VISMA_ICP_RING_TARGET"] = old
    ring.set_ring_search(1)
    for c in (ref, ring):
        c.set_nn_mode(lib.NN_GRID)
        c.set_clouds_f64(src, tgt)
    return ref, ring


CASES = [
    # name, ns, nt, radius, duplicates, cell occupancy asked for, motions of the passes after the first
    ("literal motion 4k-65k r=0.15", 4096, 65536, 0.15, 0, None, ("small", "small", "large", "far")),
    ("many rings", 3000, 50000, 0.2, 0, 1.5, ("small", "large", "small")),
    ("radius just above a cell", 3000, 200000, 0.03, 0, 6.0, ("small", "large")),
    ("points given four times", 2500, 30000, 0.1, 3, None, ("small", "large", "small")),
    ("seven targets", 300, 7, 0.5, 0, 1.0, ("small", "large")),
    ("several queries per octet", 300000, 100000, 0.1, 0, None, ("small", "large")),
    ("65k-1M r=0.1", 65536, 1048576, 0.1, 0, None, ("small", "small")),
    # (more queries than lane groups: the launch's cap of 32,768 workgroups -- the variant that keeps its sums live)
    ("1.2M queries", 1200000, 150000, 0.08, 0, None, ("small",)),
]


@pytest.mark.parametrize("name,ns,nt,radius,dup,occ,motions", CASES, ids=[c[0] for c in CASES])
def test_ring_search_equals_the_radius_cell_search_bit_for_bit(lib, name, ns, nt, radius, dup, occ, motions):
    rng = np.random.default_rng(ns * 7 + nt)
    src, tgt, T_gt, _ = synth.make_pair(ns, nt, seed_t=ns + 1, seed_s=nt + 3, motion="fixed")
    if dup:
        tgt = np.concatenate([tgt] + [tgt[rng.permutation(len(tgt))[: len(tgt) // 2]] for _ in range(dup)])
    ref, ring = pair_of_contexts(lib, src, tgt, occ)
    T = np.eye(4)
    for p, motion in enumerate(("first",) + tuple(motions)):
        if motion == "small":
            T = T_gt @ rand_T(rng, 0.01, 0.004)
        elif motion == "large":
            T = T_gt @ rand_T(rng, 0.3, radius * 0.7)
        elif motion == "far":                                   # most queries leave the table altogether
            T = T_gt @ rand_T(rng, 0.2, 0.1)
            T[:3, 3] += np.array([2.6, -0.4, 0.3])
        ref.nn_pass(T, radius)
        s0 = ref.reduce()
        i0, d0 = ref.correspondence_index(), ref.get_correspondences()[2]
        ring.nn_pass(T, radius)
        s1 = ring.reduce()
        info = ring.ring_search()
        assert info["rings"] >= 2 and ring.search_kernel_used() == "ring", (name, info)
        assert ref.ring_search()["rings"] == 0 and ref.search_kernel_used() != "ring"
        assert np.array_equal(ring.correspondence_index(), i0), (name, p, motion)
        assert np.array_equal(ring.get_correspondences()[2].view(np.uint32), d0.view(np.uint32)), (name, p, motion)
        assert s1[0] == s0[0]
        scale = np.maximum(np.abs(s0), 1e-300)
        assert np.all(np.abs(s1 - s0) <= 1e-11 * np.maximum(scale, np.abs(s0).max() * 1e-3)), (name, p, motion)
        if motion == "far":
            assert s0[0] < 0.5 * ns                              # (the case is what it claims to be)


def test_the_occupancy_rule_switches_by_itself(lib):
    """Default mode: a radius-sized cell that holds hundreds of points sends the context to the ring search, the same
    target under a radius of a few point spacings does not; the count is reported."""
    src, tgt, T_gt, r_small = synth.make_pair(4096, 262144, motion="fixed")
    c = _lib.Context(0)
    c.set_nn_mode(lib.NN_GRID)
    c.set_clouds_f64(src, tgt)
    c.nn_pass(np.eye(4), 0.3)
    big = c.ring_search()
    assert big["rings"] >= 2 and big["occupancy"] >= 256 and big["cell"] < 0.3, big
    idx_ring = c.correspondence_index()
    c.nn_pass(np.eye(4), r_small)
    c.reduce()
    small = c.ring_search()
    assert small["rings"] == 0 and c.search_kernel_used() != "ring", small
    c.set_ring_search(0)
    c.nn_pass(np.eye(4), 0.3)
    assert c.ring_search()["rings"] == 0
    assert np.array_equal(c.correspondence_index(), idx_ring)


def test_between_the_thresholds_the_table_follows_the_caller(lib):
    """20 <= points per occupied radius-sized cell < 48: one registration at a time takes the ring search, a yaw sweep
    radius-sized cells (its far-off starts leave most queries without a partner); a context that alternates re-plans
    three times, the third time for good: radius-sized cells for that target.  Results do not depend on any of it."""
    src, tgt, T_gt, _ = synth.make_pair(20000, 262144, motion="fixed")
    r = 0.05
    c = _lib.Context(0)
    c.set_nn_mode(lib.NN_GRID)
    c.set_clouds_f64(src, tgt)
    a = c.run(None, r, 8)
    info = c.ring_search()
    assert 20 <= info["occupancy"] < 48 and info["rings"] >= 2 and c.search_kernel_used() == "ring", info
    sw = c.run_yaw_sweep(4, r, 6)
    assert c.ring_search()["rings"] == 0 and c.search_kernel_used() != "ring"
    b = c.run(None, r, 8)
    assert c.ring_search()["rings"] >= 2 and c.search_kernel_used() == "ring"
    assert np.abs(a.transformation_ - b.transformation_).max() < 1e-12
    sw2 = c.run_yaw_sweep(4, r, 6)                               # (the third change of caller: settled on ...)
    kept = c.ring_search()["rings"]
    d = c.run(None, r, 8)
    assert c.ring_search()["rings"] == kept == 0                  # (... radius-sized cells)
    assert np.abs(a.transformation_ - d.transformation_).max() < 1e-10
    for x, y in zip(sw[2], sw2[2]):
        assert x.iterations == y.iterations and np.abs(x.transformation_ - y.transformation_).max() < 1e-10


@pytest.mark.parametrize("loop", ["host", "device"])
def test_registrations_agree(lib, loop):
    """A whole registration from the literal motion's start: ring search == radius cells to rounding, and the literal
    ground truth is recovered."""
    src, tgt, T_gt, _ = synth.make_pair(16384, 65536, motion="fixed")
    ref, ring = pair_of_contexts(lib, src, tgt)
    out = []
    for c in (ref, ring):
        c.set_device_loop(loop == "device")
        out.append(c.run(None, 0.15, 30))
    a, b = out
    assert ring.search_kernel_used() == "ring"
    assert a.iterations == b.iterations
    assert np.abs(a.transformation_ - b.transformation_).max() < 1e-10
    assert abs(a.fitness_ - b.fitness_) < 1e-12 and abs(a.inlier_rmse_ - b.inlier_rmse_) < 1e-10
    assert np.abs(b.transformation_ - T_gt).max() < 5e-3


def test_device_loop_with_several_queries_per_lane_group(lib):
    """Device-resident loops cap a problem at 1,024 workgroups: 100,000 queries are three per lane group."""
    src, tgt, T_gt, _ = synth.make_pair(100000, 120000, motion="fixed")
    ref, ring = pair_of_contexts(lib, src, tgt)
    out = []
    for c in (ref, ring):
        c.set_device_loop(True)
        out.append(c.run(None, 0.12, 12))
    a, b = out
    assert ring.search_kernel_used() == "ring" and a.iterations == b.iterations
    assert np.abs(a.transformation_ - b.transformation_).max() < 1e-10 and abs(a.fitness_ - b.fitness_) < 1e-12
    assert np.array_equal(ref.correspondence_index(), ring.correspondence_index())


def test_yaw_sweep_and_point_to_plane_agree(lib):
    src, tgt, T_gt, _ = synth.make_pair(5000, 40000, motion="fixed")
    ref, ring = pair_of_contexts(lib, src, tgt, 4.0)
    sw = [c.run_yaw_sweep(8, 0.12, 12) for c in (ref, ring)]
    assert ring.search_kernel_used() == "ring"
    assert sw[0][1] == sw[1][1]
    for a, b in zip(sw[0][2], sw[1][2]):
        assert a.iterations == b.iterations and np.abs(a.transformation_ - b.transformation_).max() < 1e-9 and abs(a.fitness_ - b.fitness_) < 1e-12
    nrm = tgt / np.linalg.norm(tgt, axis=1, keepdims=True)      # (any unit vectors: the estimator only has to agree)
    pl = []
    for c in (ref, ring):
        c.set_target_normals_f64(nrm)
        pl.append(c.run_point_to_plane(None, 0.12, 10))
    assert pl[0].iterations == pl[1].iterations and np.abs(pl[0].transformation_ - pl[1].transformation_).max() < 1e-9


def test_literal_ground_truth_against_the_oracle(lib, oracle):
    """SURVEY 8d's literal T_gt (5 deg, 1 deg, ~3 cm) with the radius it needs, small enough for the CPU oracle:
    correspondences of the first and of a later pass, then the registration's transform."""
    src, tgt, T_gt, _ = synth.make_pair(1500, 12000, motion="fixed")
    c = _lib.Context(0)
    c.set_ring_search(1)
    c.set_nn_mode(lib.NN_GRID)
    c.set_clouds_f64(src, tgt)
    r = 0.15
    rng = np.random.default_rng(5)
    for T in (np.eye(4), T_gt @ rand_T(rng, 0.02, 0.01)):
        c.nn_pass(T, r)
        k, idx, d2, _ = oracle.nn_pass(oracle.transform_points(src, T), tgt, r)
        assert np.array_equal(c.correspondence_index(), idx) and len(c.get_correspondences()[0]) == k
        assert c.search_kernel_used() == "ring"
    res = c.run(None, r, 30)
    want = oracle.registration_icp(src, tgt, r, None, 30)
    assert np.abs(res.transformation_ - want.T).max() < 1e-9 and res.iterations == want.iters


def test_empty_and_tiny_clouds_with_the_ring_search_asked_for(lib):
    """An empty source, an empty target, one point against one point: nothing to match or one pair, whatever is asked for."""
    src, tgt, _, _ = synth.make_pair(2000, 30000, motion="fixed")
    c = _lib.Context(0)
    c.set_ring_search(1)
    c.set_nn_mode(lib.NN_GRID)
    c.set_clouds_f64(np.zeros((0, 3)), tgt)
    c.nn_pass(np.eye(4), 0.2)
    assert c.reduce()[0] == 0.0
    c.set_clouds_f64(src, np.zeros((0, 3)))
    c.nn_pass(np.eye(4), 0.2)
    assert c.reduce()[0] == 0.0 and np.all(c.correspondence_index() == -1)
    c.set_clouds_f64(src[:1], tgt[:1] * 0.0 + src[:1] + 0.01)
    c.nn_pass(np.eye(4), 0.2)
    st = c.reduce()
    assert st[0] == 1.0 and c.correspondence_index()[0] == 0 and abs(st[1] - 3 * 0.01 ** 2) < 1e-12
    # and a registration over a real pair afterwards is the one a fresh context computes
    c.set_clouds_f64(src, tgt)
    a = c.run(None, 0.2, 5)
    d = _lib.Context(0)
    d.set_ring_search(1)
    d.set_nn_mode(lib.NN_GRID)
    d.set_clouds_f64(src, tgt)
    b = d.run(None, 0.2, 5)
    assert c.search_kernel_used() == "ring" and np.array_equal(a.transformation_, b.transformation_)
