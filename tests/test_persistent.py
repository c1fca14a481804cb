"""The PERSISTENT launch of a host loop (visma_amd/csrc/grid_coop.hip: nn_coop_kernel_persist): one
launch of the certificate kernel runs the remaining passes of visma_icp_run / visma_icp_iterate, the next transform
handed over through a command block in mapped host memory.  It must never change a result -- transformation,
correspondences, distances, iteration counts BIT for bit against one launch per pass --, it must really run (or the
test would pass on a library that never starts one), it must end when the loop stops early, start again for the next
loop, refuse what it cannot do (several queries per lane), and when its host does not come back in time it must end by
itself and the registration must carry on with ordinary launches to the same result."""
import numpy as np
import pytest

from visma_amd import _lib, synth

pytestmark = pytest.mark.gpu


def pair_of_contexts(src, tgt, normals=False):
    per_pass, persist = _lib.Context(0), _lib.Context(0)
    per_pass.set_persistent(False)
    persist.set_persistent(True)
    nrm = persist.estimate_normals(tgt, knn=12) if normals else None
    for c in (per_pass, persist):
        c.set_nn_mode(_lib.NN_GRID)
        c.set_ring_search(0)       # (these tests are about the persistent launch of the certificate kernel: a larger radius --
                                   #  20 points per radius-sized cell and more -- would send a registration to grid_ring.hip)
        c.set_clouds_f64(src, tgt)
        if normals:
            c.set_target_normals_f64(nrm)
    return per_pass, persist


def same_result(a, b):
    assert np.array_equal(a.transformation_, b.transformation_)
    assert a.num_correspondences == b.num_correspondences and a.iterations == b.iterations
    assert a.fitness_ == b.fitness_ and a.inlier_rmse_ == b.inlier_rmse_


CASES = [
    # name, ns, nt, fraction of the source without a partner, persistent launches expected
    ("5k-20k", 5000, 20000, 0.0, True),
    ("partial overlap", 30000, 120000, 0.5, True),
    ("all outside", 4000, 20000, 1.0, True),
    ("64k-1M", 65536, 1048576, 0.3, True),
    ("262144-1M: every compute unit", 262144, 1048576, 0.1, True),
    ("several queries per lane", 600000, 1000000, 0.1, False),
]


@pytest.mark.parametrize("name,ns,nt,out_frac,expect", CASES, ids=[c[0] for c in CASES])
def test_persistent_loop_equals_one_launch_per_pass_bit_for_bit(name, ns, nt, out_frac, expect):
    rng = np.random.default_rng(ns + 3 * nt)
    src, tgt, T_gt, r = synth.make_pair(ns, nt, seed_t=ns + 5, seed_s=nt + 7, motion="radius")
    if out_frac > 0:
        k = int(ns * out_frac)
        sel = rng.permutation(ns)[:k]
        src = src.copy()
        src[sel[: k // 2]] += np.array([5.0, 0.0, 0.0])
        src[sel[k // 2:]] += rng.standard_normal((k - k // 2, 3)) * 3.0 * r
    a, b = pair_of_contexts(src, tgt)
    b.set_profiling(1)
    b.get_timing(reset=True)
    # fixed iterations: 13 from the identity, then three blocks that carry on (each block: one persistent launch)
    Ta, Tb = np.eye(4), np.eye(4)
    for steps in (13, 5, 2, 9):
        Ta, ra = a.iterate(Ta, r, steps)
        Tb, rb = b.iterate(Tb, r, steps)
        assert np.array_equal(Ta, Tb), (name, steps)
        same_result(ra, rb)
        for x, y in zip(a.get_correspondences(), b.get_correspondences()):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (name, steps)
    tm = b.get_timing(reset=True)
    if expect:
        # 13 steps: the first is the cold pass, 12 in ONE launch (sources of 131,072 points and more: the cold pass runs
        # inside the launch too, round 5); the other blocks: one launch each
        import os
        cold_inside = 1 if (ns >= 131072 and os.environ.get("VISMA_ICP_COLD_IN_LAUNCH") == "1") else 0
        assert tm["persist_launches"] == 4 and tm["persist_passes"] == cold_inside + 12 + 5 + 2 + 9, tm
        assert tm["persist_aborts"] == 0
    else:
        assert tm["persist_launches"] == 0 and tm["nn_launches"] == 29, tm
    # whole registrations with the stop test on the host: the loop ends before its budget (STOP to the launch), the next
    # registration starts its own; a radius change in between
    for radius, tol in ((r, 1e-6), (1.7 * r, 1e-6), (r, 1e-3), (r, 0.0)):
        for c in (a, b):
            c.forget_winners()
        same_result(a.run(None, radius, 40, tol, tol), b.run(None, radius, 40, tol, tol))
        assert np.array_equal(a.correspondence_index(), b.correspondence_index())
    tm = b.get_timing(reset=True)
    assert (tm["persist_launches"] == 4) == expect and tm["persist_aborts"] == 0, tm
    # the per-pass API between loops is untouched by all this
    for c in (a, b):
        c.nn_pass(T_gt, r)
    assert np.array_equal(a.reduce().view(np.uint64), b.reduce().view(np.uint64))
    a.close()
    b.close()


def test_command_block_in_host_memory_is_polled_by_the_publisher_and_relayed():
    """Without a large PCIe BAR (forced here: VISMA_ICP_PERSIST_CMD=host) the command block lies in mapped host memory:
    only the workgroup that published polls it, the others take the words from its relay in device memory."""
    import os
    src, tgt, T_gt, r = synth.make_pair(65536, 400000, seed_t=68, seed_s=69, motion="radius")
    a = _lib.Context(0)
    a.set_persistent(False)
    old = os.environ.get("VISMA_ICP_PERSIST_CMD")
    os.environ["VISMA_ICP_PERSIST_CMD"] = "host"
    try:
        b = _lib.Context(0)
    finally:
        if old is None:
            os.environ.pop("VISMA_ICP_PERSIST_CMD", None)
        else:
            os.environ["VISMA_ICP_PERSIST_CMD"] = old
    for c in (a, b):
        c.set_nn_mode(_lib.NN_GRID)
        c.set_clouds_f64(src, tgt)
    b.set_profiling(1)
    b.get_timing(reset=True)
    same_result(a.run(None, r, 30, 0.0, 0.0), b.run(None, r, 30, 0.0, 0.0))
    Ta, _ = a.iterate(None, r, 11)
    Tb, _ = b.iterate(None, r, 11)
    assert np.array_equal(Ta, Tb)
    assert np.array_equal(a.correspondence_index(), b.correspondence_index())
    tm = b.get_timing(reset=True)
    assert tm["persist_launches"] == 2 and tm["persist_passes"] == 30 + 11 and tm["persist_aborts"] == 0, tm
    a.close()
    b.close()


def test_persistent_point_to_plane_and_gauss_newton(lib):
    src, tgt, T_gt, r = synth.make_pair(20000, 90000, seed_t=28, seed_s=29, motion="radius")
    src = src.copy()
    src[::4] += np.array([0.0, 3.0, 0.0])
    a, b = pair_of_contexts(src, tgt, normals=True)
    b.set_profiling(1)
    b.get_timing(reset=True)
    same_result(a.run_point_to_plane(None, r, 30, 1e-9, 1e-9), b.run_point_to_plane(None, r, 30, 1e-9, 1e-9))
    for solver in (lib.SOLVER_KABSCH, lib.SOLVER_GN_EULER, lib.SOLVER_GN_EXPMAP):
        for c in (a, b):
            c.forget_winners()
        same_result(a.run(None, r, 25, 1e-9, 1e-9, solver), b.run(None, r, 25, 1e-9, 1e-9, solver))
    tm = b.get_timing(reset=True)
    assert tm["persist_launches"] == 4 and tm["persist_aborts"] == 0, tm
    a.close()
    b.close()


def test_a_launch_whose_host_stalls_ends_by_itself_and_the_loop_carries_on():
    src, tgt, T_gt, r = synth.make_pair(40000, 200000, seed_t=38, seed_s=39, motion="radius")
    a, b = pair_of_contexts(src, tgt)
    b.set_persistent(True, timeout_ms=5.0)
    b.get_timing(reset=True)
    ra = a.run(None, r, 30, 0.0, 0.0)
    b.test_stall_command(7, 60.0)                  # the 7th command comes 60 ms late: the launch waits 5 and leaves
    rb = b.run(None, r, 30, 0.0, 0.0)
    same_result(ra, rb)
    assert np.array_equal(a.correspondence_index(), b.correspondence_index())
    tm = b.get_timing(reset=True)
    assert tm["persist_aborts"] == 1, tm
    # persistent launches are off on this context from then on (until asked for again) ...
    b.set_profiling(1)
    for c in (a, b):
        c.forget_winners()                         # (both start with the cold pass: its summation tree is another)
    same_result(a.run(None, r, 12, 0.0, 0.0), b.run(None, r, 12, 0.0, 0.0))
    assert b.get_timing(reset=True)["persist_launches"] == 0
    # ... and work again when they are
    b.set_persistent(True, timeout_ms=200.0)
    for c in (a, b):
        c.forget_winners()
    same_result(a.run(None, r, 12, 0.0, 0.0), b.run(None, r, 12, 0.0, 0.0))
    tm = b.get_timing(reset=True)
    assert tm["persist_launches"] == 1 and tm["persist_passes"] == 12 and tm["persist_aborts"] == 0, tm
    a.close()
    b.close()


def test_a_host_that_stalls_before_the_FIRST_command_of_an_early_launch():
    """Round 6: the persistent launch is queued behind the registration's cold pass and waits for its first transform like
    for every later one.  A host that comes back too late with that FIRST command posts STOP instead: the launch leaves
    without having run a pass (nothing written), the loop carries on with ordinary launches -- same registration, bit for
    bit, one abort on record."""
    src, tgt, T_gt, r = synth.make_pair(40000, 200000, seed_t=41, seed_s=42, motion="radius")
    a, b = pair_of_contexts(src, tgt)
    b.set_persistent(True, timeout_ms=5.0)
    for c in (a, b):
        c.forget_winners()                         # (both start with the cold pass)
    b.get_timing(reset=True)
    ra = a.run(None, r, 20, 0.0, 0.0)
    b.test_stall_command(1, 60.0)                  # the launch's first command comes 60 ms late
    rb = b.run(None, r, 20, 0.0, 0.0)
    same_result(ra, rb)
    assert np.array_equal(a.correspondence_index(), b.correspondence_index())
    tm = b.get_timing(reset=True)
    assert tm["persist_aborts"] == 1, tm
    # (and with the patience back, the next fresh registration runs all its warm passes inside ONE early launch)
    b.set_persistent(True, timeout_ms=200.0)
    b.set_profiling(1)
    for c in (a, b):
        c.forget_winners()
    same_result(a.run(None, r, 12, 0.0, 0.0), b.run(None, r, 12, 0.0, 0.0))
    tm = b.get_timing(reset=True)
    assert tm["persist_aborts"] == 0 and tm["persist_launches"] == 1 and tm["persist_passes"] == 12, tm
    a.close()
    b.close()


def test_two_contexts_in_one_process_take_turns():
    """One persistent launch per device at a time: a second context whose loop starts while the first one's launch is
    alive (another host thread) runs ordinary launches -- both get the single-context results."""
    import threading
    src, tgt, T_gt, r = synth.make_pair(30000, 150000, seed_t=48, seed_s=49, motion="radius")
    ref = _lib.Context(0)
    ref.set_persistent(False)
    ref.set_nn_mode(_lib.NN_GRID)
    ref.set_clouds_f64(src, tgt)
    want = ref.run(None, r, 30, 0.0, 0.0)
    ctxs = [_lib.Context(0) for _ in range(3)]
    for c in ctxs:
        c.set_nn_mode(_lib.NN_GRID)
        c.set_clouds_f64(src, tgt)
    got = [None] * len(ctxs)

    def work(i):
        for _ in range(5):
            ctxs[i].forget_winners()
            got[i] = ctxs[i].run(None, r, 30, 0.0, 0.0)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(ctxs))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for g in got:
        same_result(want, g)
    for c in ctxs:
        assert c.get_timing()["persist_aborts"] == 0
        c.close()
    ref.close()


def _racing_process(rank, barrier, q):
    import numpy as np
    from visma_amd import _lib, synth
    src, tgt, T_gt, r = synth.make_pair(262144, 600000, seed_t=58, seed_s=59, motion="radius")
    c = _lib.Context(0)
    c.set_persistent(True, timeout_ms=20.0)
    c.set_nn_mode(_lib.NN_GRID)
    c.set_clouds_f64(src, tgt)
    c.run(None, r, 2, 0.0, 0.0)
    barrier.wait()
    out = None
    for _ in range(4):
        c.forget_winners()
        out = c.run(None, r, 25, 0.0, 0.0)
    tm = c.get_timing()
    q.put((rank, out.transformation_, out.num_correspondences, tm["persist_aborts"]))
    c.close()


def test_two_processes_whose_launches_want_every_compute_unit_both_finish():
    """Two processes start 1,024-workgroup persistent launches on the same GPU at the same moment: each may get a part of
    the compute units and wait for the rest.  Neither may hang: a launch that sees its own workgroups missing gives up
    after its patience, the registration restarts the pass cold with ordinary launches -- both end with the registration's
    result (to rounding: a cold pass sums along another tree)."""
    import multiprocessing as mp
    import time
    src, tgt, T_gt, r = synth.make_pair(262144, 600000, seed_t=58, seed_s=59, motion="radius")
    ref = _lib.Context(0)
    ref.set_persistent(False)
    ref.set_nn_mode(_lib.NN_GRID)
    ref.set_clouds_f64(src, tgt)
    want = ref.run(None, r, 25, 0.0, 0.0)
    ref.close()
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(2), ctx.Queue()
    ps = [ctx.Process(target=_racing_process, args=(i, barrier, q)) for i in range(2)]
    t0 = time.time()
    for p in ps:
        p.start()
    got = [q.get(timeout=240) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert time.time() - t0 < 200
    print("two racing processes: %.1f s, launches that gave up per process: %s" % (time.time() - t0, [g[3] for g in got]))
    for rank, T, k, aborts in got:
        assert k == want.num_correspondences
        assert synth.rel_frobenius(T, want.transformation_) < 1e-9, (rank, aborts)


def test_short_loops_and_tiny_clouds():
    """The budget edges of a host loop (0, 1, 2, 3 iterations: no launch to keep alive, a launch of one pass, ...) and clouds
    of a handful of points (one workgroup, lanes without a query)."""
    rng = np.random.default_rng(7)
    for ns, nt in ((1, 50), (3, 3), (70, 1000), (257, 5000)):
        tgt = rng.standard_normal((nt, 3))
        src = tgt[rng.integers(0, nt, ns)] + rng.standard_normal((ns, 3)) * 0.01
        a, b = pair_of_contexts(src, tgt)
        for max_iter in (0, 1, 2, 3, 7):
            for c in (a, b):
                c.forget_winners()
            same_result(a.run(None, 0.3, max_iter, 0.0, 0.0), b.run(None, 0.3, max_iter, 0.0, 0.0))
        Ta, Tb = np.eye(4), np.eye(4)
        for steps in (1, 1, 2, 3, 1):
            Ta, ra = a.iterate(Ta, 0.3, steps)
            Tb, rb = b.iterate(Tb, 0.3, steps)
            assert np.array_equal(Ta, Tb), (ns, nt, steps)
            same_result(ra, rb)
        assert b.get_timing()["persist_aborts"] == 0
        a.close()
        b.close()


def test_the_product_api_of_persistent_launches_share_cap_and_info(lib):
    """visma_icp.h since round 5: visma_icp_set_persistent / _set_persistent_cu_share / _get_persistent_info.  A process
    that may hold half of the device's workgroup slots runs a 262,144-point loop (all slots) one launch per pass and a
    65,536-point loop (a quarter) persistently -- same results either way; the info call says what happened."""
    src, tgt, T_gt, r = synth.make_pair(262144, 1048576, seed_t=91, seed_s=92, motion="radius")
    a, b = pair_of_contexts(src, tgt)
    info = b.persistent_info()
    assert info["enabled"] == 1 and info["launches"] == 0 and info["cu_share"] == 1.0 and info["timeout_ms"] == 200.0
    try:
        lib.set_persistent_cu_share(0.5)
        with pytest.raises(lib.IcpError):
            lib.set_persistent_cu_share(0.0)
        ra, rb = a.run(None, r, 12, 0.0, 0.0), b.run(None, r, 12, 0.0, 0.0)
        same_result(ra, rb)
        info = b.persistent_info()
        assert info["launches"] == 0 and info["last_loop_persistent"] == 0 and info["cu_share"] == 0.5, info
        assert info["device_slots"] >= 1024, info
        # a quarter of the slots: allowed
        for c in (a, b):
            c.set_clouds_f64(src[:65536], tgt)
        same_result(a.run(None, r, 12, 0.0, 0.0), b.run(None, r, 12, 0.0, 0.0))
        info = b.persistent_info()
        assert info["launches"] == 1 and info["last_loop_persistent"] == 1 and info["last_loop_passes"] == 12, info
        lib.set_persistent_cu_share(1.0)
        for c in (a, b):
            c.set_clouds_f64(src, tgt)
        same_result(a.run(None, r, 12, 0.0, 0.0), b.run(None, r, 12, 0.0, 0.0))
        info = b.persistent_info()
        assert info["launches"] == 2 and info["last_loop_passes"] == 12 and info["aborts"] == 0, info   # (the cold pass: its own launch)
        # switched off per context
        b.set_persistent(False)
        b.forget_winners()
        a.forget_winners()
        same_result(a.run(None, r, 5, 0.0, 0.0), b.run(None, r, 5, 0.0, 0.0))
        info = b.persistent_info()
        assert info["enabled"] == 0 and info["launches"] == 2 and info["last_loop_persistent"] == 0, info
    finally:
        lib.set_persistent_cu_share(1.0)
    a.close()
    b.close()


def test_a_loop_under_a_cu_share_leaves_room_for_a_batch_on_another_context(lib):
    """What the share is for: a C4-sized host loop and a batch of small registrations on two contexts of one process, at
    the same time.  With the whole device allowed, the loop's persistent launch holds every workgroup slot for its
    length and the batch's launches wait behind it; under a share of 1/2 the loop launches once per pass and the two
    interleave.  Both finish with the results of their solo runs; the wall times are reported."""
    import threading
    import time
    src, tgt, T_gt, r = synth.make_pair(262144, 1048576, seed_t=93, seed_s=94, motion="radius")
    loop = _lib.Context(0)
    loop.set_nn_mode(_lib.NN_GRID)
    loop.set_clouds_f64(src, tgt)
    probs = []
    for k in range(24):
        s, t, _, _ = synth.make_pair(6000 + 500 * (k % 5), 9000, seed_t=300 + k % 3, seed_s=400 + k)
        probs.append((s, t, np.eye(4), 0.02))
    batch = _lib.Context(0)

    def run_loop(out):
        t0 = time.perf_counter()
        for _ in range(6):
            loop.forget_winners()
            out["T"] = loop.run(None, r, 30, 0.0, 0.0).transformation_
        out["s"] = time.perf_counter() - t0

    def run_batch(out):
        t0 = time.perf_counter()
        for _ in range(4):
            out["res"] = [x.transformation_ for x in batch.run_batch(probs, max_iter=20)]
        out["s"] = time.perf_counter() - t0
    solo_l, solo_b = {}, {}
    run_loop({}); run_batch({})                    # warm-up (buffers, grids)
    run_loop(solo_l); run_batch(solo_b)
    report = {}
    try:
        for share in (1.0, 0.5):
            lib.set_persistent_cu_share(share)
            l, b = {}, {}
            th = [threading.Thread(target=run_loop, args=(l,)), threading.Thread(target=run_batch, args=(b,))]
            for t in th:
                t.start()
            for t in th:
                t.join()
            assert np.array_equal(l["T"], solo_l["T"])
            assert all(np.array_equal(x, y) for x, y in zip(b["res"], solo_b["res"]))
            report[share] = (l["s"] / solo_l["s"], b["s"] / solo_b["s"])
    finally:
        lib.set_persistent_cu_share(1.0)
    print("loop / batch wall time relative to solo: share 1.0 -> %.2f / %.2f, share 0.5 -> %.2f / %.2f"
          % (report[1.0] + report[0.5]))
    # (both make progress whatever the share; the ratios depend on the box -- see DESIGN.md 4.1e for the measured ones)
    assert all(v < 6.0 for pair in report.values() for v in pair), report
    loop.close()
    batch.close()


def test_cold_pass_inside_the_launch_is_the_same_registration(lib):
    """VISMA_ICP_COLD_IN_LAUNCH=1 (opt-in, round 5): sources of 131,072 points and more run their FIRST pass inside the
    persistent launch too -- the certificate kernel started cold.  Same registration bit for bit, one launch for all passes."""
    import os
    src, tgt, T_gt, r = synth.make_pair(150000, 600000, seed_t=95, seed_s=96, motion="radius")
    src = src.copy()
    src[::5] += np.array([0.0, 4.0, 0.0])                  # a fifth of the source without a partner
    a = _lib.Context(0)
    a.set_persistent(False)
    os.environ["VISMA_ICP_COLD_IN_LAUNCH"] = "1"
    try:
        b = _lib.Context(0)
    finally:
        os.environ.pop("VISMA_ICP_COLD_IN_LAUNCH", None)
    for c in (a, b):
        c.set_nn_mode(_lib.NN_GRID)
        c.set_clouds_f64(src, tgt)
    b.set_profiling(1)
    b.get_timing(reset=True)
    same_result(a.run(None, r, 14, 0.0, 0.0), b.run(None, r, 14, 0.0, 0.0))
    for x, y in zip(a.get_correspondences(), b.get_correspondences()):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    tm = b.get_timing(reset=True)
    assert tm["persist_launches"] == 1 and tm["persist_passes"] == 15 and tm["persist_aborts"] == 0, tm
    a.close()
    b.close()
