"""The certificate of the warm-started search (visma_amd/csrc/grid_coop.hip, round 4): a query whose previous
winner provably cannot have changed (triangle inequality on the motion since the pass that left a lower bound of the
runner-up distance) skips the search.  It must never change a result: correspondences, distances and all 38
statistics BIT for bit against the lane-serial kernel and against the same library with VISMA_ICP_CERT=0, along
decaying motions (what ICP does), with partial overlap (queries WITHOUT a partner certify against the radius),
duplicated points (no certificate: exact ties), a source that leaves and re-enters the radius; and it must really
fire (most queries of a converged pass), or the test would pass on a kernel that never certifies."""
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


def small_T(rng, r, m):
    """a rigid motion that moves points of a unit-sized cloud by about m radii"""
    return synth.make_T(synth.rot_y(rng.uniform(-1, 1) * m * r * 0.5) @ synth.rot_x(rng.uniform(-1, 1) * m * r * 0.3),
                        rng.standard_normal(3) * r * m * 0.5)


SERIAL = {"VISMA_ICP_COOP": "0", "VISMA_ICP_GRID_LANES": "801"}
CASES = [
    # name, ns, nt, radius scale, fraction of the source displaced out of the target's reach, duplicates
    ("5k-20k", 5000, 20000, 1.0, 0.0, 0),
    ("partial overlap", 30000, 120000, 1.0, 0.5, 0),
    ("all outside", 4000, 20000, 1.0, 1.0, 0),
    ("duplicates", 4000, 30000, 1.0, 0.0, 2),
    ("64k-1M", 65536, 1048576, 1.0, 0.3, 0),
    ("big radius", 3000, 8000, 6.0, 0.2, 0),
    ("several queries per lane", 600000, 1000000, 1.0, 0.1, 0),
]


@pytest.mark.parametrize("name,ns,nt,rscale,out_frac,dup", CASES, ids=[c[0] for c in CASES])
def test_certified_passes_equal_searched_passes_bit_for_bit(lib, name, ns, nt, rscale, out_frac, dup):
    rng = np.random.default_rng(ns * 7 + nt)
    src, tgt, T_gt, r = synth.make_pair(ns, nt, seed_t=ns + 1, seed_s=nt + 3, motion="radius")
    r *= rscale
    if out_frac > 0:
        # part of the source displaced: some points far outside the target's bounding box, some just outside the
        # radius of anything (these certify "still no partner" only while the motion is small)
        k = int(ns * out_frac)
        sel = rng.permutation(ns)[:k]
        src = src.copy()
        src[sel[: k // 2]] += np.array([5.0, 0.0, 0.0])
        src[sel[k // 2:]] += rng.standard_normal((k - k // 2, 3)) * 3.0 * r
    if dup:
        tgt = np.concatenate([tgt] + [tgt[rng.permutation(len(tgt))[: len(tgt) // 2]] for _ in range(dup)])
    ref = ctx_env(SERIAL)
    nocert = ctx_env({"VISMA_ICP_CERT": "0"})
    c = _lib.Context(0)
    for x in (ref, nocert, c):
        x.set_nn_mode(lib.NN_GRID)
        x.set_ring_search(0)       # (this test is about the certificate kernel: "big radius" would go to grid_ring.hip by itself)
        x.set_clouds_f64(src, tgt)
    c.set_profiling(1)
    # ICP-like: a pose near the truth, then motions decaying geometrically, a standstill, one jump, decay again
    motions = [None, 0.5, 0.2, 0.08, 0.03, 0.01, 0.003, 0.0, 0.0, 2.5, 0.05, 0.01, 0.001, 0.0]
    T = T_gt @ small_T(rng, r, 0.7)
    certified = []
    for p, m in enumerate(motions):
        if m is not None:
            T = small_T(rng, r, m) @ T
        outs = []
        for x in (ref, nocert, c):
            x.nn_pass(T, r)
            st = x.reduce()
            outs.append((x.correspondence_index(), x.get_correspondences()[2].view(np.uint32), st.view(np.uint64)))
        tm = c.get_timing(reset=True)
        certified.append(tm["grid_certified"] / ns)
        for which, o in (("cert=0", outs[1]), ("default", outs[2])):
            assert np.array_equal(o[0], outs[0][0]), (name, which, p)
            assert np.array_equal(o[1], outs[0][1]), (name, which, p)
            if p > 0:                                             # (the cold pass may use more lanes per query)
                assert np.array_equal(o[2], outs[0][2]), (name, which, p)
    # pass 0 (lane-serial, nothing known) cannot certify; pass 1 may since round 5 (the lane-serial kernel leaves a bound
    # too: runner-up, skipped rows, outside of the 27 cells); a standstill after two decaying steps certifies nearly
    # everything that is not an exact tie
    assert certified[0] == 0.0, certified
    if dup:
        assert max(certified) < 0.9, certified                    # duplicated points are exact ties: never certified
    else:
        assert certified[8] > 0.9, certified
        assert certified[9] < certified[8], certified            # the jump
        assert certified[13] > 0.9, certified
    for x in (ref, nocert, c):
        x.close()


def test_registrations_with_and_without_certificates_are_identical(lib):
    """Whole registrations -- host loop, device loop, point-to-plane, the yaw sweep, a batch with own clouds -- give
    the same iterations, correspondences and (bit for bit) transformation with the certificate on and off."""
    src, tgt, T_gt, r = synth.make_pair(20000, 90000, seed_t=18, seed_s=19, motion="radius")
    src = src.copy()
    src[::3] += np.array([0.0, 3.0, 0.0])                         # a third of the source has no partner
    off = ctx_env({"VISMA_ICP_CERT": "0"})
    on = _lib.Context(0)
    nrm = on.estimate_normals(tgt, knn=12)
    for x in (off, on):
        x.set_clouds_f64(src, tgt)
        x.set_target_normals_f64(nrm)
    for loop in (False, True):
        for x in (off, on):
            x.set_device_loop(loop)
        a, b = off.run(None, r, 30, 1e-9, 1e-9), on.run(None, r, 30, 1e-9, 1e-9)
        assert a.num_correspondences == b.num_correspondences and a.iterations == b.iterations
        assert np.array_equal(np.asarray(a.transformation_), np.asarray(b.transformation_)), loop
        assert np.array_equal(off.correspondence_index(), on.correspondence_index())
        a, b = off.run_point_to_plane(None, r, 12, 0, 0), on.run_point_to_plane(None, r, 12, 0, 0)
        assert a.num_correspondences == b.num_correspondences
        assert np.array_equal(np.asarray(a.transformation_), np.asarray(b.transformation_)), loop
    for x in (off, on):
        x.set_device_loop(None)
    ba, la, pa = off.run_yaw_sweep(8, 3 * r, 15)
    bb, lb, pb = on.run_yaw_sweep(8, 3 * r, 15)
    assert la == lb
    for u, v in zip(pa, pb):
        assert u.num_correspondences == v.num_correspondences and u.iterations == v.iterations
        assert np.array_equal(np.asarray(u.transformation_), np.asarray(v.transformation_))
    probs = []
    for k in range(5):
        s, t, _, rr = synth.make_pair(3000 + 700 * k, 9000 + 2000 * k, seed_t=20 + k, seed_s=40 + k, motion="radius")
        probs.append((s, t, np.eye(4), rr))
    ga, gb = off.run_batch(probs, max_iter=15), on.run_batch(probs, max_iter=15)
    for u, v in zip(ga, gb):
        assert u.num_correspondences == v.num_correspondences and u.iterations == v.iterations
        assert np.array_equal(np.asarray(u.transformation_), np.asarray(v.transformation_))
    off.close(); on.close()
