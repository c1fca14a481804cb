"""The persistent SWEEP launch (visma_amd/csrc/grid_wave.hip: nn_wave_kernel_sweep, round 6): after their first pass the
registrations of a yaw sweep -- feh::RegisterModelToScene's 24 starts over shared clouds (src/annotation.cpp:35-61) -- and
of a device-resident loop run ALL their remaining passes inside one launch: search, fold, closed-form update, compose and
stop test on the device, the next transform handed from the folding workgroup to the problem's others through tagged
words.  It must be the same registrations as one search launch + one solve launch per pass (VISMA_ICP_SWEEP_PERSIST=0),
bit for bit: every start's transformation, correspondence count, fitness, RMSE and iteration count, with and without the
stop test, for clouds of one workgroup and of hundreds; and it must really run (or the test would pass on a library that
never starts it)."""
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


def same(a, b, what):
    assert a.iterations == b.iterations, what
    assert a.num_correspondences == b.num_correspondences, what
    assert np.array_equal(np.asarray(a.transformation_), np.asarray(b.transformation_)), what
    assert a.fitness_ == b.fitness_ and a.inlier_rmse_ == b.inlier_rmse_, what


CASES = [
    # ns, nt, radius, level, (max_iter, rel_fitness, rel_rmse)
    (5000, 20000, 0.075, 24, (30, 1e-6, 1e-6)),        # config 2's sweep: the stop test ends starts at different passes
    (5000, 20000, 0.075, 24, (20, 0.0, 0.0)),          # fixed iterations
    (200, 3000, 0.1, 8, (15, 1e-6, 1e-6)),             # one workgroup per start
    (30000, 60000, 0.03, 7, (12, 0.0, 0.0)),           # 118 workgroups per start, 826 in the launch
    (4000, 9000, 0.02, 24, (30, 1e-6, 1e-6)),          # small radius: most queries without a partner
]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("ns,nt,r,level,crit", CASES, ids=["%dx%d-%d-%s" % (c[0], c[1], c[3], "stop" if c[4][1] else "fixed") for c in CASES])
def test_sweep_in_one_launch_equals_one_launch_per_pass(lib, ns, nt, r, level, crit):
    src, tgt, _, _ = synth.make_pair(ns, nt, seed_t=ns + 7, seed_s=nt + 9)
    per_pass = ctx_env({"VISMA_ICP_SWEEP_PERSIST": "0"})
    c = _lib.Context(0)
    for x in (per_pass, c):
        x.set_clouds_f64(src, tgt)
    for rep in range(2):                               # (the second sweep re-uses relay and tags)
        bw, ww, pw = per_pass.run_yaw_sweep(level, r, *crit)
        bg, wg, pg = c.run_yaw_sweep(level, r, *crit)
        assert wg == ww
        same(bg, bw, "best")
        for k, (a, b) in enumerate(zip(pg, pw)):
            same(a, b, (rep, k))
    info = c.sweep_info()
    assert info["launches"] >= 2 and info["aborts"] == 0, info
    assert per_pass.sweep_info()["launches"] == 0
    per_pass.close()
    c.close()


@pytest.mark.timeout(900)
def test_device_loop_of_one_registration_in_one_launch(lib):
    src, tgt, T_gt, r = synth.make_pair(20000, 80000, seed_t=21, seed_s=22, motion="radius")
    per_pass = ctx_env({"VISMA_ICP_SWEEP_PERSIST": "0"})
    c = _lib.Context(0)
    for x in (per_pass, c):
        x.set_device_loop(True)
        x.set_clouds_f64(src, tgt)
    for crit in ((25, 0.0, 0.0), (40, 1e-6, 1e-6)):
        a, b = c.run(None, r, *crit), per_pass.run(None, r, *crit)
        same(a, b, crit)
        assert np.array_equal(c.correspondence_index(), per_pass.correspondence_index())
    assert c.sweep_info()["launches"] >= 2
    per_pass.close()
    c.close()


@pytest.mark.timeout(900)
def test_a_sweep_too_large_for_the_device_runs_as_before(lib):
    """24 starts of a 40,000-point source = 3,768 workgroups: more than the device holds at once -- one launch per pass"""
    src, tgt, _, _ = synth.make_pair(40000, 60000, seed_t=31, seed_s=32)
    c = _lib.Context(0)
    c.set_clouds_f64(src, tgt)
    best, which, per = c.run_yaw_sweep(24, 0.03, 6, 0.0, 0.0)
    assert c.sweep_info()["launches"] == 0 and all(p.iterations == 6 for p in per)
    c.close()
