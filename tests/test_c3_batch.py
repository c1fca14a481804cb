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
