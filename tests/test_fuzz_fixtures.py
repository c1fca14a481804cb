"""Random registrations whose expected results come from the COMPILED REFERENCE
(tests/golden/fuzz_ref.npz, written by tests/golden/gen_fuzz.py from oracle/_ref).

GPU: every case, every search path that is exact by construction (single runs on the grid, the
batched device loop): correspondence count EQUAL to the reference's, transformation within 1e-9.
CPU: the oracle restatement against the same fixtures."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from gen_fuzz import FIELDS, make_case  # noqa: E402

from visma_amd import _lib, synth  # noqa: E402

G = np.load(os.path.join(HERE, "golden", "fuzz_ref.npz"))
N = len(G["ns"])


def case(i):
    return make_case({k: G[k][i] for k in FIELDS})


def test_fixture_set_is_large_enough():
    assert N >= 50
    assert (G["cut"] > 0).sum() >= 10          # partial overlaps are in


def test_oracle_restatement_matches_the_reference_on_the_fuzz_cases(oracle):
    small = np.argsort(G["ns"] * G["nt"])[:10]
    for i in small:
        src, tgt, init, r, iters = case(int(i))
        o = oracle.registration_icp(src, tgt, r, init=init, max_iter=iters)
        assert o.k == int(G["ref_k"][i]), i
        assert synth.rel_frobenius(o.T, G["ref_T"][i]) < 1e-9, i


@pytest.mark.gpu
def test_every_fuzz_case_matches_the_reference_exactly(lib):
    ctx = _lib.Context(0)
    worst = 0.0
    for i in range(N):
        src, tgt, init, r, iters = case(i)
        ctx.set_clouds_f64(src, tgt)
        got = ctx.run(init, r, iters, 1e-6, 1e-6)
        assert ctx.search_mode_used() == "exact"
        assert got.num_correspondences == int(G["ref_k"][i]), (i, got.num_correspondences, int(G["ref_k"][i]))
        e = synth.rel_frobenius(got.transformation_, G["ref_T"][i])
        worst = max(worst, e)
        assert e < 1e-9, (i, e)
        assert abs(got.fitness_ - float(G["ref_fitness"][i])) < 1e-12
        assert abs(got.inlier_rmse_ - float(G["ref_rmse"][i])) < 1e-9 * max(1.0, float(G["ref_rmse"][i]))
    print("worst rel-Frobenius over %d cases: %.3e" % (N, worst))


@pytest.mark.gpu
def test_fuzz_cases_batched_in_flight_match_too(lib):
    """The batched device loop (BASELINE config 3) on 24 of the cases at once."""
    pick = list(range(0, N, max(1, N // 24)))[:24]
    probs, want = [], []
    for i in pick:
        src, tgt, init, r, iters = case(i)
        probs.append((src, tgt, init, r))
        want.append(i)
    # one iteration budget for the whole batch: re-run the reference semantics per case is not possible,
    # so compare against single runs of the library (already pinned above) at a common budget
    ctx = _lib.Context(0)
    got = ctx.run_batch(probs, max_iter=12, rel_fitness=1e-6, rel_rmse=1e-6)
    one = _lib.Context(0)
    one.set_device_loop(False)
    for g, (src, tgt, init, r) in zip(got, probs):
        one.set_clouds_f64(src, tgt)
        w = one.run(init, r, 12, 1e-6, 1e-6)
        assert g.num_correspondences == w.num_correspondences
        assert g.iterations == w.iterations
        assert synth.rel_frobenius(g.transformation_, w.transformation_) < 1e-11
