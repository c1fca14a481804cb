"""The brute-force kernels in their exact flavour (default precision + VISMA_ICP_NN_BRUTE): the
north_star kernel -- LDS-tiled all-pairs search -- must return the reference's correspondences
too, not only the grid search.  fp32 ranking; the winning 64-target sub-chunk and, when another
sub-chunk reaches into the rounding band, the whole target are re-ranked in f64."""
import os
import sys

import numpy as np
import pytest

from visma_amd import _lib, synth

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from gen_fuzz import FIELDS, make_case  # noqa: E402

pytestmark = pytest.mark.gpu


def pair(nn):
    c = _lib.Context(0)
    c.set_nn_mode(nn)
    return c


@pytest.mark.parametrize("ns,nt,radius", [(5000, 20000, None), (3000, 8000, 0.075), (2000, 500, 0.2), (20000, 70000, None)])
def test_brute_exact_equals_grid_exact_pass_by_pass(lib, ns, nt, radius):
    src, tgt, T_gt, r = synth.make_pair(ns, nt, seed_t=ns, seed_s=nt, motion="radius")
    tgt = np.concatenate([tgt, tgt[:200]])                        # exact duplicates: ties resolved by index
    r = radius or r
    g, b = pair(_lib.NN_GRID), pair(_lib.NN_BRUTE)
    g.set_clouds_f64(src, tgt)
    b.set_clouds_f64(src, tgt)
    rng = np.random.default_rng(ns)
    for k in range(3):
        T = np.eye(4) if k == 0 else T_gt @ synth.make_T(synth.rot_y(rng.uniform(-r, r)), rng.standard_normal(3) * r * 0.4)
        g.nn_pass(T, r); sg = g.reduce(); ig = g.correspondence_index()
        b.nn_pass(T, r); sb = b.reduce(); ib = b.correspondence_index()
        assert b.search_mode_used() == "exact" and g.search_mode_used() == "exact"
        assert np.array_equal(ig, ib)
        assert sg[0] == sb[0] and np.max(np.abs(sg - sb)) <= 1e-11 * np.max(np.abs(sg))
    a, c = g.run(None, r, 8, 0, 0), b.run(None, r, 8, 0, 0)
    assert a.num_correspondences == c.num_correspondences
    assert synth.rel_frobenius(a.transformation_, c.transformation_) < 1e-12
    b.set_device_loop(True)                                       # the on-device loop runs the same kernels
    d = b.run(None, r, 8, 0, 0)
    assert d.num_correspondences == a.num_correspondences
    assert synth.rel_frobenius(d.transformation_, a.transformation_) < 1e-12


def test_brute_exact_reproduces_the_reference_fuzz_cases(lib):
    G = np.load(os.path.join(HERE, "golden", "fuzz_ref.npz"))
    ctx = pair(_lib.NN_BRUTE)
    for i in range(0, len(G["ns"]), 4):
        src, tgt, init, r, iters = make_case({k: G[k][i] for k in FIELDS})
        ctx.set_clouds_f64(src, tgt)
        got = ctx.run(init, r, iters, 1e-6, 1e-6)
        assert got.num_correspondences == int(G["ref_k"][i]), i
        assert synth.rel_frobenius(got.transformation_, G["ref_T"][i]) < 1e-9, i


def test_more_undecided_queries_than_the_rescan_list_holds(lib):
    """A target given twice: EVERY query has two equally near targets in different sub-chunks, so every query
    is left to the rescan passes; past their capacity (4096) the wave scans in place.  Either way the winner
    is the lower index."""
    src, tgt, T_gt, r = synth.make_pair(6000, 6000, seed_t=77, seed_s=78, motion="radius")
    tgt2 = np.concatenate([tgt, tgt])
    g, b = pair(_lib.NN_GRID), pair(_lib.NN_BRUTE)
    g.set_clouds_f64(src, tgt2)
    b.set_clouds_f64(src, tgt2)
    for T in (np.eye(4), T_gt):
        g.nn_pass(T, r); sg = g.reduce(); ig = g.correspondence_index()
        b.nn_pass(T, r); sb = b.reduce(); ib = b.correspondence_index()
        assert b.nn_mode_used() == _lib.NN_BRUTE and b.search_mode_used() == "exact"
        assert (ib[ib >= 0] < len(tgt)).all()                     # the first copy wins every tie
        assert np.array_equal(ig, ib)
        assert sg[0] == sb[0] and sg[0] > 4096 and np.max(np.abs(sg - sb)) <= 1e-11 * np.max(np.abs(sg))
