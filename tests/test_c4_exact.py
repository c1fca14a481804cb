"""BASELINE config 4 at FULL size (262,144 -> 4,194,304) through the DEFAULT (exact) kernel -- the
instantiation bench.py times -- against the COMPILED REFERENCE's own output
(tests/golden/c4_ref.npz, written by tests/golden/gen_c4.py from oracle/_ref: Open3D's
RegistrationICP / EvaluateRegistration with KDTreeFlann, Registration.cpp:41-186).

Bars: K equal, the correspondence set equal (two checksums over the index array), fitness equal,
rmse within 1e-12 relative, transformation within 1e-9 relative Frobenius after 10 iterations."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import gen_c4  # noqa: E402

from visma_amd import _lib, synth  # noqa: E402

G = np.load(os.path.join(HERE, "golden", "c4_ref.npz"))


@pytest.fixture(scope="module")
def c4():
    src, tgt, T_gt, r = gen_c4.clouds()
    return src, tgt, r


def test_fixture_inputs_regenerate_bit_for_bit(c4):
    """The fixture stores a recipe, not 100 MB of points: the clouds the GPU test feeds the kernel
    must be the ones the reference saw."""
    src, tgt, r = c4
    assert src.shape == (int(G["ns"]), 3) and tgt.shape == (int(G["nt"]), 3)
    assert gen_c4.input_checksum(src) == int(G["src_checksum"])
    assert gen_c4.input_checksum(tgt) == int(G["tgt_checksum"])
    assert r == float(G["radius"])


def test_checksum_notices_one_changed_or_swapped_partner():
    idx = np.arange(1000, dtype=np.int32)[::-1].copy()
    a = gen_c4.checksum(idx)
    j = idx.copy(); j[[3, 700]] = j[[700, 3]]
    assert gen_c4.checksum(j)[0] == a[0] and gen_c4.checksum(j)[1] != a[1]
    j = idx.copy(); j[5] += 1
    assert gen_c4.checksum(j) != a
    j = idx.copy(); j[17] = -1
    assert gen_c4.checksum(j)[2] == a[2] - 1


@pytest.fixture(scope="module")
def ctx(lib):
    c = _lib.Context(0)                       # default search: exact
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nn", ["grid", "brute"])
def test_c4_one_pass_equals_the_reference_evaluation(lib, ctx, c4, nn):
    """EvaluateRegistration (Registration.cpp:98-116) at a pose off the identity: the reference's own
    K, correspondence set, fitness and rmse -- from the exact kernel at the size the bench times."""
    src, tgt, r = c4
    ctx.set_nn_mode({"grid": lib.NN_GRID, "brute": lib.NN_BRUTE}[nn])
    ctx.set_clouds_f64(src, tgt)
    ctx.nn_pass(G["eval_T"], r)
    st = ctx.reduce()
    assert ctx.search_mode_used() == "exact"
    idx = ctx.correspondence_index()
    ctx.set_nn_mode(lib.NN_AUTO)
    s1, s2, k = gen_c4.checksum(idx)
    assert k == int(G["eval_k"]) == int(st[0])
    assert (s1, s2) == (int(G["eval_sum"]), int(G["eval_wsum"]))
    assert k / len(src) == float(G["eval_fitness"])
    rmse = np.sqrt(st[1] / st[0])
    assert abs(rmse - float(G["eval_rmse"])) < 1e-12 * float(G["eval_rmse"])


@pytest.mark.gpu
@pytest.mark.parametrize("loop", ["host", "device"])
def test_c4_ten_iterations_equal_the_reference_run(lib, ctx, c4, loop):
    """RegistrationICP (Registration.cpp:141-186) from identity, criteria (0, 0, 10)."""
    src, tgt, r = c4
    ctx.set_nn_mode(lib.NN_AUTO)
    ctx.set_device_loop(loop == "device")
    ctx.set_clouds_f64(src, tgt)
    got = ctx.run(np.eye(4), r, int(G["iters"]), 0.0, 0.0)
    ctx.set_device_loop(None)
    assert ctx.search_mode_used() == "exact"
    assert ctx.nn_mode_used() == lib.NN_GRID            # the kernel bench.py times
    assert got.num_correspondences == int(G["ref_k"])
    s1, s2, k = gen_c4.checksum(ctx.correspondence_index())
    assert (s1, s2, k) == (int(G["ref_sum"]), int(G["ref_wsum"]), int(G["ref_k"]))
    e = synth.rel_frobenius(got.transformation_, G["ref_T"])
    print("C4 exact kernel vs compiled reference after %d iterations: %.3e" % (int(G["iters"]), e))
    assert e < 1e-9
    assert got.fitness_ == float(G["ref_fitness"])
    assert abs(got.inlier_rmse_ - float(G["ref_rmse"])) < 1e-12 * float(G["ref_rmse"])


@pytest.mark.gpu
def test_c4_exact_search_is_bit_identical_to_the_all_f64_search(lib, ctx, c4):
    """fp32 ranking + f64 re-rank of the rounding band returns what an all-f64 scan
    returns -- correspondences, distances and all 38 statistics, bit for bit, at C4."""
    src, tgt, r = c4
    f64 = _lib.Context(0)
    f64.set_search_precision("f64")
    out = {}
    for name, c in (("exact", ctx), ("f64", f64)):
        c.set_nn_mode(lib.NN_GRID)
        c.set_clouds_f64(src, tgt)
        c.nn_pass(G["eval_T"], r)
        st = c.reduce()
        assert c.search_mode_used() == name
        out[name] = (c.correspondence_index(), c.get_correspondences()[2], st)
        c.set_nn_mode(lib.NN_AUTO)
    f64.close()
    assert np.array_equal(out["exact"][0], out["f64"][0])
    assert np.array_equal(out["exact"][1].view(np.uint32), out["f64"][1].view(np.uint32))
    assert np.array_equal(out["exact"][2].view(np.uint64), out["f64"][2].view(np.uint64))


# ---- the partial-overlap variant (bench.py: `partial_overlap`): the whole model against a scan of half of its surface.
# Half of the 262,144 queries have NO partner within the radius: the reject path at full size, every pass
# (tests/golden/c4_partial_ref.npz, written by `gen_c4.py --partial` from the compiled reference).
GP = np.load(os.path.join(HERE, "golden", "c4_partial_ref.npz"))


@pytest.fixture(scope="module")
def c4p():
    src, tgt, T_gt, r = gen_c4.partial_clouds()
    return src, tgt, r


def test_partial_fixture_inputs_regenerate_bit_for_bit(c4p):
    src, tgt, r = c4p
    assert src.shape == (int(GP["ns"]), 3) and tgt.shape == (int(GP["nt"]), 3)
    assert gen_c4.input_checksum(src) == int(GP["src_checksum"])
    assert gen_c4.input_checksum(tgt) == int(GP["tgt_checksum"])
    assert r == float(GP["radius"])
    assert 0.45 < float(GP["ref_fitness"]) < 0.55             # what the variant is for


@pytest.mark.gpu
def test_c4_partial_one_pass_equals_the_reference_evaluation(lib, ctx, c4p):
    src, tgt, r = c4p
    ctx.set_nn_mode(lib.NN_GRID)
    ctx.set_clouds_f64(src, tgt)
    ctx.nn_pass(GP["eval_T"], r)
    st = ctx.reduce()
    idx = ctx.correspondence_index()
    ctx.set_nn_mode(lib.NN_AUTO)
    s1, s2, k = gen_c4.checksum(idx)
    assert k == int(GP["eval_k"]) == int(st[0])
    assert (s1, s2) == (int(GP["eval_sum"]), int(GP["eval_wsum"]))
    assert k / len(src) == float(GP["eval_fitness"])
    rmse = np.sqrt(st[1] / st[0])
    assert abs(rmse - float(GP["eval_rmse"])) < 1e-12 * float(GP["eval_rmse"])


@pytest.mark.gpu
@pytest.mark.parametrize("loop", ["host", "device"])
def test_c4_partial_ten_iterations_equal_the_reference_run(lib, ctx, c4p, loop):
    """RegistrationICP from identity, criteria (0, 0, 10): passes 2..11 run the warm-started kernel -- certificates for
    the matched queries AND for the ones that stay without a partner, the compacted search for the rest."""
    src, tgt, r = c4p
    ctx.set_nn_mode(lib.NN_AUTO)
    ctx.set_device_loop(loop == "device")
    ctx.set_clouds_f64(src, tgt)
    got = ctx.run(np.eye(4), r, int(GP["iters"]), 0.0, 0.0)
    ctx.set_device_loop(None)
    assert ctx.search_mode_used() == "exact" and ctx.nn_mode_used() == lib.NN_GRID
    assert ctx.search_kernel_used() == "warm"
    assert got.num_correspondences == int(GP["ref_k"])
    s1, s2, k = gen_c4.checksum(ctx.correspondence_index())
    assert (s1, s2, k) == (int(GP["ref_sum"]), int(GP["ref_wsum"]), int(GP["ref_k"]))
    e = synth.rel_frobenius(got.transformation_, GP["ref_T"])
    print("C4 partial overlap vs compiled reference after %d iterations: %.3e" % (int(GP["iters"]), e))
    assert e < 1e-9
    assert got.fitness_ == float(GP["ref_fitness"])
    assert abs(got.inlier_rmse_ - float(GP["ref_rmse"])) < 1e-12 * float(GP["ref_rmse"])


# ---- SURVEY 8d's LITERAL ground truth (bench.py: `literal_T_gt`): T_gt = R_y(5 deg) R_x(1 deg), t = (0.02, -0.01, 0.015) at
# C4's sizes with the radius that motion needs, 0.15 m -- 3,200 points per occupied radius-sized cell: the library goes to
# the ring search over cells of a few point spacings by itself (visma_amd/csrc/grid_ring.hip, round 6)
# (tests/golden/c4_literal_ref.npz, written by `gen_c4.py --literal` from the compiled reference).
GL = np.load(os.path.join(HERE, "golden", "c4_literal_ref.npz"))


@pytest.fixture(scope="module")
def c4l():
    src, tgt, T_gt, r = gen_c4.literal_clouds()
    return src, tgt, r


def test_literal_fixture_inputs_regenerate_bit_for_bit(c4l):
    src, tgt, r = c4l
    assert src.shape == (int(GL["ns"]), 3) and tgt.shape == (int(GL["nt"]), 3)
    assert gen_c4.input_checksum(src) == int(GL["src_checksum"])
    assert gen_c4.input_checksum(tgt) == int(GL["tgt_checksum"])
    assert r == float(GL["radius"]) == 0.15
    assert float(GL["ref_fitness"]) == 1.0 and float(GL["eval_rmse"]) > 0.02      # (nearest neighbours centimetres away)


@pytest.mark.gpu
def test_c4_literal_one_pass_equals_the_reference_evaluation(lib, ctx, c4l):
    src, tgt, r = c4l
    ctx.set_nn_mode(lib.NN_AUTO)
    ctx.set_clouds_f64(src, tgt)
    ctx.nn_pass(GL["eval_T"], r)
    st = ctx.reduce()
    idx = ctx.correspondence_index()
    assert ctx.search_kernel_used() == "ring" and ctx.ring_search()["occupancy"] > 1000
    s1, s2, k = gen_c4.checksum(idx)
    assert k == int(GL["eval_k"]) == int(st[0])
    assert (s1, s2) == (int(GL["eval_sum"]), int(GL["eval_wsum"]))
    assert k / len(src) == float(GL["eval_fitness"])
    rmse = np.sqrt(st[1] / st[0])
    assert abs(rmse - float(GL["eval_rmse"])) < 1e-12 * float(GL["eval_rmse"])


@pytest.mark.gpu
@pytest.mark.parametrize("loop", ["host", "device"])
def test_c4_literal_ten_iterations_equal_the_reference_run(lib, ctx, c4l, loop):
    """RegistrationICP from identity, criteria (0, 0, 10), every pass through the ring search (the first bounded by the
    radius, the others by the previous winner): the reference's correspondences and, to 1e-9, its transform."""
    src, tgt, r = c4l
    ctx.set_nn_mode(lib.NN_AUTO)
    ctx.set_device_loop(loop == "device")
    ctx.set_clouds_f64(src, tgt)
    got = ctx.run(np.eye(4), r, int(GL["iters"]), 0.0, 0.0)
    ctx.set_device_loop(None)
    assert ctx.nn_mode_used() == lib.NN_GRID and ctx.search_kernel_used() == "ring"
    assert got.num_correspondences == int(GL["ref_k"])
    s1, s2, k = gen_c4.checksum(ctx.correspondence_index())
    assert (s1, s2, k) == (int(GL["ref_sum"]), int(GL["ref_wsum"]), int(GL["ref_k"]))
    e = synth.rel_frobenius(got.transformation_, GL["ref_T"])
    print("C4 literal motion vs compiled reference after %d iterations: %.3e" % (int(GL["iters"]), e))
    assert e < 1e-9
    assert got.fitness_ == float(GL["ref_fitness"])
    assert abs(got.inlier_rmse_ - float(GL["ref_rmse"])) < 1e-12 * float(GL["ref_rmse"])
