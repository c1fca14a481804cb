"""The LDS-streamed search kernel (visma_amd/csrc/tile.hip; experimental, VISMA_ICP_TILE=1): a chunk
of Morton-ordered queries streams the rows of cells it needs into LDS once and searches from there.
It must return what the default (per-lane gather) exact search returns: same correspondences, same
statistics -- in every geometry (2 / 4 / 1 lanes per query), when a footprint has to be split, and
when it cannot be tiled at all (searched from global memory)."""
import os
import subprocess
import sys

import numpy as np
import pytest

# The kernel lost (2.4-3x slower than the gather kernels) and is not part of libvisma_icp.so any
# more: it lives in the side build (-DVISMA_WITH_TILE, visma_amd.build.build_experiments), which this module
# loads INSTEAD of the product library -- in a process of its own, so that the rest of the suite keeps the product.
CHILD = os.environ.get("VISMA_TILE_TEST_CHILD") == "1"


@pytest.mark.gpu
@pytest.mark.timeout(1800)
@pytest.mark.skipif(CHILD, reason="this is the child")
def test_streamed_search_in_the_side_build():
    from visma_amd import build
    lib = build.build_experiments()
    env = dict(os.environ, VISMA_TILE_TEST_CHILD="1", VISMA_ICP_LIB=lib)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=1700)
    assert p.returncode == 0 and " passed" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]


from visma_amd import _lib, synth


def ctx_with(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return _lib.Context(0)                      # the switches are read when the context is created
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


CASES = [(5000, 20000, None, None), (3000, 8000, 0.075, None), (2000, 500, 0.2, None),
         (20000, 100000, None, [3.0, -2.0, 1.0]), (65536, 1048576, None, None)]
VARIANTS = [{"VISMA_ICP_TILE_CONFIG": "0"}, {"VISMA_ICP_TILE_CONFIG": "1"}, {"VISMA_ICP_TILE_CONFIG": "3"},
            {"VISMA_ICP_TILE_CONFIG": "4"}, {"VISMA_ICP_TILE_FALLBACK": "1"}]


@pytest.mark.gpu
@pytest.mark.skipif(not CHILD, reason="runs in a child process against the side build (test_streamed_search_in_the_side_build)")
@pytest.mark.parametrize("ns,nt,radius,offset", CASES)
def test_streamed_search_equals_the_default_exact_search(lib, ns, nt, radius, offset):
    src, tgt, T_gt, r = synth.make_pair(ns, nt, seed_t=ns + 1, seed_s=nt + 2, offset=offset, motion="radius")
    r = radius or r
    ref = _lib.Context(0)
    ref.set_clouds_f64(src, tgt)
    ref.set_nn_mode(_lib.NN_GRID)
    rng = np.random.default_rng(ns)
    Ts = [np.eye(4)] + [T_gt @ synth.make_T(synth.rot_y(rng.uniform(-r, r)), rng.standard_normal(3) * r * 0.4) for _ in range(2)]
    want = []
    for T in Ts:
        ref.nn_pass(T, r)
        want.append((ref.reduce(), ref.correspondence_index()))
    assert ref.search_mode_used() == "exact"
    for env in VARIANTS:
        c = ctx_with(dict(env, VISMA_ICP_TILE="1"))
        c.set_clouds_f64(src, tgt)
        c.set_nn_mode(_lib.NN_GRID)
        for T, (st0, i0) in zip(Ts, want):
            c.nn_pass(T, r)
            st = c.reduce()
            assert np.array_equal(c.correspondence_index(), i0), env
            assert st[0] == st0[0]
            assert np.max(np.abs(st - st0)) <= 1e-11 * np.max(np.abs(st0)), env
        a = c.run(None, r, 6, 0, 0)
        b = ref.run(None, r, 6, 0, 0)
        assert a.num_correspondences == b.num_correspondences
        assert synth.rel_frobenius(a.transformation_, b.transformation_) < 1e-12
        c.close()
