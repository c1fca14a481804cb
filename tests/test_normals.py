"""open3d::EstimateNormals (O3D/Core/Geometry/EstimateNormals.cpp:114-153): the oracle's restatement
against outputs of the compiled reference (tests/golden/normals.npz, generator gen_golden.py), and the
GPU implementation (visma_icp_estimate_normals) against both."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visma_amd import _lib, synth  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "normals.npz")

CASES = {
    "chair_knn30": ("chair", dict(knn=30)),
    "chair_knn5": ("chair", dict(knn=5)),
    "chair_radius": ("chair", dict(knn=None, radius=0.05)),
    "chair_hybrid": ("chair", dict(knn=30, radius=0.05)),
    "chair_hybrid_small": ("chair", dict(knn=10, radius=0.01)),
    "frag_knn30": ("frag", dict(knn=30)),
    "frag_hybrid_keep_sign": ("frag", dict(knn=30, radius=0.1, normals=+1)),
    "frag_knn_keep_sign": ("frag", dict(knn=20, normals=-1)),
    "dup_knn10": ("dup", dict(knn=10)),
    "line_knn8": ("line", dict(knn=8)),
    "line_knn8_keep": ("line", dict(knn=8, normals="y")),
    "chair_knn2": ("chair50", dict(knn=2)),
}


def _inputs(G, name):
    key, kw = CASES[name]
    kw = dict(kw)
    pts = G["chair"][:50] if key == "chair50" else G[key]
    pts = pts.astype(np.float64)
    nrm = kw.pop("normals", None)
    if isinstance(nrm, int):
        nrm = nrm * G["frag_normals"].astype(np.float64)
    elif nrm == "y":
        nrm = np.tile([0.0, 1.0, 0.0], (len(pts), 1))
    return pts, kw, nrm


def _check(got, want, name, tol):
    assert got.shape == want.shape
    # the reference's fallbacks are exact values
    fb = (want == [0.0, 0.0, 1.0]).all(1)
    assert np.array_equal(got[fb], want[fb]), name
    err = np.abs(got - want).max(1)
    if name.startswith("dup"):
        # exactly equal distances: flann orders them by tree traversal, so WHICH of two coincident points
        # makes the list can differ -- the points are the same, the moments are summed in another order
        assert np.quantile(err, 0.99) < 1e-6, name
    else:
        assert err.max() < tol, (name, err.max(), int(err.argmax()))


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_equals_the_compiled_reference(oracle, name):
    G = np.load(GOLD)
    pts, kw, nrm = _inputs(G, name)
    got = oracle.estimate_normals(pts, normals=nrm, **kw)
    _check(got, G[name], name, 1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_equals_the_compiled_reference(lib, name):
    G = np.load(GOLD)
    pts, kw, nrm = _inputs(G, name)
    ctx = _lib.Context(0)
    got = ctx.estimate_normals(pts, normals=nrm, **kw)
    _check(got, G[name], name, 1e-9)


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_gpu_equals_oracle_on_other_clouds_and_offsets(lib, oracle):
    ctx = _lib.Context(0)
    for seed, n, off in ((1, 3000, 0.0), (2, 6000, 3.0), (3, 1500, -7.5)):
        pts = synth.surface_points(n, seed) + np.random.default_rng(seed).normal(size=(n, 3)) * 1e-3 + off
        for kw in (dict(knn=30), dict(knn=12), dict(knn=None, radius=0.09), dict(knn=25, radius=0.09), dict(knn=64)):
            want = oracle.estimate_normals(pts, **kw)
            got = ctx.estimate_normals(pts, **kw)
            fb = (want == [0.0, 0.0, 1.0]).all(1)
            assert np.array_equal(got[fb], want[fb])
            # the covariance is E[xx] - E[x]E[x]: its rounding grows with the square of the offset
            tol = 1e-9 * max(1.0, off * off) * 10
            err = np.abs(got - want).max(1)
            # (Radius searches too: the moments are summed in the order of the reference's result list)
            assert err.max() < tol, (seed, kw, err.max())


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_gpu_large_cloud_properties(lib):
    """1M points: unit length, and the plane of a sampled box face is recovered."""
    n = 1 << 20
    pts = synth.surface_points(n, 9)
    ctx = _lib.Context(0)
    got = ctx.estimate_normals(pts, knn=30)
    ln = np.linalg.norm(got, axis=1)
    assert np.abs(ln - 1.0).max() < 1e-12
    # noise-free box faces are exact planes: the normal is an axis
    ax = np.abs(got).max(1) > 1.0 - 1e-6
    assert ax.mean() > 0.15
    again = ctx.estimate_normals(pts, knn=30)
    assert np.array_equal(got, again)                        # deterministic
    hyb = ctx.estimate_normals(pts, knn=30, radius=1.0)      # radius far above the 30th neighbour: the same lists
    assert np.array_equal(hyb, got)


@pytest.mark.gpu
def test_gpu_longest_neighbour_list(lib, oracle):
    """knn = 170, the longest list the LDS holds (64 threads x 170 entries)."""
    pts = synth.surface_points(2500, 31) + np.random.default_rng(31).normal(size=(2500, 3)) * 1e-3
    ctx = _lib.Context(0)
    for kw in (dict(knn=170), dict(knn=170, radius=0.5)):
        want = oracle.estimate_normals(pts, **kw)
        got = ctx.estimate_normals(pts, **kw)
        assert np.abs(got - want).max() < 1e-9, kw


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_gpu_lists_longer_than_the_lds_holds(lib, oracle):
    """knn / max_nn above 170 and dense radius searches: the result list is a heap per point in global memory,
    heap-sorted into the reference's list order before the moments are summed."""
    pts = synth.surface_points(4000, 41) + np.random.default_rng(41).normal(size=(4000, 3)) * 1e-3
    ctx = _lib.Context(0)
    for kw in (dict(knn=171), dict(knn=400), dict(knn=600, radius=0.45), dict(knn=None, radius=0.4), dict(knn=4000)):
        want = oracle.estimate_normals(pts, **kw)
        got = ctx.estimate_normals(pts, **kw)
        assert np.abs(got - want).max() < 1e-9, kw
    far = pts + np.array([120.0, -80.0, 40.0])              # 150 m from the origin: binning is relative to the cloud
    want = oracle.estimate_normals(far, knn=None, radius=0.02)
    got = ctx.estimate_normals(far, knn=None, radius=0.02)
    fb = (want == [0.0, 0.0, 1.0]).all(1)
    assert np.array_equal(got[fb], want[fb])                 # same neighbour COUNTS (< 3 -> the fall-back normal)
    assert np.abs(got - want).max() < 1e-3                   # (the covariance itself loses digits at this offset)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_gpu_radius_search_with_one_dense_corner(lib, oracle):
    """a Radius search used to give EVERY query a list as long as the longest one in the cloud -- one dense
    corner turned the whole call into global-memory heaps.  Lists are sized per run of 16,384 queries (cell order)
    now: the sparse runs keep their short LDS lists, the corner's run spills; values as before."""
    rng = np.random.default_rng(77)
    sparse = synth.surface_points(40000, 51) + rng.normal(size=(40000, 3)) * 1e-3
    corner = np.array([1.05, 0.45, -0.9]) + rng.normal(size=(900, 3)) * 0.004        # 900 points within a few mm
    pts = np.concatenate([sparse, corner])
    ctx = _lib.Context(0)
    want = oracle.estimate_normals(pts, knn=None, radius=0.03)
    got = ctx.estimate_normals(pts, knn=None, radius=0.03)
    assert np.abs(got - want).max() < 1e-9
    assert np.array_equal(got, ctx.estimate_normals(pts, knn=None, radius=0.03))     # deterministic


@pytest.mark.gpu
def test_gpu_argument_errors(lib):
    ctx = _lib.Context(0)
    pts = np.random.default_rng(0).normal(size=(100, 3))
    assert ctx.L.visma_icp_estimate_normals(ctx._h, None, 5, None, 0, 30, 0.0, None) != 0      # NULL arrays
    out = ctx.estimate_normals(pts, knn=None, radius=0.0)    # no neighbours anywhere
    assert np.array_equal(out, np.tile([0.0, 0.0, 1.0], (100, 1)))
    assert ctx.estimate_normals(np.empty((0, 3)), knn=30).shape == (0, 3)
