"""open3d::VoxelDownSample (O3D/Core/Geometry/DownSample.cpp:179-220): oracle vs
the reference's outputs (CPU) and the GPU implementation vs the same fixtures.

The reference emits voxels in its hash map's iteration order; both the oracle
and the GPU path emit them in ascending (ix,iy,iz) order, so rows are compared
after a lexicographic sort -- and then they must match BIT FOR BIT (same f64
voxel-index expression, sums taken in input order)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ("v005", "v02", "v1em4")


def lexsorted(p, *others):
    k = np.lexsort((p[:, 2], p[:, 1], p[:, 0]))
    return (p[k],) + tuple(o[k] for o in others)


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def test_oracle_matches_reference_outputs(oracle):
    g = np.load(os.path.join(G, "voxel.npz"))
    for c in CASES:
        p, n, col = oracle.voxel_down_sample(g["xyz"], float(g[c + "_size"]), g["normals"], g["colors"])
        p, n, col = lexsorted(p, n, col)
        assert same(p, g[c + "_p"]) and same(n, g[c + "_n"]) and same(col, g[c + "_c"]), c
    # known-answer unit test (UnitTest/Core/Geometry/PointCloud.cpp:677-783): 20 points, voxel 0.5
    p, _, _ = oracle.voxel_down_sample(g["ka_points"], 0.5)
    assert same(lexsorted(p)[0], g["ka_out_sorted"])
    assert np.abs(p - g["ka_ref_first"]).max(1).min() < 1e-6
    # degenerate arguments (DownSample.cpp:183-195)
    assert len(oracle.voxel_down_sample(g["xyz"], 0.0)[0]) == 0
    assert len(oracle.voxel_down_sample(g["xyz"], -1.0)[0]) == 0
    assert len(oracle.voxel_down_sample(g["xyz"], 1e-12)[0]) == 0      # voxel * INT_MAX < extent


def test_oracle_matches_live_reference(oracle, ref):
    rng = np.random.default_rng(4)
    pts = rng.standard_normal((5000, 3)) * np.array([3.0, 0.5, 1.0])
    nrm = rng.standard_normal((5000, 3))
    nrm[::97] = np.nan
    for v in (0.07, 0.5, 4.0):
        a = lexsorted(*oracle.voxel_down_sample(pts, v, nrm, None)[:2])
        b = lexsorted(*ref.voxel_down_sample(pts, v, nrm, None)[:2])
        assert same(a[0], b[0]) and same(a[1], b[1])


@pytest.mark.gpu
def test_gpu_matches_reference_outputs(gpu_ctx_auto, oracle):
    ctx = gpu_ctx_auto
    g = np.load(os.path.join(G, "voxel.npz"))
    for c in CASES:
        p, n, col = ctx.voxel_down_sample(g["xyz"], float(g[c + "_size"]), g["normals"], g["colors"])
        o = oracle.voxel_down_sample(g["xyz"], float(g[c + "_size"]), g["normals"], g["colors"])
        assert same(p, o[0]) and same(n, o[1]) and same(col, o[2]), c      # same order as the oracle
        p, n, col = lexsorted(p, n, col)
        assert same(p, g[c + "_p"]) and same(n, g[c + "_n"]) and same(col, g[c + "_c"]), c
    p, n, col = ctx.voxel_down_sample(g["ka_points"], 0.5)
    assert n is None and col is None and same(lexsorted(p)[0], g["ka_out_sorted"])
    for v in (0.0, -2.0, 1e-12):
        assert len(ctx.voxel_down_sample(g["xyz"], v)[0]) == 0
    assert len(ctx.voxel_down_sample(np.zeros((0, 3)), 0.1)[0]) == 0
    # a larger cloud: 1M points, points only
    from visma_amd import synth
    big = synth.surface_points(1 << 20, 77)
    a = ctx.voxel_down_sample(big, 0.01)[0]
    b = oracle.voxel_down_sample(big, 0.01)[0]
    assert same(a, b)


@pytest.mark.gpu
def test_voxel_grid_target_made_on_the_device_equals_the_two_steps(lib):
    """visma_icp_set_clouds_f64_voxel_target: the scene is down-sampled and installed as the ICP target without
    leaving the device.  Same points as visma_icp_voxel_down_sample, and the registration that follows is the
    one after the two separate calls, bit for bit (the centroid is summed on the device in the host's order)."""
    from visma_amd import _lib, synth
    for ns, n_scene, voxel, off in ((8000, 300000, 0.02, 0.0), (3000, 40000, 0.05, 2.5), (20000, 1200000, 0.01, -1.0)):
        src, scene, T_gt, _ = synth.make_pair(ns, n_scene, seed_t=n_scene, seed_s=ns, offset=[off, 0.5 * off, -off])
        scene = scene + np.random.default_rng(ns).normal(size=scene.shape) * 1e-4          # (not fp32-representable)
        a, b = _lib.Context(0), _lib.Context(0)
        down, _, _ = a.voxel_down_sample(scene, voxel)
        a.set_clouds_f64(src, down)
        ra = a.run(None, 0.06, 20, 1e-6, 1e-6)
        nt = b.set_clouds_f64_voxel_target(src, scene, voxel)
        assert nt == len(down)
        assert np.array_equal(b.get_voxel_target(nt), down)
        rb = b.run(None, 0.06, 20, 1e-6, 1e-6)
        assert ra.num_correspondences == rb.num_correspondences and ra.iterations == rb.iterations
        assert np.array_equal(ra.transformation_, rb.transformation_)
        assert np.array_equal(a.correspondence_index(), b.correspondence_index())
        a.close(); b.close()
    c = _lib.Context(0)
    with pytest.raises(_lib.IcpError):
        c.set_clouds_f64_voxel_target(src, scene, 0.0)
    c.close()
