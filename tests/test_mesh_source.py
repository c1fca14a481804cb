"""visma_icp_set_clouds_meshes_f64: feh::ICPRefinement's clouds (src/evaluation.cpp:248-271) made on the device --
every model's mesh sampled (include/geometry.h:29-64), moved by model_to_scene (PointCloud.cpp:75-80), concatenated;
the scan voxel-down-sampled -- against the same steps taken one call at a time through host arrays."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def box_mesh(sx, sy, sz, off):
    """a closed box, 12 triangles, plus a tilted quad: faces of very different areas"""
    V = np.array([[x, y, z] for x in (0, sx) for y in (0, sy) for z in (0, sz)], np.float64) + np.asarray(off, np.float64)
    F = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6],
                  [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]], np.int32)
    return V, F


def transform_like_open3d(T, p):
    """PointCloud::Transform (PointCloud.cpp:75-80): transformation * (x, y, z, 1), rows 0..2, summed left to right"""
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    return np.stack([T[r, 0] * x + T[r, 1] * y + T[r, 2] * z + T[r, 3] for r in range(3)], axis=1)


def rigid(rng, angle, shift):
    a = rng.normal(size=3); a /= np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = shift
    return T


def test_abi_exports_the_mesh_source_entry_points(lib):
    L = lib.load()
    for name in ("visma_icp_set_clouds_meshes_f64", "visma_icp_get_mesh_source"):
        assert hasattr(L, name), name


@pytest.mark.gpu
@pytest.mark.parametrize("quirks", [0, 1])
@pytest.mark.parametrize("voxel", [0.0, 0.02])
def test_device_made_clouds_equal_the_separate_steps(lib, quirks, voxel):
    from visma_amd import _lib
    rng = np.random.default_rng(11 + quirks)
    meshes = []
    for k, (dims, n) in enumerate((((0.6, 0.4, 0.5), 30000), ((0.3, 0.9, 0.2), 12000), ((0.5, 0.5, 0.5), 7001))):
        V, F = box_mesh(*dims, off=[0.1 * k, -0.2 * k, 0.05])
        T = None if k == 1 else rigid(rng, 0.3 + 0.2 * k, [0.4 * k, 0.1, -0.3 * k])
        meshes.append((V, F, n, T))
    seed = 77
    a, b = _lib.Context(0), _lib.Context(0)
    # one call at a time: sample -> (host) transform -> concatenate -> [down-sample] -> upload
    parts = []
    for k, (V, F, n, T) in enumerate(meshes):
        p = a.sample_mesh(V, F, n, quirks=bool(quirks), seed=seed + k)
        parts.append(p if T is None else transform_like_open3d(T, p))
    scene_est = np.concatenate(parts)
    # the scan: the estimated scene seen again after a small motion, denser, with noise
    big = []
    for k, (V, F, n, T) in enumerate(meshes):
        p = a.sample_mesh(V, F, 6 * n, quirks=False, seed=1000 + k)
        big.append(p if T is None else transform_like_open3d(T, p))
    T_gt = rigid(rng, 0.02, [0.01, -0.008, 0.006])
    scan = transform_like_open3d(T_gt, np.concatenate(big)) + rng.normal(size=(sum(len(x) for x in big), 3)) * 2e-4
    if voxel > 0:
        down, _, _ = a.voxel_down_sample(scan, voxel)
        a.set_clouds_f64(scene_est, down)
    else:
        down = scan
        a.set_clouds_f64(scene_est, scan)
    ra = a.run(None, 0.05, 30, 1e-6, 1e-6)
    # ... and in one call on the device
    ns, nt = b.set_clouds_meshes_f64(meshes, scan, voxel_size=voxel, reference_quirks=quirks, seed=seed)
    assert ns == len(scene_est) and nt == len(down)
    assert (ns < sum(m[2] for m in meshes)) == bool(quirks)            # (the reference's mapping drops draws)
    assert np.array_equal(b.get_mesh_source(ns), scene_est)            # the same points, bit for bit
    if voxel > 0:
        assert np.array_equal(b.get_voxel_target(nt), down)
    rb = b.run(None, 0.05, 30, 1e-6, 1e-6)
    assert ra.num_correspondences == rb.num_correspondences > 0.4 * ns and ra.iterations == rb.iterations
    assert np.array_equal(ra.transformation_, rb.transformation_)
    assert np.array_equal(a.correspondence_index(), b.correspondence_index())
    # (the reference's mapping puts about half of the draws off the surface: the registration is looser there)
    assert np.abs(rb.transformation_ - T_gt).max() < (3e-2 if quirks else 5e-3)
    # a later ordinary upload takes the context back: no stale mesh source
    b.set_clouds_f64(scene_est[:1000], down)
    with pytest.raises(_lib.IcpError):
        b.get_mesh_source(ns)
    a.close(); b.close()


@pytest.mark.gpu
def test_mesh_source_argument_errors(lib):
    from visma_amd import _lib
    V, F = box_mesh(1, 1, 1, [0, 0, 0])
    scan = np.random.default_rng(0).uniform(size=(5000, 3))
    c = _lib.Context(0)
    bad = F.copy(); bad[3, 1] = 99                                      # face index out of range
    with pytest.raises(_lib.IcpError):
        c.set_clouds_meshes_f64([(V, bad, 100, None)], scan)
    with pytest.raises(_lib.IcpError):
        c.set_clouds_meshes_f64([(V, F, -5, None)], scan)
    ns, nt = c.set_clouds_meshes_f64([], scan)                          # no model: an empty source, like the reference's loop
    assert ns == 0 and nt == len(scan)
    ns, nt = c.set_clouds_meshes_f64([(V, F, 0, None), (V, F, 64, None)], scan)
    assert ns == 64
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("binary", ["mesh_refine_driver", "mesh_refine_driver_rowmajor"])
def test_shim_icp_refinement_from_meshes(lib, binary):
    """feh::gpu::ICPRefinement(scene, models, ...) == the separate shim calls (tests/cpp/mesh_refine_driver.cpp),
    with either Eigen storage order (VISMA compiles with -DEIGEN_DEFAULT_TO_ROW_MAJOR).  Prebuilt where Eigen
    headers exist (tests/cpp/build_shim.py); the binary travels with the snapshot."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "cpp"))
    import build_shim
    if build_shim.eigen_dir() is not None:
        build_shim.build()
    exe = os.path.join(ROOT, "tests", "cpp", "_build", binary)
    if not os.path.exists(exe):
        pytest.skip("mesh_refine_driver not prebuilt and no Eigen headers here")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "MESH_REFINE_OK" in p.stdout, p.stdout + p.stderr
