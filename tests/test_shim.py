"""The C++/Eigen shim (include/constrained_ICP.h, include/Core/, visma_icp_open3d.hpp).

tests/cpp/shim_driver.cpp calls it exactly the way the reference's callers do
(`open3d::RegistrationICP(*model, *scene, threshold, init,
open3d::cicp::TransformationEstimationPointToPoint4DoF(), criteria)`,
src/annotation.cpp:51-56; src/evaluation.cpp:260-271).  It is compiled twice --
with and without -DEIGEN_DEFAULT_TO_ROW_MAJOR (VISMA's CMakeLists.txt:11-12) --
to show that the Eigen storage order of the caller cannot leak through the
C ABI.  Binaries are prebuilt where Eigen headers exist (tests/cpp/build_shim.py).
"""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from visma_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
sys.path.insert(0, os.path.join(HERE, "cpp"))
import build_shim  # noqa: E402

BINS = ["shim_driver", "shim_driver_rowmajor"]


@pytest.fixture(scope="module")
def bins(lib):
    if build_shim.eigen_dir() is not None:
        build_shim.build()
    paths = [os.path.join(HERE, "cpp", "_build", b) for b in BINS]
    if not all(os.path.exists(p) for p in paths):
        pytest.skip("shim driver not prebuilt and no Eigen headers here")
    return paths


def run(binary, mode, tmp_path, src, tgt, radius, iters=0, level=0, init=None, tn=None, sn=None):
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    init = np.eye(4) if init is None else np.asarray(init, np.float64)
    with open(inp, "wb") as f:
        f.write(struct.pack("<qqdii", len(src), len(tgt), float(radius), int(iters), int(level)))
        f.write(init.astype("<f8").tobytes())
        f.write(np.ascontiguousarray(src, "<f8").tobytes())
        f.write(np.ascontiguousarray(tgt, "<f8").tobytes())
        if tn is not None:
            f.write(np.ascontiguousarray(tn, "<f8").tobytes())
            f.write(np.ascontiguousarray(sn, "<f8").tobytes())
    p = subprocess.run([binary, mode, inp, outp], capture_output=True, text=True, timeout=300)
    if p.returncode != 0:
        return p.returncode, p.stderr, None
    o = np.fromfile(outp, "<f8")
    return 0, p.stderr, dict(T=o[:16].reshape(4, 4), fitness=o[16], rmse=o[17], k=int(o[18]), extra=o[19])


def test_estimator_entry_points_host_only(bins, tmp_path):
    """ComputeTransformation / ComputeRMSE with explicit correspondences need no GPU."""
    from oracle.oracle import Oracle
    o = Oracle()
    f = np.load(os.path.join(G, "fragments.npz"))
    src, tgt = f["src"].astype(np.float64)[:900], f["tgt"].astype(np.float64)
    corr = np.stack([np.arange(len(src)), (np.arange(len(src)) * 7919) % len(tgt)], 1).astype(np.int32)
    for scaling in (0, 1):
        want = o.umeyama(src, tgt, corr, with_scaling=bool(scaling))
        for b in bins:
            rc, err, r = run(b, "estimator", tmp_path, src, tgt, 0.1, level=scaling)
            assert rc == 0, err
            assert synth.rel_frobenius(r["T"], want) < 1e-9
            assert abs(r["rmse"] - o.compute_rmse(src, tgt, corr)) < 1e-12


def test_shim_file_readers_host_only(bins, tmp_path):
    """open3d::ReadPointCloudFromPLY and feh::gpu::LoadMesh through the shim need no GPU."""
    from visma_amd import _lib
    ply = os.path.join(G, "io", "gen_le.ply"); objf = os.path.join(G, "io", "tri.obj")
    c = _lib.read_ply(ply); V, F = _lib.read_obj(objf)
    pcd = os.path.join(G, "io", "ref_compressed.pcd"); d = _lib.read_pcd(pcd)
    os.environ["SHIM_PLY"], os.environ["SHIM_OBJ"], os.environ["SHIM_PCD"] = ply, objf, pcd
    try:
        for b in bins:
            rc, err, r = run(b, "io", tmp_path, np.zeros((1, 3)), np.zeros((1, 3)), 0.1)
            assert rc == 0, err
            t = r["T"]
            assert (t[0, 0], t[0, 1], t[0, 2]) == (len(c["xyz"]), len(c["normals"]), len(c["colors"]))
            assert t[0, 3] == c["xyz"][-1, 2] and t[2, 0] == c["colors"][0, 1]
            assert (t[1, 0], t[1, 1]) == (len(V), len(F)) and t[1, 2] == V[-1, 1] and t[1, 3] == F[-1, 2]
            assert "Read PLY failed" in err and "Read PCD failed" in err
            assert (t[2, 1], t[2, 2]) == (len(d["xyz"]), len(d["normals"])) and t[2, 3] == d["xyz"][-1, 0]
    finally:
        del os.environ["SHIM_PLY"], os.environ["SHIM_OBJ"], os.environ["SHIM_PCD"]


def test_shim_without_gpu_fails_loudly(bins, tmp_path):
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    g = np.load(os.path.join(G, "edge_cases.npz"))
    rc, err, _ = run(bins[0], "icp4dof", tmp_path, g["src"], g["tgt"], 0.05, iters=3)
    assert rc == 3 and "visma_icp_create failed" in err       # no silent CPU path


@pytest.mark.gpu
def test_shim_registration_icp_matches_reference(bins, tmp_path):
    g = np.load(os.path.join(G, "chair_5k_20k.npz"))
    src, tgt = g["src"].astype(np.float64), g["tgt"].astype(np.float64)
    for b in bins:   # column-major and row-major Eigen callers
        rc, err, r = run(b, "icp4dof", tmp_path, src, tgt, float(g["radius"]), iters=20)
        assert rc == 0, err
        row = g["trace"][20]
        assert synth.rel_frobenius(r["T"], row[:16].reshape(4, 4)) < 1e-5
        assert r["k"] == row[18] and abs(r["fitness"] - row[16]) < 1e-12
        # a user-defined estimator plugin: GPU NN passes + host virtual solve
        rc, err, p = run(b, "plugin", tmp_path, src, tgt, float(g["radius"]), iters=6)
        assert rc == 0, err
        assert p["extra"] == 6
        assert synth.rel_frobenius(p["T"], g["trace"][6][:16].reshape(4, 4)) < 1e-5
        assert p["k"] == g["trace"][6][18]
        # a class derived from a stock estimator keeps ITS ComputeTransformation (no silent fast path)
        rc, err, d = run(b, "derived", tmp_path, src, tgt, float(g["radius"]), iters=5)
        assert rc == 0, err
        assert d["extra"] == 5
        assert synth.rel_frobenius(d["T"], g["trace"][5][:16].reshape(4, 4)) < 1e-5
    e = np.load(os.path.join(G, "estimators.npz"))
    rc, err, r = run(bins[0], "default", tmp_path, src, tgt, 0.075)   # default estimator + criteria
    assert rc == 0, err
    assert synth.rel_frobenius(r["T"], e["termination_T"]) < 1e-5


@pytest.mark.gpu
def test_shim_point_to_plane_sweep_evaluate(bins, tmp_path):
    f = np.load(os.path.join(G, "fragments.npz"))
    src, tgt = f["src"].astype(np.float64), f["tgt"].astype(np.float64)
    rc, err, r = run(bins[1], "plane", tmp_path, src, tgt, float(f["radius"]), iters=10, init=f["init"],
                     tn=f["tgt_normals"].astype(np.float64), sn=f["src_normals"].astype(np.float64))
    assert rc == 0, err
    assert synth.rel_frobenius(r["T"], f["trace_p2plane"][10][:16].reshape(4, 4)) < 1e-5
    y = np.load(os.path.join(G, "yaw_sweep.npz"))
    rc, err, r = run(bins[0], "sweep", tmp_path, y["model"].astype(np.float64),
                     y["scene"].astype(np.float64), float(y["radius"]), level=int(y["level"]))
    assert rc == 0, err
    assert synth.rel_frobenius(r["T"], y["T"][int(y["best"])]) < 1e-5
    assert r["k"] == y["k"][int(y["best"])]
    g = np.load(os.path.join(G, "edge_cases.npz"))
    rc, err, r = run(bins[0], "evaluate", tmp_path, g["src"].astype(np.float64),
                     g["tgt"].astype(np.float64), 0.05, init=g["evaluate_T"])
    assert rc == 0, err
    assert r["k"] == g["evaluate_frk"][2] and abs(r["rmse"] - g["evaluate_frk"][1]) < 1e-6


@pytest.mark.gpu
def test_shim_estimate_normals_then_point_to_plane(bins, tmp_path):
    """open3d::EstimateNormals + Orient* + the point-to-plane estimator on clouds WITHOUT normals."""
    from oracle.oracle import Oracle, EST_POINT_TO_PLANE
    o = Oracle()
    f = np.load(os.path.join(G, "fragments.npz"))
    model, scene = f["src"].astype(np.float64)[::2], f["tgt"].astype(np.float64)[::2]
    r_icp = float(f["radius"])
    sn = o.estimate_normals(scene, knn=30, radius=2.0 * r_icp)
    sn[(sn @ [0.0, 0.0, 1.0]) < 0] *= -1.0                               # OrientNormalsToAlignWithDirection
    mn = o.estimate_normals(model, knn=20)
    flip = ((np.array([0.0, 0.0, 10.0]) - model) * mn).sum(1) < 0          # OrientNormalsTowardsCameraLocation
    mn[flip] *= -1.0
    want = o.registration_icp(model, scene, r_icp, init=f["init"], max_iter=8, rel_fitness=0, rel_rmse=0,
                              estimator=EST_POINT_TO_PLANE, tgt_normals=sn)
    for b in bins:
        rc, err, r = run(b, "normals_plane", tmp_path, model, scene, r_icp, iters=8, level=20, init=f["init"])
        assert rc == 0, err
        assert r["k"] == want.k
        assert synth.rel_frobenius(r["T"], want.T) < 1e-7
        assert abs(r["extra"] - sn[len(sn) // 2, 2]) < 1e-9


@pytest.mark.gpu
def test_shim_icp_refinement_with_voxel_down_sample(bins, tmp_path):
    """cicp::ICPRefinement = VoxelDownSample(scene, 0.05) + RegistrationICP (src/evaluation.cpp:258-271)."""
    from oracle.oracle import Oracle
    o = Oracle()
    f = np.load(os.path.join(G, "fragments.npz"))
    model, scene = f["src"].astype(np.float64), f["tgt"].astype(np.float64)
    down = o.voxel_down_sample(scene, 0.05)[0]
    want = o.registration_icp(model, down, 0.25, init=f["init"], max_iter=30)
    rc, err, r = run(bins[0], "refine", tmp_path, model, scene, 0.25, init=f["init"])
    assert rc == 0, err
    assert int(r["extra"]) == len(down)
    assert synth.rel_frobenius(r["T"], want.T) < 1e-5 and abs(r["k"] - want.k) <= 2


@pytest.mark.gpu
def test_shim_mesh_steps(bins, tmp_path, lib):
    """feh::gpu::{MeasureSurfaceError, SamplePointCloudFromMesh, ComputeErrorMetric} (include/geometry.h)."""
    from oracle.oracle import Oracle
    from visma_amd import _lib
    o = Oracle()
    m = np.load(os.path.join(G, "mesh.npz"))
    V, F = m["V"], m["F"]
    T = synth.make_T(synth.rot_y(0.03), [0.004, -0.002, 0.003])
    n = 3000
    ctx = _lib.Context(0)
    pts = ctx.sample_mesh(V, F, n, quirks=False, seed=0)
    Vt = V @ T[:3, :3].T + T[:3, 3]
    want = o.error_metric(np.sqrt(o.point_mesh_sqdist(pts, Vt, F)[0]))
    q = ctx.sample_mesh(V, F, n, quirks=True, seed=3)
    for b in bins:                                   # both Eigen storage orders
        rc, err, r = run(b, "mesh", tmp_path, V, F.astype(np.float64), 0.0, level=n, init=T)
        assert rc == 0, err
        got = r["T"]
        # Vt is recomputed in C++ (Eigen product): values agree to rounding, not to the bit
        assert abs(got[0, 0] - want["mean"]) < 1e-12 and abs(got[0, 1] - want["std"]) < 1e-10
        assert abs(got[0, 2] - want["median"]) < 1e-12 and abs(got[1, 0] - want["max"]) < 1e-12
        assert int(r["extra"]) == len(q)
        assert abs(got[1, 1] - np.sort(np.linalg.norm(q, axis=1))[len(q) >> 1]) < 1e-14
