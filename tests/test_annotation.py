"""The orientation-constraint steps of feh::AnnotationTool around RegisterModelToScene (src/annotation.cpp:82-91, 111-153):
gravity alignment from the floor fragment (FindPlaneNormal, include/geometry.h:18-26; RotationBetweenVectors,
core/utils.h:229-233), the centring transforms T1 / T2 and the total pose Ttot = (T1 T0)^-1 T3 T2.

tests/golden/annotation.npz (gen_annotation.py) holds what the reference's own code returns -- Eigen 3.3.2's JacobiSVD /
Quaternion (the sign of the floor normal included), Open3D's VoxelDownSample / Transform / RegistrationICP -- for the
pieces and for ONE whole object.  The library's port of Eigen's 3 x 3 JacobiSVD is bit-identical to the library's."""
import os

import numpy as np
import pytest

from visma_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "annotation.npz"))


def test_jacobi_svd3_is_eigens_bit_for_bit(lib):
    for A, U, S, V in zip(G["svd_A"], G["svd_U"], G["svd_S"], G["svd_V"]):
        u, s, v = lib.jacobi_svd3(A)
        assert np.array_equal(u, U) and np.array_equal(s, S) and np.array_equal(v, V)
        if S.max() > 0:
            assert np.abs(u @ np.diag(s) @ v.T - A).max() <= 1e-13 * S.max()


def test_rotation_between_vectors_is_eigens(lib):
    for (u, v), R in zip(G["rot_uv"], G["rot_R"]):
        r = lib.rotation_between_vectors(u, v)
        assert np.abs(r - R).max() < 1e-15
        assert np.abs(r @ (u / np.linalg.norm(u)) - v / np.linalg.norm(v)).max() < 1e-12
    with pytest.raises(lib.IcpError):
        lib.rotation_between_vectors([0, 0, 0], [0, 1, 0])
    # opposite vectors: a half turn about some axis orthogonal to u (Eigen picks its axis from an SVD; any is valid)
    r = lib.rotation_between_vectors([0.0, -2.0, 0.0], [0.0, 1.0, 0.0])
    assert np.abs(r @ [0, -1, 0] - [0, 1, 0]).max() < 1e-12 and abs(np.linalg.det(r) - 1) < 1e-12


def test_find_plane_normal_has_eigens_sign_and_value(lib):
    for P, n in zip(G["plane_pts"], G["plane_n"]):
        got = lib.find_plane_normal(P.astype(np.float64))
        assert np.abs(got - n).max() < 1e-11                # (not -n: the sign is the reference's)
    assert np.array_equal(lib.find_plane_normal(np.zeros((0, 3))), [0, 0, 1])


def test_against_the_library_itself_where_the_compiled_reference_is_here(lib):
    from oracle.oracle import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref not built here")
    ref = Ref()
    rng = np.random.default_rng(5)
    for k in range(400):
        A = rng.standard_normal((3, 3)) * 10 ** rng.uniform(-8, 8)
        if k % 2:
            A = A @ A.T
        for a, b in zip(lib.jacobi_svd3(A), ref.jacobi_svd3(A)):
            assert np.array_equal(a, b)
    P = rng.standard_normal((5000, 3)) * [2.0, 0.01, 1.0] + [3, -2, 9]
    assert np.abs(lib.find_plane_normal(P) - ref.find_plane_normal(P)).max() < 1e-11


def test_centring_and_total_pose(lib):
    T0, T1, T2, T3 = G["T0"], G["T1"], G["T2"], G["T3"]
    assert np.abs(lib.annot_total_pose(T0, T1, T2, T3) - G["Ttot"]).max() < 1e-13
    m = G["model_pts"].astype(np.float64)
    assert np.abs(lib.centre_on_floor(m) - T2[:3, 3]).max() < 1e-13
    with pytest.raises(lib.IcpError):
        lib.centre_on_floor(np.zeros((0, 3)))


def annotate_object(lib, ctx, floor, scan_raw, model_pts, voxel, level, thr):
    """The body of AnnotationTool's loop for one object, on the library (what visma_geometry.hpp: feh::gpu::AnnotateObject
    does in C++)."""
    n = lib.find_plane_normal(floor)
    T0 = np.eye(4); T0[:3, :3] = lib.rotation_between_vectors(n, [0, 1, 0])
    scan = ctx.voxel_down_sample(scan_raw, voxel)[0]                      # :112 on the device
    scan = scan @ T0[:3, :3].T + T0[:3, 3]
    T1 = np.eye(4); T1[:3, 3] = lib.centre_on_floor(scan)
    scan = scan + T1[:3, 3]
    T2 = np.eye(4); T2[:3, 3] = lib.centre_on_floor(model_pts)
    model = model_pts + T2[:3, 3]
    ctx.set_clouds_f64(model, scan)
    best, bl, per = ctx.run_yaw_sweep(level, thr)                         # :144 RegisterModelToScene: 24 starts in flight
    return n, T0, T1, T2, best.transformation_, lib.annot_total_pose(T0, T1, T2, best.transformation_), bl, per, len(scan)


@pytest.mark.gpu
def test_one_whole_annotation_object_matches_the_reference(lib, tmp_path):
    ctx = lib.Context(0)
    floor, scan_raw, model_pts = (G[k].astype(np.float64) for k in ("floor", "scan_raw", "model_pts"))
    n, T0, T1, T2, T3, Ttot, bl, per, nvox = annotate_object(lib, ctx, floor, scan_raw, model_pts, float(G["voxel"]),
                                                             int(G["level"]), float(G["threshold"]))
    assert nvox == int(G["n_scan_voxels"])
    assert np.abs(n - G["floor_n"]).max() < 1e-11 and np.abs(T0 - G["T0"]).max() < 1e-11
    assert np.abs(T1 - G["T1"]).max() < 1e-10 and np.abs(T2 - G["T2"]).max() < 1e-12
    assert bl == int(G["best"])
    assert [p.num_correspondences for p in per] == G["sweep_k"].tolist()   # every start's K is the reference's
    assert synth.rel_frobenius(T3, G["T3"]) < 1e-9
    assert synth.rel_frobenius(Ttot, G["Ttot"]) < 1e-9
    # alignment.json as the tool writes it (src/annotation.cpp:156, 170-186) and src/evaluation.cpp:126-137 reads it
    path = tmp_path / "alignment.json"
    lib.write_alignment_json(path, [("hermanmiller_aeron_0", Ttot)])
    back = lib.read_alignment_json(path)
    assert back[0]["name"] == "hermanmiller_aeron_0" and np.abs(back[0]["T"] - Ttot[:3]).max() < 1e-15
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("binary", ["annotate_driver", "annotate_driver_rowmajor"])
def test_annotation_loop_through_the_cpp_shim(lib, tmp_path, binary):
    """feh::gpu::GravityAlignment + AnnotateObjects (include/visma_geometry.hpp) on two copies of the fixture's object:
    the scan down-sampled and the model sampled ON THE DEVICE (its own Philox stream: T3 is compared to the
    reference's loosely, Ttot by what it does -- the model lands on the raw scan), both through the native work queue,
    alignment.json written."""
    import struct
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    exe = os.path.join(HERE, "cpp", "_build", binary)
    if not os.path.exists(exe):
        pytest.skip("annotate driver not prebuilt")
    mesh = np.load(os.path.join(HERE, "golden", "mesh.npz"))              # the reference's chair (misc/hermanmiller_aeron.obj)
    V, F = mesh["V"].astype(np.float64), mesh["F"].astype(np.int32)
    floor, scan = G["floor"].astype(np.float64), G["scan_raw"].astype(np.float64)
    inp, outp, js = str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "alignment.json")
    with open(inp, "wb") as f:
        f.write(struct.pack("<qqqqddii", len(floor), len(scan), len(V), len(F), float(G["voxel"]), float(G["threshold"]), int(G["level"]), 2))
        for a in (floor, scan, np.asarray(V, np.float64)[:, :3]):
            f.write(np.ascontiguousarray(a, "<f8").tobytes())
        f.write(np.ascontiguousarray(F, "<i4").tobytes())
    p = subprocess.run([exe, inp, outp, js], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    o = np.fromfile(outp, "<f8")
    T0 = o[:16].reshape(4, 4)
    assert np.abs(T0 - G["T0"]).max() < 1e-11
    per = o[16:].reshape(2, 66)
    poses = lib.read_alignment_json(js)
    assert [q["name"] for q in poses] == ["hermanmiller_aeron_0", "hermanmiller_aeron_1"]
    ctx = lib.Context(0)
    for k in range(2):
        T1, T2, T3, Ttot = (per[k, 16 * j:16 * j + 16].reshape(4, 4) for j in range(4))
        assert per[k, 64] == int(G["n_scan_voxels"]) and per[k, 65] == 2 * per[k, 64]
        assert np.abs(T1 - G["T1"]).max() < 1e-10
        assert np.abs(lib.annot_total_pose(T0, T1, T2, T3) - Ttot).max() < 1e-12
        assert np.abs(poses[k]["T"] - Ttot[:3]).max() < 1e-15
        # the same alignment as the reference's (other model samples: same basin, not the same digits) ...
        assert np.abs(Ttot - G["Ttot"]).max() < 2e-2
        # ... and it does what a pose is for: the model, moved by Ttot, lies on the raw scan
        m = G["model_pts"].astype(np.float64)
        ctx.set_clouds_f64(m, scan)
        ctx.nn_pass(Ttot, float(G["threshold"]))
        st = ctx.reduce()
        assert st[0] > 0.8 * len(m)
    ctx.close()
