"""Link-time substitution of open3d::RegistrationICP / open3d::EvaluateRegistration
(interpose/open3d_registration_interpose.cpp).

tests/cpp/interpose_driver.cpp includes ONLY the reference's real Open3D headers and calls
`open3d::RegistrationICP(*scene_est, *scene, max_distance, T_scene_src[, TransformationEstimationPointToPlane()])` as
src/evaluation.cpp:260-271 does.  It is linked with libvisma_open3d_interpose.so AHEAD of oracle/_ref/libvisma_ref.so,
which -- compiled from Open3D's own Registration.cpp -- defines the very same symbols and plays Open3D's libCore.  The
binaries are prebuilt in the build container (tests/cpp/build_shim.py: build_real, which also compiles the interposer,
shim_driver.cpp and mesh_refine_driver.cpp against the real headers in BOTH Eigen storage orders) and travel."""
import os
import subprocess
import sys

import numpy as np
import pytest

from visma_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
B = os.path.join(HERE, "cpp", "_build")
sys.path.insert(0, os.path.join(HERE, "cpp"))
import build_shim  # noqa: E402
from test_shim import run  # noqa: E402

ICP = "_ZN6open3d15RegistrationICPERKNS_10PointCloudES2_dRKN5Eigen6MatrixIdLi4ELi4ELi%dELi4ELi4EEERKNS_24TransformationEstimationERKNS_22ICPConvergenceCriteriaE"
EVAL = "_ZN6open3d20EvaluateRegistrationERKNS_10PointCloudES2_dRKN5Eigen6MatrixIdLi4ELi4ELi%dELi4ELi4EEE"


@pytest.fixture(scope="module")
def built(lib):
    if os.path.exists(build_shim.O3D_SRC) and build_shim.eigen_dir() is not None:
        build_shim.build_real()
    need = ["libvisma_open3d_interpose.so", "libvisma_open3d_interpose_rowmajor.so", "interpose_driver"]
    if not all(os.path.exists(os.path.join(B, n)) for n in need):
        pytest.skip("interposer not prebuilt and no reference checkout here")
    return B


def defined(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if ln.split()}


def test_interposer_exports_the_mangled_names_of_open3ds_own_definitions(built):
    """Same symbols as the ones Open3D's Registration.cpp defines (read off the compiled reference), column-major;
    the row-major build carries the other storage order in Eigen::Matrix4d's mangled options."""
    col = defined(os.path.join(built, "libvisma_open3d_interpose.so"))
    row = defined(os.path.join(built, "libvisma_open3d_interpose_rowmajor.so"))
    assert ICP % 0 in col and EVAL % 0 in col
    assert ICP % 1 in row and EVAL % 1 in row
    ref = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libvisma_ref.so")
    if os.path.exists(ref):
        r = defined(ref)
        assert ICP % 0 in r and EVAL % 0 in r             # what the interposer shadows is really there
    for stem in ("shim_driver", "mesh_refine_driver", "interpose_driver"):
        for tag in ("", "_rowmajor"):
            assert os.path.exists(os.path.join(built, "%s_real_headers%s.o" % (stem, tag)))    # compiled against the real headers


def test_unchanged_caller_reaches_the_gpu_library_not_open3ds_cpu_code(built, tmp_path):
    """Without a GPU the substituted entry point fails loudly -- Open3D's own definition (also in the link) would have
    run on the CPU and succeeded: the link order did substitute."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    g = np.load(os.path.join(G, "edge_cases.npz"))
    rc, err, _ = run(os.path.join(built, "interpose_driver"), "default", tmp_path, g["src"], g["tgt"], 0.05)
    assert rc == 3 and "visma_icp_create failed" in err


@pytest.mark.gpu
def test_unchanged_caller_runs_on_the_gpu_and_matches_the_reference(built, tmp_path):
    exe = os.path.join(built, "interpose_driver")
    g = np.load(os.path.join(G, "chair_5k_20k.npz"))
    src, tgt = g["src"].astype(np.float64), g["tgt"].astype(np.float64)
    rc, err, r = run(exe, "criteria", tmp_path, src, tgt, float(g["radius"]), iters=20)
    assert rc == 0, err
    assert r["extra"] >= 1                                   # the substituted entry point ran
    row = g["trace"][20]
    assert synth.rel_frobenius(r["T"], row[:16].reshape(4, 4)) < 1e-9
    assert r["k"] == row[18] and abs(r["fitness"] - row[16]) < 1e-12
    e = np.load(os.path.join(G, "estimators.npz"))
    rc, err, r = run(exe, "default", tmp_path, src, tgt, 0.075)       # src/evaluation.cpp:267-270: default estimator + criteria
    assert rc == 0, err
    assert r["extra"] >= 1 and synth.rel_frobenius(r["T"], e["termination_T"]) < 1e-9
    # src/evaluation.cpp:261-265: point-to-plane (Open3D's own estimator object, its type recognised by the adapter)
    f = np.load(os.path.join(G, "fragments.npz"))
    fs, ft = f["src"].astype(np.float64), f["tgt"].astype(np.float64)
    rc, err, p = run(exe, "plane", tmp_path, fs, ft, float(f["radius"]), init=f["init"],
                     tn=f["tgt_normals"].astype(np.float64), sn=f["src_normals"].astype(np.float64))
    assert rc == 0, err
    assert p["extra"] >= 1 and p["k"] > 0
    # ... against Open3D's own code run on the host (the compiled reference, default criteria as in the call above)
    from oracle.oracle import Ref, EST_POINT_TO_PLANE
    if Ref.available():
        w = Ref().registration_icp(fs, ft, float(f["radius"]), init=f["init"], estimator=EST_POINT_TO_PLANE,
                                   src_normals=f["src_normals"].astype(np.float64), tgt_normals=f["tgt_normals"].astype(np.float64))
        assert p["k"] == w.k and synth.rel_frobenius(p["T"], np.asarray(w.T).reshape(4, 4)) < 1e-9
    rc, err, v = run(exe, "evaluate", tmp_path, src, tgt, float(g["radius"]), init=g["trace"][20][:16].reshape(4, 4))
    assert rc == 0, err
    assert v["k"] == g["trace"][20][18]
