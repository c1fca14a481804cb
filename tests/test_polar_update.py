"""The closed-form rigid update through the polar factor (round 5: visma_amd/csrc/host_math.hpp: polar_rotation3 -- Newton's
iteration X <- (X + X^-T) / 2 on the cross-covariance, instead of Jacobi sweeps) against the oracle's restatement of
Eigen::umeyama (Umeyama.h:93-162: SVD, S = diag(1, 1, sign)) on correspondence sets of every kind the fast path and its
fall-back have to get right: generic clouds, large rotations, anisotropic and nearly planar sets, exactly planar and
collinear sets (rank-deficient covariance: SVD path), mirrored partners (det < 0: the reflection case of Umeyama), a
handful of pairs.  CPU only: visma_icp_solve_from_stats is host code."""
import numpy as np
import pytest

from visma_amd import synth


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def cases(rng):
    out = []
    for k in range(60):
        n = int(rng.integers(4, 4000))
        p = rng.standard_normal((n, 3)) * rng.uniform(0.05, 3.0, 3)          # anisotropic extents
        ang = rng.uniform(-3.0, 3.0)
        ax = rng.standard_normal(3)
        ax /= np.linalg.norm(ax)
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
        q = p @ R.T + rng.standard_normal(3) + rng.standard_normal((n, 3)) * 10.0 ** rng.uniform(-6, -1)
        out.append(("generic %d" % k, p, q))
    for k in range(12):
        n = 500
        p = rng.standard_normal((n, 3)) * np.array([1.0, 1.0, 10.0 ** rng.uniform(-9, -2)])   # nearly planar
        q = p @ synth.rot_y(0.3).T + rng.standard_normal((n, 3)) * 1e-4
        out.append(("thin %d" % k, p, q))
    # thin sets in GENERAL position (walls and floors seen at an angle): the covariance's singular vectors are
    # random on both sides and s3 / s1 runs from 1e-7 to 1e-3 -- where a Newton step from an ill-conditioned first iterate
    # loses 2e-17 / (s3 / s1); the polar path must hand those to the SVD (first-iterate threshold in polar_rotation3)
    for k in range(40):
        n = 600
        ratio = 10.0 ** rng.uniform(-7, -3)                                   # s3 / s1 of the covariance
        A, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        B, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        B = B * np.sign(np.linalg.det(B))
        p = (rng.standard_normal((n, 3)) * np.array([1.0, rng.uniform(0.3, 1.0), np.sqrt(ratio)])) @ A.T
        q = p @ B.T + rng.standard_normal(3)
        out.append(("thin oriented %d" % k, p, q))
    p = rng.standard_normal((300, 3)); p[:, 2] = 0.0
    out.append(("planar", p, p @ synth.rot_x(0.2).T + 0.1))
    t = rng.standard_normal((200, 1))
    out.append(("collinear", t * np.array([[1.0, 2.0, -1.0]]), t * np.array([[1.0, 2.0, -1.0]]) + 0.5))
    for k in range(10):
        p = rng.standard_normal((800, 3))
        out.append(("mirrored %d" % k, p, p * np.array([1.0, 1.0, -1.0]) + rng.standard_normal((800, 3)) * 1e-3))
    for n in (1, 2, 3):
        p = rng.standard_normal((n, 3))
        out.append(("%d pairs" % n, p, p @ synth.rot_y(0.1).T + 0.01))
    return out


@pytest.mark.parametrize("scaling", [False, True])
def test_polar_update_equals_umeyama(lib, oracle, scaling):
    rng = np.random.default_rng(2025)
    worst = 0.0
    for name, p, q in cases(rng):
        p32, q32 = p.astype(np.float32), q.astype(np.float32)
        idx = np.arange(len(p), dtype=np.int32)
        st = oracle.k_reduce_stats(p32, q32, idx, np.eye(4)[:3])
        if scaling and len(p) < 4:
            continue                                           # (variance of a few points: both paths divide by ~0)
        T = lib.solve_from_stats(st, lib.SOLVER_KABSCH, with_scaling=scaling)
        Tref = oracle.k_solve_kabsch(st, with_scaling=scaling)
        assert np.all(np.isfinite(T)), name
        # the rotation is determined to eps / (s2 + s3) of the normalised covariance: thin sets get that much room
        pc, qc = p32.astype(np.float64) - p32.mean(0), q32.astype(np.float64) - q32.mean(0)
        s = np.linalg.svd(qc.T @ pc, compute_uv=False)
        cond = s[0] / max(s[1] + s[2], 1e-300)
        tol = 1e-12 * max(cond, 1.0)
        if name in ("planar", "collinear") or name.endswith("pairs"):
            # rank-deficient: both take the SVD path (the library's completion of U is its own); compare what is
            # determined -- the images of the points
            Pm = np.c_[p32.astype(np.float64), np.ones(len(p))]
            assert np.allclose(Pm @ np.asarray(T)[:3].T, Pm @ np.asarray(Tref)[:3].T, atol=1e-9), name
            continue
        e = rel(T, Tref)
        worst = max(worst, e / max(cond, 1.0))
        assert e < tol, (name, e, cond)
        Rm = np.asarray(T)[:3, :3]
        c = np.cbrt(np.linalg.det(Rm)) if scaling else 1.0
        assert np.allclose(Rm @ Rm.T, c * c * np.eye(3), atol=1e-11 * max(c * c, 1.0)), name
        assert np.linalg.det(Rm) > 0, name                     # never a reflection (Umeyama's S)
    print("worst relative difference / conditioning: %.2e" % worst)
