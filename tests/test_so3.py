"""SO(3) maps and their derivatives (include/visma_icp.h: visma_so3_*; visma_amd/csrc/so3.h), the
restatement of the reference's core/rodrigues.h:17-237 and core/se3.h:11-76.

Pinned against tests/golden/rodrigues.npz -- outputs of the REFERENCE's own header compiled into
oracle/_ref (oracle/ref_rodrigues.cpp, EIGEN_DEFAULT_TO_ROW_MAJOR like VISMA's build) -- and against
the properties the reference's own test checks (core/test/test_rodrigues.cpp:124-242: numeric
differentiation).  Host functions on the CPU; the same code on the GPU through the self-test."""
import ctypes as C
import os

import numpy as np
import pytest

from visma_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "rodrigues.npz"))
dp = C.POINTER(C.c_double)


def p(a):
    return a.ctypes.data_as(dp)


def rodrigues(L, w):
    R, D = np.empty(9), np.empty(27)
    assert L.visma_so3_rodrigues(p(np.ascontiguousarray(w, np.float64)), p(R), p(D)) == 0
    return R.reshape(3, 3), D.reshape(9, 3)


def invrodrigues(L, R):
    w, D = np.empty(3), np.empty(27)
    assert L.visma_so3_invrodrigues(p(np.ascontiguousarray(R, np.float64).reshape(9)), p(w), p(D)) == 0
    return w, D.reshape(3, 9)


def test_exports(lib):
    L = lib.load()
    for name in ("visma_so3_rodrigues", "visma_so3_invrodrigues", "visma_so3_project",
                 "visma_so3_matrix_derivatives", "visma_icp_selftest_so3_jac"):
        assert hasattr(L, name)


def test_rodrigues_and_its_jacobian_match_the_reference(lib):
    L = lib.load()
    for i in range(len(G["w"])):
        R, D = rodrigues(L, G["w"][i])
        assert np.max(np.abs(R - G["R"][i])) < 1e-14
        assert np.max(np.abs(D - G["dR_dw"][i])) < 1e-12, i


def test_invrodrigues_and_its_jacobian_match_the_reference(lib):
    L = lib.load()
    for i in range(len(G["w"])):
        w, D = invrodrigues(L, G["R"][i])
        assert np.max(np.abs(w - G["w_back"][i])) < 1e-12
        scale = max(1.0, np.max(np.abs(G["dw_dR"][i])))
        assert np.max(np.abs(D - G["dw_dR"][i])) < 1e-10 * scale, i


def test_numeric_differentiation(lib):
    """core/test/test_rodrigues.cpp:124-242: central differences of the maps themselves."""
    L = lib.load()
    rng = np.random.default_rng(0)
    for _ in range(20):
        w = rng.standard_normal(3)
        w *= rng.uniform(0.05, 3.0) / np.linalg.norm(w)          # inside the ball |w| < pi: log(exp(w)) = w
        R, D = rodrigues(L, w)
        num = np.empty((9, 3))
        for k in range(3):
            e = np.zeros(3); e[k] = 1e-6
            num[:, k] = (rodrigues(L, w + e)[0] - rodrigues(L, w - e)[0]).reshape(9) / 2e-6
        assert np.max(np.abs(num - D)) < 1e-8
        wb, Dw = invrodrigues(L, R)
        assert np.max(np.abs(wb - w)) < 1e-11
        # chain rule: dw/dR . dR/dw = I on the tangent space
        assert np.max(np.abs(Dw @ D - np.eye(3))) < 1e-8


def test_small_angle_branches(lib):
    L = lib.load()
    w = np.array([3e-9, -1e-9, 2e-9])
    R, D = rodrigues(L, w)
    H = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    assert np.array_equal(R, np.eye(3) + H)
    dh = np.empty(27)
    assert L.visma_so3_matrix_derivatives(None, None, None, None, None, p(dh), None) == 0
    assert np.array_equal(D, dh.reshape(9, 3))
    wb, Dw = invrodrigues(L, R)
    dv = np.empty(27)
    assert L.visma_so3_matrix_derivatives(None, None, None, None, None, None, p(dv)) == 0
    assert np.array_equal(Dw, 0.5 * dv.reshape(3, 9)) and np.allclose(wb, w, atol=1e-17)


def test_matrix_product_derivatives(lib):
    L = lib.load()
    rng = np.random.default_rng(1)
    A, B = rng.standard_normal((3, 3)), rng.standard_normal((3, 3))
    dA, dB, dT = np.empty(81), np.empty(81), np.empty(81)
    assert L.visma_so3_matrix_derivatives(p(A.reshape(9).copy()), p(B.reshape(9).copy()), p(dA), p(dB), p(dT), None, None) == 0
    dA, dB, dT = dA.reshape(9, 9), dB.reshape(9, 9), dT.reshape(9, 9)
    E = rng.standard_normal((3, 3)) * 1e-6
    assert np.allclose(((A + E) @ B - A @ B).reshape(9), dA @ E.reshape(9), atol=1e-12)
    assert np.allclose((A @ (B + E) - A @ B).reshape(9), dB @ E.reshape(9), atol=1e-12)
    assert np.array_equal(dT @ A.reshape(9), A.T.reshape(9))


def test_project_so3_is_the_orthogonal_polar_factor(lib):
    L = lib.load()
    rng = np.random.default_rng(2)
    for _ in range(20):
        A = rodrigues(L, rng.standard_normal(3))[0] + rng.standard_normal((3, 3)) * 0.05
        R = np.empty(9)
        assert L.visma_so3_project(p(A.reshape(9).copy()), p(R)) == 0
        R = R.reshape(3, 3)
        U, _, Vt = np.linalg.svd(A)
        assert np.max(np.abs(R - U @ Vt)) < 1e-12          # projectSO3: U V^T, no determinant fix
        assert np.max(np.abs(R.T @ R - np.eye(3))) < 1e-13


@pytest.mark.gpu
def test_device_versions_match_the_reference_goldens(lib):
    L = lib.load()
    w = np.ascontiguousarray(G["w"], np.float64)
    n = len(w)
    R, dR, wb, dw, proj = np.empty((n, 9)), np.empty((n, 27)), np.empty((n, 3)), np.empty((n, 27)), np.empty((n, 9))
    assert L.visma_icp_selftest_so3_jac(p(w), n, p(R), p(dR), p(wb), p(dw), p(proj)) == 0
    assert np.max(np.abs(R.reshape(n, 3, 3) - G["R"])) < 1e-14
    assert np.max(np.abs(dR.reshape(n, 9, 3) - G["dR_dw"])) < 1e-12
    assert np.max(np.abs(wb - G["w_back"])) < 1e-12
    for i in range(n):
        scale = max(1.0, np.max(np.abs(G["dw_dR"][i])))
        assert np.max(np.abs(dw[i].reshape(3, 9) - G["dw_dR"][i])) < 1e-10 * scale
        P = proj[i].reshape(3, 3)
        assert np.max(np.abs(P.T @ P - np.eye(3))) < 1e-13
        # same input as the kernel built: the host function gives the same projection
        A = (G["R"][i].reshape(9) * (1.0 + 0.01 * (np.arange(9) % 3)) + 0.003 * np.arange(9))
        Rh = np.empty(9)
        assert L.visma_so3_project(p(np.ascontiguousarray(A)), p(Rh)) == 0
        assert np.max(np.abs(Rh - proj[i])) < 1e-13
