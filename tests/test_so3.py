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


# ---- SE(3): core/se3.h:79-169 (SE3Type) -- compose / act / inverse ---------------------------------------
def _se3_lib():
    import ctypes as C
    from visma_amd import _lib
    L = _lib.load()
    dp = C.POINTER(C.c_double)
    return L, lambda a: a.ctypes.data_as(dp)


def _rand_g(rng, n):
    from scipy.spatial.transform import Rotation
    g = np.zeros((n, 3, 4))
    g[:, :, :3] = Rotation.random(n, random_state=int(rng.integers(1 << 30))).as_matrix()
    g[:, :, 3] = rng.standard_normal((n, 3)) * 3
    return g.reshape(n, 12)


SE3G = np.load(os.path.join(HERE, "golden", "se3.npz"))


def test_se3_host_functions_match_the_reference_header(lib):
    """tests/golden/se3.npz: outputs of the reference's own SE3Type / SO3Type (core/se3.h:79-169, :10-76), the header
    compiled as it is into oracle/_ref (oracle/ref_se3.cpp; generator tests/golden/gen_se3.py) -- composition, action on
    a point, inverse, the 4 x 4 form; SO3Type::exp / log / (axis, angle) against visma_so3_rodrigues / _invrodrigues."""
    L, P = _se3_lib()
    g, h, v = SE3G["g"], SE3G["h"], SE3G["v"]
    for k in range(len(g)):
        gh, gv, gi = np.empty(12), np.empty(3), np.empty(12)
        a, b = np.ascontiguousarray(g[k].reshape(12)), np.ascontiguousarray(h[k].reshape(12))
        assert L.visma_se3_compose(P(a), P(b), P(gh)) == 0 and L.visma_se3_inv(P(a), P(gi)) == 0
        assert L.visma_se3_act(P(a), P(np.ascontiguousarray(v[k])), P(gv)) == 0
        scale = 1.0 + np.abs(g[k]).max() + np.abs(h[k]).max()
        assert np.max(np.abs(gh.reshape(3, 4) - SE3G["gh"][k])) <= 4e-16 * scale * scale, k
        assert np.max(np.abs(gv - SE3G["gv"][k])) <= 4e-16 * scale * (1.0 + np.abs(v[k]).max()), k
        assert np.max(np.abs(gi.reshape(3, 4) - SE3G["ginv"][k])) <= 4e-16 * scale, k
        # the 4 x 4 form (SE3Type::matrix) is [g; 0 0 0 1]
        assert np.array_equal(SE3G["g44"][k][:3], g[k]) and np.array_equal(SE3G["g44"][k][3], [0, 0, 0, 1])
    for k in range(len(SE3G["w"])):
        R, _ = rodrigues(L, SE3G["w"][k])
        assert np.max(np.abs(R - SE3G["expw"][k])) < 1e-14, k                # SO3Type::exp = rodrigues
        w, _ = invrodrigues(L, SE3G["expw"][k])
        assert np.max(np.abs(w - SE3G["logR"][k])) < 1e-12, k                # SO3Type::log = invrodrigues
        ax = SE3G["axis"][k]
        Ra, _ = rodrigues(L, ax / np.linalg.norm(ax) * SE3G["angle"][k])     # SO3Type(axis, angle), se3.h:23-24
        assert np.max(np.abs(Ra - SE3G["axis_angle"][k])) < 1e-14, k


def test_se3_oracle_restatement_matches_the_reference_header(oracle):
    """the same vectors against oracle/icp_oracle.c: vo_se3_* (what the property tests and the device tests compare with)"""
    for k in range(len(SE3G["g"])):
        g, h, v = SE3G["g"][k], SE3G["h"][k], SE3G["v"][k]
        scale = 1.0 + np.abs(g).max() + np.abs(h).max()
        R, t = oracle.se3_compose(g[:, :3], g[:, 3], h[:, :3], h[:, 3])
        assert np.max(np.abs(np.c_[R, t] - SE3G["gh"][k])) <= 4e-16 * scale * scale, k
        assert np.max(np.abs(oracle.se3_act(g[:, :3], g[:, 3], v) - SE3G["gv"][k])) <= 4e-16 * scale * (1.0 + np.abs(v).max()), k
        Ri, ti = oracle.se3_inv(g[:, :3], g[:, 3])
        assert np.max(np.abs(np.c_[Ri, ti] - SE3G["ginv"][k])) <= 4e-16 * scale, k


def test_se3_host_functions_equal_the_oracle_restatement_and_form_a_group(lib, oracle):
    """Beside the reference's vectors above: the oracle's line-by-line restatement (oracle/icp_oracle.c: vo_se3_*) on
    random elements, and the group axioms."""
    L, P = _se3_lib()
    rng = np.random.default_rng(4)
    G, H, K = _rand_g(rng, 50), _rand_g(rng, 50), _rand_g(rng, 50)
    V = rng.standard_normal((50, 3))
    I = np.hstack([np.eye(3), np.zeros((3, 1))]).ravel()
    for g, h, k, v in zip(G, H, K, V):
        gh, gv, gi, t = np.empty(12), np.empty(3), np.empty(12), np.empty(12)
        assert L.visma_se3_compose(P(g), P(h), P(gh)) == 0 and L.visma_se3_act(P(g), P(v), P(gv)) == 0
        assert L.visma_se3_inv(P(g), P(gi)) == 0
        Rg, tg, Rh, th = g.reshape(3, 4)[:, :3], g.reshape(3, 4)[:, 3], h.reshape(3, 4)[:, :3], h.reshape(3, 4)[:, 3]
        Ro, to = oracle.se3_compose(Rg, tg, Rh, th)
        assert np.allclose(gh.reshape(3, 4)[:, :3], Ro, atol=1e-15) and np.allclose(gh.reshape(3, 4)[:, 3], to, atol=1e-14)
        assert np.allclose(gv, oracle.se3_act(Rg, tg, v), atol=1e-15)
        Ri, ti = oracle.se3_inv(Rg, tg)
        assert np.allclose(gi.reshape(3, 4)[:, :3], Ri, atol=0) and np.allclose(gi.reshape(3, 4)[:, 3], ti, atol=1e-15)
        # axioms: g g^-1 = e, (g h) k = g (h k), (g h)(v) = g(h(v))
        L.visma_se3_compose(P(g), P(gi), P(t))
        assert np.allclose(t, I, atol=1e-14)
        a, b, hk = np.empty(12), np.empty(12), np.empty(12)
        L.visma_se3_compose(P(gh), P(k), P(a)); L.visma_se3_compose(P(h), P(k), P(hk)); L.visma_se3_compose(P(g), P(hk), P(b))
        assert np.allclose(a, b, atol=1e-13)
        hv, ghv, w = np.empty(3), np.empty(3), np.empty(3)
        L.visma_se3_act(P(h), P(v), P(hv)); L.visma_se3_act(P(g), P(hv), P(ghv)); L.visma_se3_act(P(gh), P(v), P(w))
        assert np.allclose(w, ghv, atol=1e-13)


@pytest.mark.gpu
def test_se3_on_the_device_is_the_host_code_bit_for_bit(lib):
    """The kernels transform every source point with se3_act: the device build of so3.h must agree with the
    host build to the last bit (-ffp-contract=off on both sides)."""
    L, P = _se3_lib()
    rng = np.random.default_rng(6)
    n = 1000
    G, H, V = _rand_g(rng, n), _rand_g(rng, n), rng.standard_normal((n, 3))
    gh, gv, gi = np.empty((n, 12)), np.empty((n, 3)), np.empty((n, 12))
    assert L.visma_icp_selftest_se3(P(G), P(H), P(V), n, P(gh), P(gv), P(gi)) == 0
    for i in range(0, n, 37):
        a, b, c = np.empty(12), np.empty(3), np.empty(12)
        L.visma_se3_compose(P(G[i]), P(H[i]), P(a)); L.visma_se3_act(P(G[i]), P(V[i]), P(b)); L.visma_se3_inv(P(G[i]), P(c))
        assert np.array_equal(a, gh[i]) and np.array_equal(b, gv[i]) and np.array_equal(c, gi[i])
