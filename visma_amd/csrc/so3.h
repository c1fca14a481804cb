// so3.h -- SO(3)/SE(3) math shared by the HIP kernels and the host driver.
//
// A from-scratch restatement of the operations of the reference's
// core/rodrigues.h and core/se3.h as plain fixed-size functions that compile
// for both host and gfx950 device code (no Eigen, no templates over
// expression types):
//   hat            core/rodrigues.h:8-15
//   vee            core/rodrigues.h:37-41
//   rodrigues      core/rodrigues.h:143-182   (R only; th < 1e-8 -> I + hat(w))
//   invrodrigues   core/rodrigues.h:184-226   (w only; tmp > 1-1e-10 -> vee/2)
//   SE3 compose / act / inv     core/se3.h:96-110
//   dhat / dvee                 core/rodrigues.h:17-35, 43-56   (constant 9x3 / 3x9)
//   dAt_dA, dAB_dA, dAB_dB      core/rodrigues.h:58-141         (3x3 operands)
//   rodrigues_jac               core/rodrigues.h:143-182        (R and dR/dw, 9x3)
//   invrodrigues_jac            core/rodrigues.h:184-226        (w and dw/dR, 3x9)
//   so3_exp / so3_log           core/se3.h:50-56                (SO3Type::exp / log)
//   (projectSO3 / SO3Type::fitToSO3 need the 3x3 SVD: host_math.hpp, project_so3)
// Matrices are row-major T[9] / T[12] (3x4 = [R|t]).  A derivative of (or with respect to) a 3x3
// matrix indexes it by its ROW-MAJOR vectorisation, the convention of the reference (which is
// built with EIGEN_DEFAULT_TO_ROW_MAJOR): dR_dw[(3 i + j) * 3 + k] = d R(i,j) / d w(k),
// dw_dR[k * 9 + 3 i + j] = d w(k) / d R(i,j).
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#define VISMA_HD __host__ __device__ __forceinline__
#else
#define VISMA_HD inline
#endif

namespace visma {

template <typename T>
struct Vec3 {
    T x, y, z;
};

template <typename T>
VISMA_HD void hat(const T u[3], T M[9])
{
    M[0] = T(0); M[1] = -u[2]; M[2] = u[1];
    M[3] = u[2]; M[4] = T(0);  M[5] = -u[0];
    M[6] = -u[1]; M[7] = u[0]; M[8] = T(0);
}

template <typename T>
VISMA_HD void vee(const T R[9], T v[3])
{
    v[0] = R[7] - R[5];
    v[1] = R[2] - R[6];
    v[2] = R[3] - R[1];
}

// J rows of one point-to-point correspondence: d(R(w) p + t)/d(w,t) at w = 0
// is [-hat(p) | I]; Open3D's row convention J_k = [p x e_k | e_k] is the k-th
// row of [hat(p)^T | I] = [-hat(p) | I].
template <typename T>
VISMA_HD void point_jacobian_rows(const T p[3], T J[3][6])
{
    T H[9];
    hat(p, H);
    for (int k = 0; k < 3; k++) {
        J[k][0] = -H[k * 3 + 0];
        J[k][1] = -H[k * 3 + 1];
        J[k][2] = -H[k * 3 + 2];
        J[k][3] = (k == 0) ? T(1) : T(0);
        J[k][4] = (k == 1) ? T(1) : T(0);
        J[k][5] = (k == 2) ? T(1) : T(0);
    }
}

template <typename T>
VISMA_HD void mat3_mul(const T A[9], const T B[9], T C[9])
{
    T R[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    for (int i = 0; i < 9; i++) C[i] = R[i];
}

template <typename T>
VISMA_HD void rodrigues(const T w[3], T R[9])
{
    T th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    T H[9];
    if (th < T(1e-8)) {
        hat(w, H);
        for (int i = 0; i < 9; i++) R[i] = H[i];
        R[0] += T(1); R[4] += T(1); R[8] += T(1);
        return;
    }
    T inv = T(1) / th;
    T u[3] = {w[0] * inv, w[1] * inv, w[2] * inv};
    T s = sin(th), c = cos(th);
    T H2[9];
    hat(u, H);
    mat3_mul(H, H, H2);
    for (int i = 0; i < 9; i++) R[i] = H[i] * s + H2[i] * (T(1) - c);
    R[0] += T(1); R[4] += T(1); R[8] += T(1);
}

template <typename T>
VISMA_HD void invrodrigues(const T R[9], T w[3])
{
    T tmp = T(0.5) * (R[0] + R[4] + R[8] - T(1));
    T v[3];
    vee(R, v);
    if (tmp > T(1.0 - 1e-10)) {
        for (int i = 0; i < 3; i++) w[i] = T(0.5) * v[i];
        return;
    }
    T th = acos(tmp);
    T is = T(1) / sin(th);
    for (int i = 0; i < 3; i++) w[i] = th * T(0.5) * v[i] * is;
}

// g = [R|t] row-major 3x4.
template <typename T>
VISMA_HD void se3_act(const T g[12], const T v[3], T out[3])
{
    T o0 = g[0] * v[0] + g[1] * v[1] + g[2] * v[2] + g[3];
    T o1 = g[4] * v[0] + g[5] * v[1] + g[6] * v[2] + g[7];
    T o2 = g[8] * v[0] + g[9] * v[1] + g[10] * v[2] + g[11];
    out[0] = o0; out[1] = o1; out[2] = o2;
}

template <typename T>
VISMA_HD void se3_compose(const T a[12], const T b[12], T out[12])
{
    T r[12];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            r[i * 4 + j] = a[i * 4] * b[j] + a[i * 4 + 1] * b[4 + j] + a[i * 4 + 2] * b[8 + j];
        r[i * 4 + 3] = a[i * 4] * b[3] + a[i * 4 + 1] * b[7] + a[i * 4 + 2] * b[11] + a[i * 4 + 3];
    }
    for (int i = 0; i < 12; i++) out[i] = r[i];
}

template <typename T>
VISMA_HD void se3_inv(const T g[12], T out[12])
{
    T r[12];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r[i * 4 + j] = g[j * 4 + i];
    for (int i = 0; i < 3; i++)
        r[i * 4 + 3] = -(r[i * 4] * g[3] + r[i * 4 + 1] * g[7] + r[i * 4 + 2] * g[11]);
    for (int i = 0; i < 12; i++) out[i] = r[i];
}

// ---- derivatives --------------------------------------------------------------------------
// d hat(u) / d u: 9x3, constant.
template <typename T>
VISMA_HD void dhat(T D[27])
{
    for (int i = 0; i < 27; i++) D[i] = T(0);
    // hat(u) = [0 -u2 u1; u2 0 -u0; -u1 u0 0]: entry (i,j) depends on u_k with sign eps(i,k,j)
    D[1 * 3 + 2] = T(-1); D[2 * 3 + 1] = T(1);
    D[3 * 3 + 2] = T(1);  D[5 * 3 + 0] = T(-1);
    D[6 * 3 + 1] = T(-1); D[7 * 3 + 0] = T(1);
}

// d vee(R) / d R: 3x9, constant (vee(R) = (R21 - R12, R02 - R20, R10 - R01)).
template <typename T>
VISMA_HD void dvee(T D[27])
{
    for (int i = 0; i < 27; i++) D[i] = T(0);
    D[0 * 9 + 7] = T(1); D[0 * 9 + 5] = T(-1);
    D[1 * 9 + 2] = T(1); D[1 * 9 + 6] = T(-1);
    D[2 * 9 + 3] = T(1); D[2 * 9 + 1] = T(-1);
}

// d A^T / d A for a 3x3 A: the 9x9 permutation that swaps (i,j) and (j,i).
template <typename T>
VISMA_HD void dAt_dA(T D[81])
{
    for (int i = 0; i < 81; i++) D[i] = T(0);
    for (int m = 0; m < 3; m++)
        for (int n = 0; n < 3; n++) D[(m * 3 + n) * 9 + (n * 3 + m)] = T(1);
}

// C = A B (3x3): d C(n,p) / d A(n,m) = B(m,p);  d C(n,p) / d B(m,p) = A(n,m).
template <typename T>
VISMA_HD void dAB_dA(const T B[9], T D[81])
{
    for (int i = 0; i < 81; i++) D[i] = T(0);
    for (int n = 0; n < 3; n++)
        for (int p = 0; p < 3; p++)
            for (int m = 0; m < 3; m++) D[(n * 3 + p) * 9 + (n * 3 + m)] = B[m * 3 + p];
}
template <typename T>
VISMA_HD void dAB_dB(const T A[9], T D[81])
{
    for (int i = 0; i < 81; i++) D[i] = T(0);
    for (int n = 0; n < 3; n++)
        for (int p = 0; p < 3; p++)
            for (int m = 0; m < 3; m++) D[(n * 3 + p) * 9 + (m * 3 + p)] = A[n * 3 + m];
}

// R = exp(hat(w)) and dR/dw (9x3).  With u = w / th, H = hat(u):
//   R = I + sin(th) H + (1 - cos th) H^2
//   dR/dw_k = sum_j [ sin(th) E_j + (1 - cos th)(E_j H + H E_j) ] (delta_jk - u_j u_k) / th
//             + (cos(th) H + sin(th) H^2) u_k,            E_j = hat(e_j)
// th < 1e-8: R = I + hat(w), dR/dw = dhat  (the reference's small-angle branch).
template <typename T>
VISMA_HD void rodrigues_jac(const T w[3], T R[9], T dR_dw[27])
{
    const T th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (th < T(1e-8)) {
        T H[9];
        hat(w, H);
        for (int i = 0; i < 9; i++) R[i] = H[i];
        R[0] += T(1); R[4] += T(1); R[8] += T(1);
        dhat(dR_dw);
        return;
    }
    const T inv = T(1) / th;
    const T u[3] = {w[0] * inv, w[1] * inv, w[2] * inv};
    const T s = sin(th), c = cos(th);
    T H[9], H2[9];
    hat(u, H);
    mat3_mul(H, H, H2);
    for (int i = 0; i < 9; i++) R[i] = H[i] * s + H2[i] * (T(1) - c);
    R[0] += T(1); R[4] += T(1); R[8] += T(1);
    // dR/du_j (as 3x3 matrices), then the chain rule through u(w) and th(w)
    T dRdu[3][9];
    for (int j = 0; j < 3; j++) {
        T e[3] = {T(0), T(0), T(0)}, E[9], EH[9], HE[9];
        e[j] = T(1);
        hat(e, E);
        mat3_mul(E, H, EH);
        mat3_mul(H, E, HE);
        for (int i = 0; i < 9; i++) dRdu[j][i] = s * E[i] + (T(1) - c) * (EH[i] + HE[i]);
    }
    for (int i = 0; i < 9; i++) {
        const T dth = c * H[i] + s * H2[i];
        for (int k = 0; k < 3; k++) {
            T v = T(0);
            for (int j = 0; j < 3; j++) v += dRdu[j][i] * (((j == k) ? T(1) : T(0)) - u[j] * u[k]) * inv;
            dR_dw[i * 3 + k] = v + dth * u[k];
        }
    }
}

// w = log(R) and dw/dR (3x9).  tmp = (tr R - 1) / 2, th = acos(tmp), u = vee(R) / (2 sin th), w = th u:
//   dth/dR = -1 / sqrt(1 - tmp^2) * 0.5 vec(I)^T
//   du/dR  = 0.5 ( dvee / sin th - vee(R) cos th / sin^2 th  dth/dR )
//   dw/dR  = u dth/dR + th du/dR
// tmp > 1 - 1e-10: w = vee(R) / 2, dw/dR = dvee / 2  (the reference's small-angle branch).
template <typename T>
VISMA_HD void invrodrigues_jac(const T R[9], T w[3], T dw_dR[27])
{
    const T tmp = T(0.5) * (R[0] + R[4] + R[8] - T(1));
    T v[3], DV[27];
    vee(R, v);
    dvee(DV);
    if (tmp > T(1.0 - 1e-10)) {
        for (int i = 0; i < 3; i++) w[i] = T(0.5) * v[i];
        for (int i = 0; i < 27; i++) dw_dR[i] = T(0.5) * DV[i];
        return;
    }
    const T th = acos(tmp);
    const T sn = sin(th), is = T(1) / sn, cs = cos(th);
    T u[3];
    for (int i = 0; i < 3; i++) {
        u[i] = T(0.5) * v[i] * is;
        w[i] = th * u[i];
    }
    const T dth_dtmp = T(-1) / sqrt(T(1) - tmp * tmp);
    for (int k = 0; k < 3; k++)
        for (int e = 0; e < 9; e++) {
            const T dth = (e == 0 || e == 4 || e == 8) ? T(0.5) * dth_dtmp : T(0);
            const T du = T(0.5) * (DV[k * 9 + e] * is - v[k] * cs * is * is * dth);
            dw_dR[k * 9 + e] = u[k] * dth + th * du;
        }
}

// SO3Type::exp / SO3Type::log (core/se3.h:50-56)
template <typename T>
VISMA_HD void so3_exp(const T w[3], T R[9]) { rodrigues(w, R); }
template <typename T>
VISMA_HD void so3_log(const T R[9], T w[3]) { invrodrigues(R, w); }

}  // namespace visma
