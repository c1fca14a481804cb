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
// Matrices are row-major T[9] / T[12] (3x4 = [R|t]).
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

}  // namespace visma
