// host_math.hpp -- the tiny f64 solves of the ICP loop.  Every function is
// host+device: the synchronous API solves on the host, the asynchronous
// on-device loop (icp_loop.hip) runs the very same code in a one-thread kernel.
//
// The reference does these with Eigen (umeyama / JacobiSVD / LDLT).  Eigen is
// not a dependency of this library (there is none on the target image), so the
// few fixed-size operations needed are written out here:
//   kabsch_from_stats   Eigen/src/Geometry/Umeyama.h:118-159 evaluated from the
//                       reduced moments instead of from gathered 3xK matrices
//   gn_from_stats       O3D/Core/Utility/Eigen.cpp:35-56,88-106 (6x6 solve with
//                       the |det| < 1e-6 guard) + :58-68 (Euler ZYX) or the
//                       exponential map of core/rodrigues.h:143-182
// All matrices row-major.
#pragma once

#include <math.h>
#include <string.h>

#include "so3.h"

namespace visma {

struct Mat4 {
    double m[16];
    VISMA_HD static Mat4 identity()
    {
        Mat4 r;
        for (int i = 0; i < 16; i++) r.m[i] = 0.0;
        r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0;
        return r;
    }
    VISMA_HD static Mat4 from(const double *p)
    {
        Mat4 r;
        for (int i = 0; i < 16; i++) r.m[i] = p[i];
        return r;
    }
    VISMA_HD double &operator()(int i, int j) { return m[i * 4 + j]; }
    VISMA_HD double operator()(int i, int j) const { return m[i * 4 + j]; }
};

VISMA_HD Mat4 operator*(const Mat4 &a, const Mat4 &b)
{
    Mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += a(i, k) * b(k, j);
            r(i, j) = s;
        }
    return r;
}

// x' = x - c  ==>  [R | t]  <->  [R | R c + t - c]
VISMA_HD Mat4 to_centred(const Mat4 &T, const double c[3])
{
    Mat4 r = T;
    for (int i = 0; i < 3; i++)
        r(i, 3) = T(i, 0) * c[0] + T(i, 1) * c[1] + T(i, 2) * c[2] + T(i, 3) - c[i];
    return r;
}

VISMA_HD Mat4 from_centred(const Mat4 &Tc, const double c[3])
{
    Mat4 r = Tc;
    for (int i = 0; i < 3; i++)
        r(i, 3) = Tc(i, 3) - (Tc(i, 0) * c[0] + Tc(i, 1) * c[1] + Tc(i, 2) * c[2]) + c[i];
    return r;
}

VISMA_HD double det3(const double A[9])
{
    return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) +
           A[2] * (A[3] * A[7] - A[4] * A[6]);
}

// Singular value decomposition of a 3x3 by one-sided Jacobi rotations on the
// columns (A V = U S); singular values sorted descending, U completed to a
// full orthonormal basis when A is rank deficient.
VISMA_HD void cross3(const double *a, const double *b, double *c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

struct Svd3 {
    double U[9], s[3], V[9];
};

// Written so that every array index is a compile-time constant after unrolling: on the GPU
// (one-thread epilogue of the batched device loop) the first version indexed G / W / the sort
// permutation dynamically, which put them in scratch memory -- 720 B of private segment and
// most of the 8 us the solve took.  Same operations in the same order as before.
VISMA_HD Svd3 svd3(const double A[9])
{
    double G[3][3], W[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) G[i][j] = A[i * 3 + j];
    // Convergence: every column pair orthogonal to 2 ulp (|g| <= 4.4e-16 sqrt(ab)).
    // A tighter bound only burns sweeps (each rotation is a chain of dependent f64
    // div/sqrt) without changing R.
    bool any = false;
    auto rotate = [&](const int p, const int q) {
        double a = 0, b = 0, g = 0;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            a += G[r][p] * G[r][p];
            b += G[r][q] * G[r][q];
            g += G[r][p] * G[r][q];
        }
        if (g == 0.0 || g * g <= 1.9e-31 * (a * b)) return;
        any = true;
        const double zeta = (b - a) / (2.0 * g);
        const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const double gp = G[r][p], gq = G[r][q];
            G[r][p] = c * gp - s * gq;
            G[r][q] = s * gp + c * gq;
            const double wp = W[r][p], wq = W[r][q];
            W[r][p] = c * wp - s * wq;
            W[r][q] = s * wp + c * wq;
        }
    };
    for (int sweep = 0; sweep < 30; sweep++) {
        any = false;
        rotate(0, 1);
        rotate(0, 2);
        rotate(1, 2);
        if (!any) break;
    }
    double n[3];
#pragma unroll
    for (int j = 0; j < 3; j++) n[j] = sqrt(G[0][j] * G[0][j] + G[1][j] * G[1][j] + G[2][j] * G[2][j]);
    // columns by descending norm: the same three comparisons as a sort of an index permutation
    // (a, b) = (0,1), (0,2), (1,2), carried out on the columns themselves
    auto order = [&](const int a, const int b) {
        if (n[b] > n[a]) {
            const double t = n[a]; n[a] = n[b]; n[b] = t;
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const double tg = G[r][a]; G[r][a] = G[r][b]; G[r][b] = tg;
                const double tw = W[r][a]; W[r][a] = W[r][b]; W[r][b] = tw;
            }
        }
    };
    order(0, 1);
    order(0, 2);
    order(1, 2);
    Svd3 out;
    double u[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    int rank = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        out.s[j] = n[j];
#pragma unroll
        for (int r = 0; r < 3; r++) out.V[r * 3 + j] = W[r][j];
        if (n[j] > 1e-300 && n[j] > 1e-15 * n[0]) {
#pragma unroll
            for (int r = 0; r < 3; r++) u[j][r] = G[r][j] / n[j];
            rank = j + 1;
        }
    }
    if (rank == 0) {
        // zero matrix: no rotation is needed; U = V = I, which is also what the
        // reference's JacobiSVD returns (so the update is a pure translation)
#pragma unroll
        for (int i = 0; i < 9; i++) out.U[i] = out.V[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return out;
    }
    if (rank == 1) {
        // the axis along which u0 is smallest (first one on ties), crossed with u0
        int m = 0;
        double mv = fabs(u[0][0]);
        if (fabs(u[0][1]) < mv) { m = 1; mv = fabs(u[0][1]); }
        if (fabs(u[0][2]) < mv) { m = 2; }
        const double e[3] = {m == 0 ? 1.0 : 0.0, m == 1 ? 1.0 : 0.0, m == 2 ? 1.0 : 0.0};
        cross3(u[0], e, u[1]);
        const double l = sqrt(u[1][0] * u[1][0] + u[1][1] * u[1][1] + u[1][2] * u[1][2]);
#pragma unroll
        for (int r = 0; r < 3; r++) u[1][r] /= l;
        rank = 2;
    }
    if (rank == 2) cross3(u[0], u[1], u[2]);
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int r = 0; r < 3; r++) out.U[r * 3 + j] = u[j][r];
    return out;
}

// projectSO3 (core/rodrigues.h:229-237) and SO3Type::fitToSO3 (core/se3.h:58-61): U V^T of the SVD,
// the orthogonal factor nearest to A (no determinant fix, like the reference: a reflection stays one).
VISMA_HD void project_so3(const double A[9], double R[9])
{
    const Svd3 d = svd3(A);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            R[i * 3 + j] = d.U[i * 3] * d.V[j * 3] + d.U[i * 3 + 1] * d.V[j * 3 + 1] + d.U[i * 3 + 2] * d.V[j * 3 + 2];
}

// The rotation nearest to A -- the orthogonal polar factor Q of A = Q H -- by Newton's iteration
// X <- (X + X^-T) / 2 from X0 = A / |A|_F (Higham, "Computing the polar decomposition -- with applications", 1986):
// the singular vectors of X never change, every singular value s goes to (s + 1/s) / 2 -> 1 quadratically, and
// det X keeps its sign.  For det A > 0 that Q is exactly Umeyama's U S V^T (Umeyama.h:118-159: S = I when
// det U det V > 0), with tr(Q^T A) the sum of A's singular values -- WITHOUT the one-sided Jacobi sweeps of svd3,
// whose ~18 rotations are a chain of dependent f64 divisions and square roots: 9 us in one GPU thread (the solve
// launch between every two search launches of a batch: 18 % of config 3's GPU time until round 5) against ~1 us
// here (one division per step, 8-13 steps).  Only +, -, x, / in a fixed order: host and device agree bit for bit.
// Returns false -- the caller then takes the SVD path -- for det A <= 0 (a reflection is nearest: Umeyama flips the
// smallest singular direction, which only the SVD names), for a matrix too close to singular for the inverse to be
// trusted, and for non-finite input.  "Too close": the first step inverts X0, and a Newton step from an X0 with singular
// values s1 >= s2 >= s3 loses ~2e-17 / (s3 / s1) of the singular VECTORS for good (measured: |Q - U V^T| = 2e-12 at
// s3 / s1 = 1e-5, 6e-11 at 3e-7 -- walls and floors seen at an angle land there).  The SVD path is good to
// 1e-16 at any conditioning, so the Newton path is taken only from det X0 >= 1e-3 (|X0|_F = 1: s3 / s1 >~ 2e-3, error
// <~ 1e-14) and everything thinner -- nearly planar, planar, collinear sets, a handful of pairs -- goes to svd3.
VISMA_HD bool polar_rotation3(const double A[9], double Q[9])
{
    double f2 = 0.0;
#pragma unroll
    for (int i = 0; i < 9; i++) f2 += A[i] * A[i];
    if (!(f2 > 0.0) || !(f2 < 1e300)) return false;
    const double inv_f = 1.0 / sqrt(f2);
    double X[9];
#pragma unroll
    for (int i = 0; i < 9; i++) X[i] = A[i] * inv_f;
    for (int it = 0; it < 64; it++) {
        // cofactors: C = det(X) X^-T
        double C[9];
        C[0] = X[4] * X[8] - X[5] * X[7];
        C[1] = X[5] * X[6] - X[3] * X[8];
        C[2] = X[3] * X[7] - X[4] * X[6];
        C[3] = X[2] * X[7] - X[1] * X[8];
        C[4] = X[0] * X[8] - X[2] * X[6];
        C[5] = X[1] * X[6] - X[0] * X[7];
        C[6] = X[1] * X[5] - X[2] * X[4];
        C[7] = X[2] * X[3] - X[0] * X[5];
        C[8] = X[0] * X[4] - X[1] * X[3];
        const double det = X[0] * C[0] + X[1] * C[1] + X[2] * C[2];
        // (|X0|_F = 1: det X0 = s1 s2 s3 <= 3^-3/2; later iterates have every s >= 1)
        if (!(det > (it == 0 ? 1e-3 : 0.5))) return false;
        const double inv = 1.0 / det;
        double d2 = 0.0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const double xn = 0.5 * (X[i] + C[i] * inv);
            const double d = xn - X[i];
            d2 += d * d;
            X[i] = xn;
        }
        // quadratic convergence: a step below 1e-8 leaves an error of ~1e-16 -- one more step then changes nothing
        // above rounding
        if (d2 < 1e-16) {
#pragma unroll
            for (int i = 0; i < 9; i++) Q[i] = X[i];
            return true;
        }
    }
    return false;
}

struct NormalEq {
    double K, r2;
    double JTJ[36];
    double JTr[6];
    double M[9];  // sum q p^T
};

VISMA_HD NormalEq unpack_stats(const double *st)
{
    NormalEq e;
    e.K = st[0];
    e.r2 = st[1];
    int o = 2;
    for (int a = 0; a < 6; a++)
        for (int b = a; b < 6; b++) {
            e.JTJ[a * 6 + b] = st[o];
            e.JTJ[b * 6 + a] = st[o];
            ++o;
        }
    for (int a = 0; a < 6; a++) e.JTr[a] = st[o++];
    for (int a = 0; a < 9; a++) e.M[a] = st[o++];
    return e;
}

// Closed-form least-squares rigid (optionally similarity) update for the
// fixed correspondence set, from the moments.
VISMA_HD Mat4 kabsch_from_stats(const double *st, bool with_scaling)
{
    const NormalEq e = unpack_stats(st);
    if (!(e.K > 0.0)) return Mat4::identity();
    // the cross block of J^T J is hat(sum p); J^T r's tail is sum p - sum q
    const double P[3] = {e.JTJ[2 * 6 + 4], e.JTJ[0 * 6 + 5], e.JTJ[1 * 6 + 3]};
    const double Q[3] = {P[0] - e.JTr[3], P[1] - e.JTr[4], P[2] - e.JTr[5]};
    const double inv = 1.0 / e.K;
    double pm[3], qm[3], sigma[9];
    for (int a = 0; a < 3; a++) { pm[a] = P[a] * inv; qm[a] = Q[a] * inv; }
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) sigma[a * 3 + b] = e.M[a * 3 + b] * inv - qm[a] * pm[b];
    // R = U S V^T (Umeyama.h:139-152) and sum_i s_i S_i (:155-158): through the polar factor when det sigma > 0 and
    // sigma is well conditioned (every registration of a real surface), else through the SVD
    double R[9], strace;
    if (polar_rotation3(sigma, R)) {
        strace = 0.0;                                        // tr(R^T sigma) = s1 + s2 + s3
        for (int i = 0; i < 9; i++) strace += R[i] * sigma[i];
    } else {
        const Svd3 d = svd3(sigma);
        double S[3] = {1.0, 1.0, 1.0};
        if (det3(d.U) * det3(d.V) < 0.0) S[2] = -1.0;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                R[i * 3 + j] = d.U[i * 3] * S[0] * d.V[j * 3] + d.U[i * 3 + 1] * S[1] * d.V[j * 3 + 1] +
                               d.U[i * 3 + 2] * S[2] * d.V[j * 3 + 2];
        strace = d.s[0] * S[0] + d.s[1] * S[1] + d.s[2] * S[2];
    }
    double c = 1.0;
    if (with_scaling) {
        // tr(sum(|p|^2 I - p p^T)) = 2 sum |p|^2
        const double sum_p2 = 0.5 * (e.JTJ[0] + e.JTJ[7] + e.JTJ[14]);
        const double var = sum_p2 * inv - (pm[0] * pm[0] + pm[1] * pm[1] + pm[2] * pm[2]);
        c = strace / var;
    }
    Mat4 T = Mat4::identity();
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T(i, j) = c * R[i * 3 + j];
        T(i, 3) = qm[i] - c * (R[i * 3] * pm[0] + R[i * 3 + 1] * pm[1] + R[i * 3 + 2] * pm[2]);
    }
    return T;
}

// Solve A x = b (6x6, partial pivoting); returns det(A).
VISMA_HD double solve6(const double A[36], const double b[6], double x[6])
{
    double M[6][7];
    for (int i = 0; i < 6; i++) {
        for (int j = 0; j < 6; j++) M[i][j] = A[i * 6 + j];
        M[i][6] = b[i];
    }
    double det = 1.0;
    for (int c = 0; c < 6; c++) {
        int piv = c;
        for (int r = c + 1; r < 6; r++)
            if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
        if (M[piv][c] == 0.0) {
            for (int i = 0; i < 6; i++) x[i] = 0.0;
            return 0.0;
        }
        if (piv != c) {
            for (int j = 0; j < 7; j++) { const double t = M[c][j]; M[c][j] = M[piv][j]; M[piv][j] = t; }
            det = -det;
        }
        det *= M[c][c];
        for (int r = c + 1; r < 6; r++) {
            const double f = M[r][c] / M[c][c];
            for (int j = c; j < 7; j++) M[r][j] -= f * M[c][j];
        }
    }
    for (int r = 5; r >= 0; r--) {
        double s = M[r][6];
        for (int j = r + 1; j < 6; j++) s -= M[r][j] * x[j];
        x[r] = s / M[r][r];
    }
    return det;
}

VISMA_HD Mat4 euler_zyx_to_mat4(const double x[6])
{
    const double ca = cos(x[0]), sa = sin(x[0]);
    const double cb = cos(x[1]), sb = sin(x[1]);
    const double cg = cos(x[2]), sg = sin(x[2]);
    Mat4 T = Mat4::identity();
    // Rz(g) * Ry(b) * Rx(a), written out
    T(0, 0) = cg * cb; T(0, 1) = cg * sb * sa - sg * ca; T(0, 2) = cg * sb * ca + sg * sa;
    T(1, 0) = sg * cb; T(1, 1) = sg * sb * sa + cg * ca; T(1, 2) = sg * sb * ca - cg * sa;
    T(2, 0) = -sb;     T(2, 1) = cb * sa;                T(2, 2) = cb * ca;
    T(0, 3) = x[3]; T(1, 3) = x[4]; T(2, 3) = x[5];
    return T;
}

// One Gauss-Newton step on J^T J x = -J^T r.  ok=false (Identity) when the
// determinant guard rejects the system.
VISMA_HD Mat4 gn_from_stats(const double *st, bool expmap, bool *ok)
{
    const NormalEq e = unpack_stats(st);
    *ok = false;
    if (!(e.K > 0.0)) return Mat4::identity();
    double nb[6], x[6];
    for (int i = 0; i < 6; i++) nb[i] = -e.JTr[i];
    const double det = solve6(e.JTJ, nb, x);
    if (fabs(det) < 1e-6 || isnan(det) || isinf(det)) return Mat4::identity();
    *ok = true;
    if (!expmap) return euler_zyx_to_mat4(x);
    double R[9];
    rodrigues(x, R);
    Mat4 T = Mat4::identity();
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T(i, j) = R[i * 3 + j];
        T(i, 3) = x[3 + i];
    }
    return T;
}

}  // namespace visma
