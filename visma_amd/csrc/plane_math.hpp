// plane_math.hpp -- host math of the gravity-alignment steps of feh::AnnotationTool (src/annotation.cpp:82-91, 111-153):
// FindPlaneNormal (include/geometry.h:18-26), RotationBetweenVectors (core/utils.h:229-233), the centring transforms and
// the total pose.  The reference leaves the arithmetic to Eigen 3.3.2 (vendored, O3D/3rdparty/Eigen); what a drop-in has
// to reproduce is Eigen's RESULT INCLUDING THE SIGN of the singular vector -- the floor normal's sign decides whether the
// scene ends up upright or upside down -- so the two-sided Jacobi SVD below follows Eigen's published algorithm step by
// step (Eigen/src/SVD/JacobiSVD.h:compute, Eigen/src/Jacobi/Jacobi.h: makeJacobi, real_2x2_jacobi_svd): same sweep
// order, same thresholds, same sign fix, same sort.  Pinned against the library itself (oracle/ref_igl.cpp:
// ref_jacobi_svd3 on the vendored headers; tests/test_annotation.py).  Row-major 3x3 everywhere.
#pragma once

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>

namespace visma {
namespace plane {

struct Rot { double c, s; };   // Eigen::JacobiRotation<double>

// Jacobi.h: makeJacobi(x, y, z) for the real symmetric 2x2 [[x, y], [y, z]]
inline Rot make_jacobi(double x, double y, double z)
{
    const double deno = 2.0 * std::fabs(y);
    if (deno < DBL_MIN) return Rot{1.0, 0.0};
    const double tau = (x - z) / deno;
    const double w = std::sqrt(tau * tau + 1.0);
    const double t = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
    const double sign_t = t > 0.0 ? 1.0 : -1.0;
    const double n = 1.0 / std::sqrt(t * t + 1.0);
    return Rot{n, -sign_t * (y / std::fabs(y)) * std::fabs(t) * n};
}
// Jacobi.h: rows / columns p, q of a 3x3 under a plane rotation: x' = c x + s y, y' = -s x + c y
inline void rot_rows(double M[9], int p, int q, Rot j)
{
    for (int i = 0; i < 3; i++) {
        const double x = M[3 * p + i], y = M[3 * q + i];
        M[3 * p + i] = j.c * x + j.s * y;
        M[3 * q + i] = -j.s * x + j.c * y;
    }
}
inline void rot_cols(double M[9], int p, int q, Rot j)     // applyOnTheRight(p, q, j) = rows' rule with j.transpose()
{
    for (int i = 0; i < 3; i++) {
        const double x = M[3 * i + p], y = M[3 * i + q];
        M[3 * i + p] = j.c * x - j.s * y;
        M[3 * i + q] = j.s * x + j.c * y;
    }
}

// Eigen::JacobiSVD<Matrix3d>(A, ComputeFullU | ComputeFullV): A = U diag(S) V^T, S descending
inline void jacobi_svd3(const double A[9], double U[9], double S[3], double V[9])
{
    double W[9];
    double scale = 0.0;
    for (int i = 0; i < 9; i++) scale = std::max(scale, std::fabs(A[i]));
    if (scale == 0.0) scale = 1.0;
    for (int i = 0; i < 9; i++) { W[i] = A[i] / scale; U[i] = V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    const double considerAsZero = DBL_MIN, precision = 2.0 * DBL_EPSILON;
    double maxDiag = std::max(std::fabs(W[0]), std::max(std::fabs(W[4]), std::fabs(W[8])));
    bool finished = false;
    while (!finished) {
        finished = true;
        for (int p = 1; p < 3; p++)
            for (int q = 0; q < p; q++) {
                const double threshold = std::max(considerAsZero, precision * maxDiag);
                if (std::fabs(W[3 * p + q]) > threshold || std::fabs(W[3 * q + p]) > threshold) {
                    finished = false;
                    // real_2x2_jacobi_svd(W, p, q, &j_left, &j_right)
                    double m00 = W[3 * p + p], m01 = W[3 * p + q], m10 = W[3 * q + p], m11 = W[3 * q + q];
                    Rot rot1;
                    const double t = m00 + m11, d = m10 - m01;
                    if (std::fabs(d) < DBL_MIN) {
                        rot1 = Rot{1.0, 0.0};
                    } else {
                        const double u = t / d, tmp = std::sqrt(1.0 + u * u);
                        rot1 = Rot{u / tmp, 1.0 / tmp};
                    }
                    // m.applyOnTheLeft(0, 1, rot1)
                    {
                        const double a0 = rot1.c * m00 + rot1.s * m10, a1 = rot1.c * m01 + rot1.s * m11;
                        const double b0 = -rot1.s * m00 + rot1.c * m10, b1 = -rot1.s * m01 + rot1.c * m11;
                        m00 = a0; m01 = a1; m10 = b0; m11 = b1;
                    }
                    const Rot jr = make_jacobi(m00, m01, m11);
                    // j_left = rot1 * j_right.transpose()  (JacobiRotation::operator*: c = c1 c2 - s1 s2', s = c1 s2' + s1 c2)
                    const Rot jrt{jr.c, -jr.s};
                    const Rot jl{rot1.c * jrt.c - rot1.s * jrt.s, rot1.c * jrt.s + rot1.s * jrt.c};
                    rot_rows(W, p, q, jl);
                    rot_cols(U, p, q, Rot{jl.c, -jl.s});               // U.applyOnTheRight(p, q, j_left.transpose())
                    rot_cols(W, p, q, jr);
                    rot_cols(V, p, q, jr);
                    maxDiag = std::max(maxDiag, std::max(std::fabs(W[3 * p + p]), std::fabs(W[3 * q + q])));
                }
            }
    }
    for (int i = 0; i < 3; i++) {
        const double a = std::fabs(W[4 * i]);
        S[i] = a;
        if (a != 0.0)
            for (int r = 0; r < 3; r++) U[3 * r + i] *= W[4 * i] / a;
    }
    for (int i = 0; i < 3; i++) S[i] *= scale;
    for (int i = 0; i < 3; i++) {
        int pos = i;
        for (int k = i + 1; k < 3; k++)
            if (S[k] > S[pos]) pos = k;
        if (S[pos] == 0.0) break;
        if (pos != i) {
            std::swap(S[i], S[pos]);
            for (int r = 0; r < 3; r++) { std::swap(U[3 * r + i], U[3 * r + pos]); std::swap(V[3 * r + i], V[3 * r + pos]); }
        }
    }
}

// include/geometry.h:18-26: the unit normal of the plane through a set of points = the right singular vector of the
// smallest singular value of their covariance (pts_n^T pts_n / N, means taken off), normalised
inline void find_plane_normal(const double *xyz, int64_t n, double out[3])
{
    out[0] = out[1] = 0.0; out[2] = 1.0;
    if (n <= 0) return;
    double mean[3] = {0.0, 0.0, 0.0};
    for (int64_t i = 0; i < n; i++) for (int a = 0; a < 3; a++) mean[a] += xyz[3 * i + a];
    for (int a = 0; a < 3; a++) mean[a] /= (double)n;
    double P[9] = {0};
    for (int64_t i = 0; i < n; i++) {
        const double d[3] = {xyz[3 * i] - mean[0], xyz[3 * i + 1] - mean[1], xyz[3 * i + 2] - mean[2]};
        for (int a = 0; a < 3; a++) for (int b = a; b < 3; b++) P[3 * a + b] += d[a] * d[b];
    }
    for (int a = 0; a < 3; a++) for (int b = a; b < 3; b++) { P[3 * a + b] /= (double)n; P[3 * b + a] = P[3 * a + b]; }
    double U[9], S[3], V[9];
    jacobi_svd3(P, U, S, V);
    const double v[3] = {V[2], V[5], V[8]};
    const double len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (len > 0.0) for (int a = 0; a < 3; a++) out[a] = v[a] / len;
}

// core/utils.h:231-232: Eigen::Quaternion::FromTwoVectors(u, v).toRotationMatrix() (Quaternion.h: setFromTwoVectors).
// Opposite vectors (c < -1 + 1e-12): Eigen takes the axis from an SVD of the 2x3 stack; any unit axis orthogonal to u
// gives a valid half turn -- this one picks the most orthogonal coordinate axis (a measure-zero case: a floor whose
// fitted normal is exactly -Y).
inline void rotation_between_vectors(const double u[3], const double v[3], double R[9])
{
    const double lu = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), lv = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const double a[3] = {u[0] / lu, u[1] / lu, u[2] / lu}, b[3] = {v[0] / lv, v[1] / lv, v[2] / lv};
    double c = b[0] * a[0] + b[1] * a[1] + b[2] * a[2];
    double qx, qy, qz, qw;
    if (c < -1.0 + 1e-12) {
        c = std::max(c, -1.0);
        int k = 0;
        if (std::fabs(a[1]) < std::fabs(a[k])) k = 1;
        if (std::fabs(a[2]) < std::fabs(a[k])) k = 2;
        double e[3] = {0, 0, 0}; e[k] = 1.0;
        double ax[3] = {a[1] * e[2] - a[2] * e[1], a[2] * e[0] - a[0] * e[2], a[0] * e[1] - a[1] * e[0]};
        const double l = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
        const double w2 = (1.0 + c) * 0.5, sc = std::sqrt(1.0 - w2);
        qw = std::sqrt(w2); qx = ax[0] / l * sc; qy = ax[1] / l * sc; qz = ax[2] / l * sc;
    } else {
        const double ax[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
        const double s = std::sqrt((1.0 + c) * 2.0), invs = 1.0 / s;
        qx = ax[0] * invs; qy = ax[1] * invs; qz = ax[2] * invs; qw = s * 0.5;
    }
    // QuaternionBase::toRotationMatrix
    const double tx = 2.0 * qx, ty = 2.0 * qy, tz = 2.0 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx;
    const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.0 - (txx + tyy);
}

// src/annotation.cpp:114-119 / 128-132: the translation (-mean_x, -min_y, -mean_z) that centres a cloud over the origin
// and puts its lowest point on the floor plane y = 0
inline void centre_on_floor(const double *xyz, int64_t n, double t[3])
{
    double sx = 0.0, sz = 0.0, miny = DBL_MAX;
    for (int64_t i = 0; i < n; i++) {
        sx += xyz[3 * i];
        sz += xyz[3 * i + 2];
        if (xyz[3 * i + 1] < miny) miny = xyz[3 * i + 1];
    }
    t[0] = n > 0 ? -sx / (double)n : 0.0;
    t[1] = -miny;
    t[2] = n > 0 ? -sz / (double)n : 0.0;
}

}  // namespace plane
}  // namespace visma
