/*
 * icp_oracle.c -- CPU oracle for the VISMA orientation-constrained ICP path.
 *
 * TEST INFRASTRUCTURE ONLY (see icp_oracle.h).  Plain C99, optional OpenMP.
 * Build: see oracle/Makefile (-O2 -ffp-contract=off -mfma so that fmaf() is
 * a single correctly-rounded instruction and nothing else is contracted).
 *
 * Reference citations are relative to /root/reference;
 * O3D = thirdparty/Open3D/src, EIG = thirdparty/Open3D/3rdparty/Eigen.
 */
#include "icp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int vo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ */
/* small dense helpers (row-major)                                     */
/* ------------------------------------------------------------------ */
static void mat4_identity(double T[16])
{
    memset(T, 0, 16 * sizeof(double));
    T[0] = T[5] = T[10] = T[15] = 1.0;
}

static void mat4_mul(const double A[16], const double B[16], double C[16])
{
    double R[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 4 + j];
            R[i * 4 + j] = s;
        }
    memcpy(C, R, sizeof(R));
}

static int mat4_is_identity(const double T[16])
{
    /* Eigen isIdentity(prec=1e-12): |a_ij - delta_ij| small relative to 1.
     * (O3D Registration.cpp:163) -- exact identity is what callers pass. */
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double e = (i == j) ? 1.0 : 0.0;
            if (fabs(T[i * 4 + j] - e) > 1e-12) return 0;
        }
    return 1;
}

static void mat3_mul(const double A[9], const double B[9], double C[9])
{
    double R[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] +
                           A[i * 3 + 1] * B[1 * 3 + j] +
                           A[i * 3 + 2] * B[2 * 3 + j];
    memcpy(C, R, sizeof(R));
}

static double mat3_det(const double A[9])
{
    return A[0] * (A[4] * A[8] - A[5] * A[7]) -
           A[1] * (A[3] * A[8] - A[5] * A[6]) +
           A[2] * (A[3] * A[7] - A[4] * A[6]);
}

/* ------------------------------------------------------------------ */
/* (A) reference-semantics f64                                         */
/* ------------------------------------------------------------------ */

/* O3D/Core/Geometry/PointCloud.cpp:75-80: new = T * (x,y,z,1), keep xyz. */
void vo_transform_points(double *xyz, int64_t n, const double T[16])
{
    for (int64_t i = 0; i < n; i++) {
        double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        xyz[3 * i + 0] = T[0] * x + T[1] * y + T[2] * z + T[3] * 1.0;
        xyz[3 * i + 1] = T[4] * x + T[5] * y + T[6] * z + T[7] * 1.0;
        xyz[3 * i + 2] = T[8] * x + T[9] * y + T[10] * z + T[11] * 1.0;
    }
}

/* O3D/Core/Geometry/PointCloud.cpp:81-86: normals use w = 0. */
void vo_transform_normals(double *nxyz, int64_t n, const double T[16])
{
    for (int64_t i = 0; i < n; i++) {
        double x = nxyz[3 * i], y = nxyz[3 * i + 1], z = nxyz[3 * i + 2];
        nxyz[3 * i + 0] = T[0] * x + T[1] * y + T[2] * z;
        nxyz[3 * i + 1] = T[4] * x + T[5] * y + T[6] * z;
        nxyz[3 * i + 2] = T[8] * x + T[9] * y + T[10] * z;
    }
}

/* flann L2<double>: sum of squared differences in x,y,z order
 * (3rdparty/flann/algorithms/dist.h:159-176). */
static inline double sqdist3(const double *a, const double *b)
{
    double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return dx * dx + dy * dy + dz * dz;
}

/* KDTreeFlann.cpp:184-185: radius*radius cast to float, then widened back
 * inside flann (KNNRadiusResultSet worst distance; result_set.h:552,582
 * rejects dist >= worst). */
static inline double radius2_as_reference(double max_dist)
{
    return (double)(float)(max_dist * max_dist);
}

int64_t vo_nn_pass(const double *src, int64_t ns, const double *tgt, int64_t nt,
                   double max_dist, int32_t *idx, double *d2, double *err2)
{
    /* Registration.cpp:47-49 */
    if (max_dist <= 0.0 || nt <= 0) {
        for (int64_t i = 0; i < ns; i++) { idx[i] = -1; if (d2) d2[i] = 0.0; }
        if (err2) *err2 = 0.0;
        return 0;
    }
    const double r2 = radius2_as_reference(max_dist);
    int64_t k = 0;
    double e2 = 0.0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(+ : k)
#endif
    for (int64_t i = 0; i < ns; i++) {
        double best = r2;
        int32_t bi = -1;
        for (int64_t j = 0; j < nt; j++) {
            double d = sqdist3(src + 3 * i, tgt + 3 * j);
            if (d < best) { best = d; bi = (int32_t)j; } /* strict: first min */
        }
        idx[i] = bi;
        if (d2) d2[i] = (bi >= 0) ? best : 0.0;
        if (bi >= 0) k++;
    }
    /* deterministic, index-ordered error sum (reference order is thread
     * dependent; Registration.cpp:74-82) */
    for (int64_t i = 0; i < ns; i++)
        if (idx[i] >= 0) e2 += sqdist3(src + 3 * i, tgt + 3 * (int64_t)idx[i]);
    if (err2) *err2 = e2;
    return k;
}

/* ---- uniform grid shared by the f64 and fp32 searches --------------- */
typedef struct {
    double mn[3];
    double h, inv_h;
    int dim[3];
    int64_t ncell;
    int32_t *start; /* ncell+1 */
    int32_t *order; /* nt: target indices sorted by cell, ascending index inside */
} vo_grid;

static int grid_cell_coord(const vo_grid *g, double v, int a)
{
    double f = floor((v - g->mn[a]) * g->inv_h);
    if (f < 0) return -1;
    if (f >= g->dim[a]) return g->dim[a];
    return (int)f;
}

static int grid_build(vo_grid *g, const double *tx, const double *ty,
                      const double *tz, int64_t nt, double h)
{
    double mx[3];
    g->mn[0] = mx[0] = tx[0];
    g->mn[1] = mx[1] = ty[0];
    g->mn[2] = mx[2] = tz[0];
    for (int64_t j = 1; j < nt; j++) {
        if (tx[j] < g->mn[0]) g->mn[0] = tx[j];
        if (tx[j] > mx[0]) mx[0] = tx[j];
        if (ty[j] < g->mn[1]) g->mn[1] = ty[j];
        if (ty[j] > mx[1]) mx[1] = ty[j];
        if (tz[j] < g->mn[2]) g->mn[2] = tz[j];
        if (tz[j] > mx[2]) mx[2] = tz[j];
    }
    /* cap the cell count: enlarge h until the grid fits */
    for (;;) {
        double n = 1.0;
        for (int a = 0; a < 3; a++) {
            double d = floor((mx[a] - g->mn[a]) / h) + 1.0;
            if (d < 1.0) d = 1.0;
            n *= d;
            g->dim[a] = (d > 2e9) ? 2000000000 : (int)d;
        }
        if (n <= 64.0 * 1024 * 1024) break;
        h *= 1.26;
    }
    g->h = h;
    g->inv_h = 1.0 / h;
    g->ncell = (int64_t)g->dim[0] * g->dim[1] * g->dim[2];
    g->start = (int32_t *)calloc((size_t)g->ncell + 1, sizeof(int32_t));
    g->order = (int32_t *)malloc((size_t)nt * sizeof(int32_t));
    if (!g->start || !g->order) return -1;
    int32_t *cell = (int32_t *)malloc((size_t)nt * sizeof(int32_t));
    if (!cell) return -1;
    for (int64_t j = 0; j < nt; j++) {
        int cx = grid_cell_coord(g, tx[j], 0), cy = grid_cell_coord(g, ty[j], 1),
            cz = grid_cell_coord(g, tz[j], 2);
        if (cx >= g->dim[0]) cx = g->dim[0] - 1;
        if (cy >= g->dim[1]) cy = g->dim[1] - 1;
        if (cz >= g->dim[2]) cz = g->dim[2] - 1;
        int64_t c = ((int64_t)cz * g->dim[1] + cy) * g->dim[0] + cx;
        cell[j] = (int32_t)c;
        g->start[c + 1]++;
    }
    for (int64_t c = 0; c < g->ncell; c++) g->start[c + 1] += g->start[c];
    int32_t *fill = (int32_t *)malloc((size_t)g->ncell * sizeof(int32_t));
    if (!fill) return -1;
    memcpy(fill, g->start, (size_t)g->ncell * sizeof(int32_t));
    for (int64_t j = 0; j < nt; j++) g->order[fill[cell[j]]++] = (int32_t)j;
    free(fill);
    free(cell);
    return 0;
}

static void grid_free(vo_grid *g)
{
    free(g->start);
    free(g->order);
}

int64_t vo_nn_pass_grid(const double *src, int64_t ns, const double *tgt,
                        int64_t nt, double max_dist, int32_t *idx, double *d2,
                        double *err2)
{
    if (max_dist <= 0.0 || nt <= 0) {
        for (int64_t i = 0; i < ns; i++) { idx[i] = -1; if (d2) d2[i] = 0.0; }
        if (err2) *err2 = 0.0;
        return 0;
    }
    const double r2 = radius2_as_reference(max_dist);
    double *tx = (double *)malloc((size_t)nt * 3 * sizeof(double));
    double *ty = tx + nt, *tz = ty + nt;
    for (int64_t j = 0; j < nt; j++) {
        tx[j] = tgt[3 * j]; ty[j] = tgt[3 * j + 1]; tz[j] = tgt[3 * j + 2];
    }
    vo_grid g;
    /* cell edge >= sqrt(r2) with a safety margin so a point at distance < r
     * is always in the 27-neighbourhood */
    grid_build(&g, tx, ty, tz, nt, sqrt(r2) * (1.0 + 1e-9) + 1e-300);
    free(tx);
    int64_t k = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : k)
#endif
    for (int64_t i = 0; i < ns; i++) {
        const double *p = src + 3 * i;
        int c[3];
        for (int a = 0; a < 3; a++) c[a] = grid_cell_coord(&g, p[a], a);
        double best = r2;
        int32_t bi = -1;
        for (int dz = -1; dz <= 1; dz++) {
            int z = c[2] + dz;
            if (z < 0 || z >= g.dim[2]) continue;
            for (int dy = -1; dy <= 1; dy++) {
                int y = c[1] + dy;
                if (y < 0 || y >= g.dim[1]) continue;
                for (int dx = -1; dx <= 1; dx++) {
                    int x = c[0] + dx;
                    if (x < 0 || x >= g.dim[0]) continue;
                    int64_t cc = ((int64_t)z * g.dim[1] + y) * g.dim[0] + x;
                    for (int32_t s = g.start[cc]; s < g.start[cc + 1]; s++) {
                        int32_t j = g.order[s];
                        double d = sqdist3(p, tgt + 3 * (int64_t)j);
                        if (d < best || (d == best && bi >= 0 && j < bi)) {
                            best = d; bi = j;
                        }
                    }
                }
            }
        }
        idx[i] = bi;
        if (d2) d2[i] = (bi >= 0) ? best : 0.0;
        if (bi >= 0) k++;
    }
    grid_free(&g);
    double e2 = 0.0;
    for (int64_t i = 0; i < ns; i++)
        if (idx[i] >= 0) e2 += sqdist3(src + 3 * i, tgt + 3 * (int64_t)idx[i]);
    if (err2) *err2 = e2;
    return k;
}

/* O3D/Core/Geometry/PointCloud.cpp:122-142 */
void vo_nn_distance(const double *src, int64_t ns, const double *tgt,
                    int64_t nt, double *dist)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < ns; i++) {
        double best = INFINITY;
        for (int64_t j = 0; j < nt; j++) {
            double d = sqdist3(src + 3 * i, tgt + 3 * j);
            if (d < best) best = d;
        }
        dist[i] = (nt > 0) ? sqrt(best) : 0.0;
    }
}

/* src/constrained_ICP.cpp:13-23 */
double vo_compute_rmse(const double *src, const double *tgt,
                       const int32_t *corr, int64_t k)
{
    if (k == 0) return 0.0;
    double err = 0.0;
    for (int64_t i = 0; i < k; i++)
        err += sqdist3(src + 3 * (int64_t)corr[2 * i],
                       tgt + 3 * (int64_t)corr[2 * i + 1]);
    return sqrt(err / (double)k);
}

/* ---- 3x3 SVD: one-sided (Hestenes) Jacobi ---------------------------- */
static void cross3(const double a[3], const double b[3], double c[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

void vo_svd3(const double A[9], double U[9], double s[3], double V[9])
{
    double G[9], W[9];
    memcpy(G, A, sizeof(G));
    memset(W, 0, sizeof(W));
    W[0] = W[4] = W[8] = 1.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        int rotated = 0;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double al = 0, be = 0, ga = 0;
                for (int r = 0; r < 3; r++) {
                    al += G[r * 3 + p] * G[r * 3 + p];
                    be += G[r * 3 + q] * G[r * 3 + q];
                    ga += G[r * 3 + p] * G[r * 3 + q];
                }
                if (ga == 0.0 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
                rotated = 1;
                double zeta = (be - al) / (2.0 * ga);
                double t = ((zeta >= 0) ? 1.0 : -1.0) /
                           (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int r = 0; r < 3; r++) {
                    double gp = G[r * 3 + p], gq = G[r * 3 + q];
                    G[r * 3 + p] = c * gp - sn * gq;
                    G[r * 3 + q] = sn * gp + c * gq;
                    double wp = W[r * 3 + p], wq = W[r * 3 + q];
                    W[r * 3 + p] = c * wp - sn * wq;
                    W[r * 3 + q] = sn * wp + c * wq;
                }
            }
        if (!rotated) break;
    }
    double nrm[3];
    int ord[3] = {0, 1, 2};
    for (int j = 0; j < 3; j++)
        nrm[j] = sqrt(G[0 * 3 + j] * G[0 * 3 + j] + G[1 * 3 + j] * G[1 * 3 + j] +
                      G[2 * 3 + j] * G[2 * 3 + j]);
    for (int a = 0; a < 2; a++) /* sort descending */
        for (int b = a + 1; b < 3; b++)
            if (nrm[ord[b]] > nrm[ord[a]]) { int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    double u[3][3];
    double smax = nrm[ord[0]];
    int rank = 0;
    for (int j = 0; j < 3; j++) {
        int o = ord[j];
        s[j] = nrm[o];
        for (int r = 0; r < 3; r++) V[r * 3 + j] = W[r * 3 + o];
        if (nrm[o] > 1e-300 && nrm[o] > 1e-15 * smax) {
            for (int r = 0; r < 3; r++) u[j][r] = G[r * 3 + o] / nrm[o];
            rank = j + 1;
        }
    }
    /* complete U to an orthonormal basis when rank deficient */
    if (rank == 0) {
        /* zero matrix: Eigen's JacobiSVD performs no rotation and returns
         * U = V = I (so Umeyama yields R = I); do the same */
        memset(U, 0, 9 * sizeof(double));
        memset(V, 0, 9 * sizeof(double));
        U[0] = U[4] = U[8] = V[0] = V[4] = V[8] = 1.0;
        return;
    }
    if (rank == 1) {
        double e[3] = {0, 0, 0};
        int m = 0;
        for (int r = 1; r < 3; r++) if (fabs(u[0][r]) < fabs(u[0][m])) m = r;
        e[m] = 1.0;
        cross3(u[0], e, u[1]);
        double n = sqrt(u[1][0] * u[1][0] + u[1][1] * u[1][1] + u[1][2] * u[1][2]);
        for (int r = 0; r < 3; r++) u[1][r] /= n;
        rank = 2;
    }
    if (rank == 2) cross3(u[0], u[1], u[2]);
    for (int j = 0; j < 3; j++)
        for (int r = 0; r < 3; r++) U[r * 3 + j] = u[j][r];
}

/* shared tail of Umeyama.h:127-159 given means, sigma, src_var */
static void umeyama_from_moments(const double src_mean[3],
                                 const double dst_mean[3],
                                 const double sigma[9], double src_var,
                                 int with_scaling, double T[16])
{
    double U[9], sv[3], V[9];
    vo_svd3(sigma, U, sv, V);
    double S[3] = {1.0, 1.0, 1.0};
    if (mat3_det(U) * mat3_det(V) < 0) S[2] = -1.0; /* Umeyama.h:135-136 */
    double R[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double a = 0.0;
            for (int k = 0; k < 3; k++) a += U[i * 3 + k] * S[k] * V[j * 3 + k];
            R[i * 3 + j] = a;
        }
    mat4_identity(T);
    double c = 1.0;
    if (with_scaling) /* Umeyama.h:144 */
        c = 1.0 / src_var * (sv[0] * S[0] + sv[1] * S[1] + sv[2] * S[2]);
    for (int i = 0; i < 3; i++) {
        double rs = R[i * 3] * src_mean[0] + R[i * 3 + 1] * src_mean[1] +
                    R[i * 3 + 2] * src_mean[2];
        T[i * 4 + 3] = dst_mean[i] - c * rs; /* Umeyama.h:147-148,154-155 */
        for (int j = 0; j < 3; j++) T[i * 4 + j] = c * R[i * 3 + j];
    }
}

/* src/constrained_ICP.cpp:25-37 -> EIG/Eigen/src/Geometry/Umeyama.h:93-162 */
void vo_umeyama(const double *src, const double *tgt, const int32_t *corr,
                int64_t k, int with_scaling, double T[16])
{
    if (k == 0) { mat4_identity(T); return; } /* constrained_ICP.cpp:29 */
    const double one_over_n = 1.0 / (double)k;
    double sm[3] = {0, 0, 0}, dm[3] = {0, 0, 0};
    for (int64_t i = 0; i < k; i++) {
        const double *p = src + 3 * (int64_t)corr[2 * i];
        const double *q = tgt + 3 * (int64_t)corr[2 * i + 1];
        for (int a = 0; a < 3; a++) { sm[a] += p[a]; dm[a] += q[a]; }
    }
    for (int a = 0; a < 3; a++) { sm[a] *= one_over_n; dm[a] *= one_over_n; }
    double var = 0.0, sig[9] = {0};
    for (int64_t i = 0; i < k; i++) {
        const double *p = src + 3 * (int64_t)corr[2 * i];
        const double *q = tgt + 3 * (int64_t)corr[2 * i + 1];
        double pc[3], qc[3];
        for (int a = 0; a < 3; a++) { pc[a] = p[a] - sm[a]; qc[a] = q[a] - dm[a]; }
        var += pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) sig[a * 3 + b] += qc[a] * pc[b];
    }
    var *= one_over_n;
    for (int a = 0; a < 9; a++) sig[a] *= one_over_n;
    umeyama_from_moments(sm, dm, sig, var, with_scaling, T);
}

/* O3D/Core/Utility/Eigen.cpp:137-182, one row */
static inline void jtj_accumulate(const double J[6], double r, double JTJ[36],
                                  double JTr[6], double *r2)
{
    for (int a = 0; a < 6; a++) {
        for (int b = 0; b < 6; b++) JTJ[a * 6 + b] += J[a] * J[b];
        JTr[a] += J[a] * r;
    }
    *r2 += r * r;
}

void vo_jtj_jtr_point_to_plane(const double *src, const double *tgt,
                               const double *tgt_normals, const int32_t *corr,
                               int64_t k, double JTJ[36], double JTr[6],
                               double *r2)
{
    memset(JTJ, 0, 36 * sizeof(double));
    memset(JTr, 0, 6 * sizeof(double));
    double rr = 0.0;
    for (int64_t i = 0; i < k; i++) {
        /* TransformationEstimation.cpp:82-92 */
        const double *vs = src + 3 * (int64_t)corr[2 * i];
        const double *vt = tgt + 3 * (int64_t)corr[2 * i + 1];
        const double *nt = tgt_normals + 3 * (int64_t)corr[2 * i + 1];
        double J[6];
        double r = (vs[0] - vt[0]) * nt[0] + (vs[1] - vt[1]) * nt[1] +
                   (vs[2] - vt[2]) * nt[2];
        cross3(vs, nt, J);
        J[3] = nt[0]; J[4] = nt[1]; J[5] = nt[2];
        jtj_accumulate(J, r, JTJ, JTr, &rr);
    }
    if (r2) *r2 = rr;
}

void vo_jtj_jtr_point_to_point(const double *src, const double *tgt,
                               const int32_t *corr, int64_t k, double JTJ[36],
                               double JTr[6], double *r2)
{
    memset(JTJ, 0, 36 * sizeof(double));
    memset(JTr, 0, 6 * sizeof(double));
    double rr = 0.0;
    for (int64_t i = 0; i < k; i++) {
        const double *vs = src + 3 * (int64_t)corr[2 * i];
        const double *vt = tgt + 3 * (int64_t)corr[2 * i + 1];
        for (int a = 0; a < 3; a++) {
            double n[3] = {0, 0, 0}, J[6];
            n[a] = 1.0;
            cross3(vs, n, J);
            J[3] = n[0]; J[4] = n[1]; J[5] = n[2];
            jtj_accumulate(J, vs[a] - vt[a], JTJ, JTr, &rr);
        }
    }
    if (r2) *r2 = rr;
}

/* O3D/Core/Utility/Eigen.cpp:58-68: R = Rz(x2) * Ry(x1) * Rx(x0) */
void vo_vector6d_to_matrix4d(const double x[6], double T[16])
{
    double ca = cos(x[0]), sa = sin(x[0]);
    double cb = cos(x[1]), sb = sin(x[1]);
    double cg = cos(x[2]), sg = sin(x[2]);
    double Rx[9] = {1, 0, 0, 0, ca, -sa, 0, sa, ca};
    double Ry[9] = {cb, 0, sb, 0, 1, 0, -sb, 0, cb};
    double Rz[9] = {cg, -sg, 0, sg, cg, 0, 0, 0, 1};
    double R[9];
    mat3_mul(Rz, Ry, R);
    mat3_mul(R, Rx, R);
    mat4_identity(T);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[i * 4 + j] = R[i * 3 + j];
        T[i * 4 + 3] = x[3 + i];
    }
}

/* LU with partial pivoting on a 6x6; returns det, solves A x = b. */
static double lu6_solve(const double A[36], const double b[6], double x[6])
{
    double M[36], y[6];
    memcpy(M, A, sizeof(M));
    memcpy(y, b, sizeof(y));
    double det = 1.0;
    for (int c = 0; c < 6; c++) {
        int piv = c;
        for (int r = c + 1; r < 6; r++)
            if (fabs(M[r * 6 + c]) > fabs(M[piv * 6 + c])) piv = r;
        if (M[piv * 6 + c] == 0.0) { det = 0.0; memset(x, 0, 6 * sizeof(double)); return det; }
        if (piv != c) {
            for (int k = 0; k < 6; k++) {
                double t = M[c * 6 + k]; M[c * 6 + k] = M[piv * 6 + k]; M[piv * 6 + k] = t;
            }
            double t = y[c]; y[c] = y[piv]; y[piv] = t;
            det = -det;
        }
        det *= M[c * 6 + c];
        for (int r = c + 1; r < 6; r++) {
            double f = M[r * 6 + c] / M[c * 6 + c];
            for (int k = c; k < 6; k++) M[r * 6 + k] -= f * M[c * 6 + k];
            y[r] -= f * y[c];
        }
    }
    for (int r = 5; r >= 0; r--) {
        double s = y[r];
        for (int k = r + 1; k < 6; k++) s -= M[r * 6 + k] * x[k];
        x[r] = s / M[r * 6 + r];
    }
    return det;
}

/* O3D/Core/Utility/Eigen.cpp:35-56 (det guard) + :88-106 */
int vo_solve_jacobian_system(const double JTJ[36], const double JTr[6],
                             double T[16])
{
    double nb[6], x[6];
    for (int a = 0; a < 6; a++) nb[a] = -JTr[a]; /* Eigen.cpp:97 */
    double det = lu6_solve(JTJ, nb, x);
    if (fabs(det) < 1e-6 || isnan(det) || isinf(det)) { /* Eigen.cpp:41-43 */
        mat4_identity(T);
        return 0;
    }
    vo_vector6d_to_matrix4d(x, T);
    return 1;
}

/* O3D TransformationEstimation.cpp:75-103 */
void vo_point_to_plane_update(const double *src, const double *tgt,
                              const double *tgt_normals, const int32_t *corr,
                              int64_t k, double T[16])
{
    if (k == 0 || !tgt_normals) { mat4_identity(T); return; }
    double JTJ[36], JTr[6], r2;
    vo_jtj_jtr_point_to_plane(src, tgt, tgt_normals, corr, k, JTJ, JTr, &r2);
    vo_solve_jacobian_system(JTJ, JTr, T);
}

static void trace_row(double *trace, int row, const double T[16], double fit,
                      double rmse, int64_t k)
{
    if (!trace) return;
    double *t = trace + (size_t)row * 19;
    memcpy(t, T, 16 * sizeof(double));
    t[16] = fit; t[17] = rmse; t[18] = (double)k;
}

/* O3D/Core/Registration/Registration.cpp:141-186 */
int vo_registration_icp(const double *src, int64_t ns, const double *tgt,
                        int64_t nt, const double *tgt_normals, double max_dist,
                        const double init[16], int estimator, int with_scaling,
                        double rel_fitness, double rel_rmse, int max_iter,
                        int use_grid, vo_result *out, int32_t *idx_out,
                        double *trace)
{
    memset(out, 0, sizeof(*out));
    memcpy(out->T, init, 16 * sizeof(double));
    if (max_dist <= 0.0) return -1;                        /* :148-151 */
    if (estimator == VO_EST_POINT_TO_PLANE && !tgt_normals) /* :152-157 */
        return -1;

    double T[16];
    memcpy(T, init, sizeof(T));                            /* :159 */
    double *pcd = (double *)malloc((size_t)(ns > 0 ? ns : 1) * 3 * sizeof(double));
    int32_t *idx = (int32_t *)malloc((size_t)(ns > 0 ? ns : 1) * sizeof(int32_t));
    int32_t *corr = (int32_t *)malloc((size_t)(ns > 0 ? ns : 1) * 2 * sizeof(int32_t));
    memcpy(pcd, src, (size_t)ns * 3 * sizeof(double));     /* :162 */
    if (!mat4_is_identity(init)) vo_transform_points(pcd, ns, init); /* :163-165 */

    double e2, fit = 0.0, rmse = 0.0;
    int64_t k;
#define NNPASS()                                                               \
    do {                                                                       \
        k = use_grid ? vo_nn_pass_grid(pcd, ns, tgt, nt, max_dist, idx, NULL, &e2) \
                     : vo_nn_pass(pcd, ns, tgt, nt, max_dist, idx, NULL, &e2); \
        if (k == 0) { fit = 0.0; rmse = 0.0; }              /* :87-90 */       \
        else { fit = (double)k / (double)ns; rmse = sqrt(e2 / (double)k); }    \
    } while (0)
    NNPASS();                                              /* :166-168 */
    trace_row(trace, 0, T, fit, rmse, k);
    int it = 0;
    for (int i = 0; i < max_iter; i++) {                   /* :169 */
        int64_t c = 0;
        for (int64_t s = 0; s < ns; s++)
            if (idx[s] >= 0) { corr[2 * c] = (int32_t)s; corr[2 * c + 1] = idx[s]; c++; }
        double upd[16];
        if (estimator == VO_EST_POINT_TO_PLANE)
            vo_point_to_plane_update(pcd, tgt, tgt_normals, corr, c, upd);
        else
            vo_umeyama(pcd, tgt, corr, c, with_scaling, upd); /* :172-173 */
        mat4_mul(upd, T, T);                               /* :174 */
        vo_transform_points(pcd, ns, upd);                 /* :175 */
        double bfit = fit, brmse = rmse;                   /* :176 */
        NNPASS();                                          /* :177-178 */
        it = i + 1;
        trace_row(trace, it, T, fit, rmse, k);
        if (fabs(bfit - fit) < rel_fitness && fabs(brmse - rmse) < rel_rmse)
            break;                                         /* :179-183 */
    }
#undef NNPASS
    memcpy(out->T, T, sizeof(T));
    out->fitness = fit; out->rmse = rmse; out->k = k; out->iters = it;
    if (idx_out) memcpy(idx_out, idx, (size_t)ns * sizeof(int32_t));
    free(pcd); free(idx); free(corr);
    return 0;
}

/* src/annotation.cpp:29-64 */
int vo_register_model_to_scene(const double *model, int64_t ns,
                               const double *scene, int64_t nt,
                               const double *scene_normals, int level,
                               double max_dist, int point_to_plane,
                               double rel_fitness, double rel_rmse,
                               int max_iter, vo_result *best, int *best_level)
{
    const double interval = 2.0 * M_PI / (double)level;    /* :35 */
    vo_result b;
    memset(&b, 0, sizeof(b));
    mat4_identity(b.T);       /* default RegistrationResult: identity, K=0 */
    int bl = -1;
    for (int i = 0; i < level; i++) {
        double a = interval * i, c = cos(a), s = sin(a);
        double init[16];
        mat4_identity(init);  /* AngleAxis(a, UnitY): :40-43 */
        init[0] = c; init[2] = s; init[8] = -s; init[10] = c;
        vo_result r;
        int rc = vo_registration_icp(
            model, ns, scene, nt, scene_normals, max_dist, init,
            point_to_plane ? VO_EST_POINT_TO_PLANE : VO_EST_POINT_TO_POINT, 0,
            rel_fitness, rel_rmse, max_iter, 1, &r, NULL, NULL);
        if (rc != 0) { r.k = 0; }
        if (r.k > b.k) { b = r; bl = i; }                  /* :59-61 strict > */
    }
    *best = b;
    if (best_level) *best_level = bl;
    return 0;
}

/* ---- O3D/Core/Geometry/DownSample.cpp:179-220 -------------------------- */
typedef struct { int32_t v[3]; int64_t i; } vox_key;
static int vox_cmp(const void *a, const void *b)
{
    const vox_key *x = (const vox_key *)a, *y = (const vox_key *)b;
    for (int k = 0; k < 3; k++)
        if (x->v[k] != y->v[k]) return x->v[k] < y->v[k] ? -1 : 1;
    return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0);   /* input order inside a voxel */
}

int64_t vo_voxel_down_sample(const double *xyz, const double *normals, const double *colors,
                             int64_t n, double voxel_size, double *out_xyz,
                             double *out_normals, double *out_colors)
{
    if (voxel_size <= 0.0 || n <= 0) return 0;               /* :183-186 */
    double mn[3], mx[3];
    for (int a = 0; a < 3; a++) { mn[a] = mx[a] = xyz[a]; }
    for (int64_t i = 1; i < n; i++)
        for (int a = 0; a < 3; a++) {
            if (xyz[3 * i + a] < mn[a]) mn[a] = xyz[3 * i + a];
            if (xyz[3 * i + a] > mx[a]) mx[a] = xyz[3 * i + a];
        }
    double vmin[3], ext = 0.0;
    for (int a = 0; a < 3; a++) {                            /* :189-190 */
        vmin[a] = mn[a] - voxel_size * 0.5;
        double e = (mx[a] + voxel_size * 0.5) - vmin[a];
        if (e > ext) ext = e;
    }
    if (voxel_size * 2147483647.0 < ext) return 0;           /* :191-195 */
    vox_key *k = (vox_key *)malloc((size_t)n * sizeof(vox_key));
    for (int64_t i = 0; i < n; i++) {                        /* :200-205 */
        for (int a = 0; a < 3; a++)
            k[i].v[a] = (int32_t)floor((xyz[3 * i + a] - vmin[a]) / voxel_size);
        k[i].i = i;
    }
    qsort(k, (size_t)n, sizeof(vox_key), vox_cmp);
    int64_t m = 0;
    for (int64_t s = 0; s < n;) {
        int64_t e = s;
        double p[3] = {0, 0, 0}, nn[3] = {0, 0, 0}, c[3] = {0, 0, 0};
        while (e < n && k[e].v[0] == k[s].v[0] && k[e].v[1] == k[s].v[1] && k[e].v[2] == k[s].v[2]) {
            const int64_t i = k[e].i;                        /* AddPoint, :50-65 */
            for (int a = 0; a < 3; a++) p[a] += xyz[3 * i + a];
            if (normals && !isnan(normals[3 * i]) && !isnan(normals[3 * i + 1]) && !isnan(normals[3 * i + 2]))
                for (int a = 0; a < 3; a++) nn[a] += normals[3 * i + a];
            if (colors) for (int a = 0; a < 3; a++) c[a] += colors[3 * i + a];
            e++;
        }
        const double cnt = (double)(e - s);
        for (int a = 0; a < 3; a++) out_xyz[3 * m + a] = p[a] / cnt;           /* :67-70 */
        if (normals && out_normals) {                                          /* :72-75 normalized() */
            const double z = nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2];
            const double inv = z > 0.0 ? 1.0 / sqrt(z) : 1.0;
            for (int a = 0; a < 3; a++) out_normals[3 * m + a] = z > 0.0 ? nn[a] / sqrt(z) : nn[a];
            (void)inv;
        }
        if (colors && out_colors) for (int a = 0; a < 3; a++) out_colors[3 * m + a] = c[a] / cnt;
        m++;
        s = e;
    }
    free(k);
    return m;
}

/* ---- include/geometry.h ------------------------------------------------ */
int64_t vo_sample_mesh(const double *V, const int32_t *F, int64_t nf, const double *u,
                       int64_t n, int quirks, double *out_xyz)
{
    if (nf <= 0 || n <= 0) return 0;
    double *cdf = (double *)malloc((size_t)nf * sizeof(double));
    double total = 0.0;
    for (int64_t i = 0; i < nf; i++) {                       /* geometry.h:33-39 */
        const double *a = V + 3 * (int64_t)F[3 * i], *b = V + 3 * (int64_t)F[3 * i + 1],
                     *c = V + 3 * (int64_t)F[3 * i + 2];
        double e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
        double e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]}, cr[3];
        cross3(e1, e2, cr);
        cdf[i] = 0.5 * sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
        total += cdf[i];
    }
    cdf[0] /= total;                                         /* :40-43 */
    for (int64_t i = 1; i < nf; i++) cdf[i] = cdf[i - 1] + cdf[i] / total;
    int64_t m = 0;
    for (int64_t i = 0; i < n; i++) {
        const double r = u[3 * i];
        double a = u[3 * i + 1], b = u[3 * i + 2];
        int64_t k = -1;
        if (quirks) {                                        /* :52-54 */
            for (int64_t j = 0; j + 1 < nf; j++)
                if (cdf[j] <= r && r < cdf[j + 1]) { k = j; break; }
        } else {
            for (int64_t j = 0; j < nf; j++)
                if (r < cdf[j]) { k = j; break; }
            if (k < 0) k = nf - 1;
            if (a + b > 1.0) { a = 1.0 - a; b = 1.0 - b; }
        }
        if (k < 0) continue;
        const double *v0 = V + 3 * (int64_t)F[3 * k], *v1 = V + 3 * (int64_t)F[3 * k + 1],
                     *v2 = V + 3 * (int64_t)F[3 * k + 2];
        for (int d = 0; d < 3; d++)                          /* :57 */
            out_xyz[3 * m + d] = v0[d] + a * (v1[d] - v0[d]) + b * (v2[d] - v0[d]);
        m++;
    }
    free(cdf);
    return m;
}

static inline double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

double vo_point_triangle_sqdist(const double p[3], const double a[3], const double b[3],
                                const double c[3], double closest[3])
{
    double ab[3], ac[3], ap[3], bp[3], cp[3], q[3];
    for (int i = 0; i < 3; i++) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; ap[i] = p[i] - a[i]; }
    const double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
    if (d1 <= 0.0 && d2 <= 0.0) { memcpy(q, a, sizeof(q)); goto done; }          /* vertex a */
    for (int i = 0; i < 3; i++) bp[i] = p[i] - b[i];
    {
        const double d3 = dot3(ab, bp), d4 = dot3(ac, bp);
        if (d3 >= 0.0 && d4 <= d3) { memcpy(q, b, sizeof(q)); goto done; }       /* vertex b */
        const double vc = d1 * d4 - d3 * d2;
        if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {                               /* edge ab */
            const double v = d1 / (d1 - d3);
            for (int i = 0; i < 3; i++) q[i] = a[i] + v * ab[i];
            goto done;
        }
        for (int i = 0; i < 3; i++) cp[i] = p[i] - c[i];
        const double d5 = dot3(ab, cp), d6 = dot3(ac, cp);
        if (d6 >= 0.0 && d5 <= d6) { memcpy(q, c, sizeof(q)); goto done; }       /* vertex c */
        const double vb = d5 * d2 - d1 * d6;
        if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {                               /* edge ac */
            const double w = d2 / (d2 - d6);
            for (int i = 0; i < 3; i++) q[i] = a[i] + w * ac[i];
            goto done;
        }
        const double va = d3 * d6 - d5 * d4;
        if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {                 /* edge bc */
            const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
            for (int i = 0; i < 3; i++) q[i] = b[i] + w * (c[i] - b[i]);
            goto done;
        }
        const double denom = 1.0 / (va + vb + vc);                               /* interior */
        const double v = vb * denom, w = vc * denom;
        for (int i = 0; i < 3; i++) q[i] = a[i] + ab[i] * v + ac[i] * w;
    }
done:
    if (closest) memcpy(closest, q, sizeof(q));
    {
        const double dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
        return dx * dx + dy * dy + dz * dz;
    }
}

void vo_point_mesh_sqdist(const double *P, int64_t np, const double *V, const int32_t *F,
                          int64_t nf, double *d2, int32_t *face, double *closest)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < np; i++) {
        double best = INFINITY, bq[3] = {0, 0, 0};
        int32_t bf = -1;
        for (int64_t f = 0; f < nf; f++) {
            double q[3];
            const double d = vo_point_triangle_sqdist(P + 3 * i, V + 3 * (int64_t)F[3 * f],
                                                      V + 3 * (int64_t)F[3 * f + 1],
                                                      V + 3 * (int64_t)F[3 * f + 2], q);
            if (d < best) { best = d; bf = (int32_t)f; memcpy(bq, q, sizeof(q)); }
        }
        d2[i] = best;
        if (face) face[i] = bf;
        if (closest) memcpy(closest + 3 * i, bq, sizeof(bq));
    }
}

static int cmp_double(const void *a, const void *b)
{
    const double x = *(const double *)a, y = *(const double *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

void vo_error_metric(const double *errors, int64_t n, double out[5])
{
    double mean = 0.0, sq = 0.0, mn = 1.79769313486231570815e308, mx = -1.79769313486231570815e308;
    for (int64_t i = 0; i < n; i++) {                       /* geometry.h:90-95 */
        mean += errors[i];
        sq += errors[i] * errors[i];
        if (errors[i] < mn) mn = errors[i];
        if (errors[i] > mx) mx = errors[i];
    }
    mean /= (double)n;
    double *s = (double *)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    memcpy(s, errors, (size_t)n * sizeof(double));
    qsort(s, (size_t)n, sizeof(double), cmp_double);
    out[0] = mean;
    out[1] = sqrt(sq / (double)n - mean * mean);            /* :97 */
    out[2] = n > 0 ? s[n >> 1] : 0.0;                       /* :99 */
    out[3] = mn;
    out[4] = mx;
    free(s);
}

/* ---- open3d::EstimateNormals (O3D/Core/Geometry/EstimateNormals.cpp) ---- */

static double vo_sqr(double x) { return x * x; }

/* FastEigen3x3, EstimateNormals.cpp:40-83: eigenvector of the smallest eigenvalue of the symmetric A
 * by the trigonometric closed form; returns 0 for a zero vector.  A row-major. */
static int vo_fast_eigen_3x3(const double A[9], double out[3])
{
    const double p1 = vo_sqr(A[1]) + vo_sqr(A[2]) + vo_sqr(A[5]);            /* :44 */
    const double trace = A[0] + A[4] + A[8];
    double ev0, ev1, ev2;
    if (p1 == 0.0) {                                                         /* :46-49 */
        ev2 = fmin(A[0], fmin(A[4], A[8]));
        ev0 = fmax(A[0], fmax(A[4], A[8]));
        ev1 = trace - ev0 - ev2;
    } else {
        const double q = trace / 3.0;                                        /* :51 */
        const double p2 = vo_sqr(A[0] - q) + vo_sqr(A[4] - q) + vo_sqr(A[8] - q) + 2 * p1;
        const double p = sqrt(p2 / 6.0);
        const double ip = 1.0 / p;
        double B[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) B[i * 3 + j] = ip * (A[i * 3 + j] - (i == j ? q : 0.0));   /* :55 */
        /* Eigen's fixed-size 3x3 determinant (Eigen/src/LU/Determinant.h: bruteforce_det3_helper) */
        const double det = B[0] * (B[4] * B[8] - B[5] * B[7]) - B[1] * (B[3] * B[8] - B[5] * B[6]) +
                           B[2] * (B[3] * B[7] - B[4] * B[6]);
        const double r = det / 2.0;                                          /* :56 */
        double phi;
        if (r <= -1) phi = M_PI / 3.0;
        else if (r >= 1) phi = 0.0;
        else phi = acos(r) / 3.0;
        ev0 = q + 2.0 * p * cos(phi);                                        /* :65-67 */
        ev2 = q + 2.0 * p * cos(phi + 2.0 * M_PI / 3.0);
        ev1 = q * 3.0 - ev0 - ev2;
    }
    (void)ev2;
    /* (A - I ev0) * (A.col(0) - (ev1, 0, 0))   :70-72 */
    const double v[3] = {A[0] - ev1, A[3], A[6]};
    double e[3];
    for (int i = 0; i < 3; i++) {
        const double m0 = A[i * 3] - (i == 0 ? ev0 : 0.0), m1 = A[i * 3 + 1] - (i == 1 ? ev0 : 0.0),
                     m2 = A[i * 3 + 2] - (i == 2 ? ev0 : 0.0);
        e[i] = m0 * v[0] + m1 * v[1] + m2 * v[2];
    }
    const double len = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    if (len == 0.0) return 0;                                                /* :74-75 */
    out[0] = e[0] / len; out[1] = e[1] / len; out[2] = e[2] / len;
    return 1;
}

typedef struct { double d; int32_t i; } vo_nbr;
static int cmp_nbr(const void *a, const void *b)
{
    const vo_nbr *x = (const vo_nbr *)a, *y = (const vo_nbr *)b;
    if (x->d < y->d) return -1;
    if (x->d > y->d) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

/* EstimateNormals.cpp:114-153 with the neighbour searches of KDTreeFlann.cpp:114-189 restated as an
 * exhaustive scan: search_type 0 = the knn nearest (SearchKNN), 1 = every point with
 * d2 < (double)(float)(r*r) (SearchRadius: flann radiusSearch is strict), 2 = the knn nearest of
 * those (SearchHybrid); results ascending in (d2, index) -- flann orders exactly equal distances by
 * tree traversal instead.  d2 is flann's L2 (dist.h:159-176).  normals_in may be NULL. */
void vo_estimate_normals(const double *xyz, int64_t n, const double *normals_in, int search_type,
                         int knn, double radius, double *out)
{
    const double r2 = (double)(float)(radius * radius);
    #pragma omp parallel
    {
        vo_nbr *nb = (vo_nbr *)malloc(sizeof(vo_nbr) * (size_t)(n > 0 ? n : 1));
        #pragma omp for schedule(static)
        for (int64_t i = 0; i < n; i++) {
            const double *q = xyz + 3 * i;
            int64_t m = 0;
            for (int64_t j = 0; j < n; j++) {
                const double dx = q[0] - xyz[3 * j], dy = q[1] - xyz[3 * j + 1], dz = q[2] - xyz[3 * j + 2];
                double d = dx * dx;
                d += dy * dy;
                d += dz * dz;
                if (search_type != 0 && !(d < r2)) continue;
                nb[m].d = d; nb[m].i = (int32_t)j; m++;
            }
            qsort(nb, (size_t)m, sizeof(vo_nbr), cmp_nbr);
            if (search_type != 1 && m > knn) m = knn < 0 ? 0 : knn;
            if (search_type != 1 && knn < 0) m = 0;
            double nrm[3] = {0.0, 0.0, 1.0};                                 /* :147-149 */
            if (m >= 3) {
                double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                for (int64_t k = 0; k < m; k++) {                            /* :95-105 */
                    const double *p = xyz + 3 * (int64_t)nb[k].i;
                    c[0] += p[0]; c[1] += p[1]; c[2] += p[2];
                    c[3] += p[0] * p[0]; c[4] += p[0] * p[1]; c[5] += p[0] * p[2];
                    c[6] += p[1] * p[1]; c[7] += p[1] * p[2]; c[8] += p[2] * p[2];
                }
                for (int k = 0; k < 9; k++) c[k] /= (double)m;               /* :106 */
                double A[9];
                A[0] = c[3] - c[0] * c[0];
                A[4] = c[6] - c[1] * c[1];
                A[8] = c[8] - c[2] * c[2];
                A[1] = A[3] = c[4] - c[0] * c[1];
                A[2] = A[6] = c[5] - c[0] * c[2];
                A[5] = A[7] = c[7] - c[1] * c[2];
                if (!vo_fast_eigen_3x3(A, nrm)) {                             /* :134-140 */
                    if (normals_in) { nrm[0] = normals_in[3 * i]; nrm[1] = normals_in[3 * i + 1]; nrm[2] = normals_in[3 * i + 2]; }
                    else { nrm[0] = 0.0; nrm[1] = 0.0; nrm[2] = 1.0; }
                }
                if (normals_in &&
                    nrm[0] * normals_in[3 * i] + nrm[1] * normals_in[3 * i + 1] + nrm[2] * normals_in[3 * i + 2] < 0.0) {
                    nrm[0] *= -1.0; nrm[1] *= -1.0; nrm[2] *= -1.0;           /* :141-143 */
                }
            }
            out[3 * i] = nrm[0]; out[3 * i + 1] = nrm[1]; out[3 * i + 2] = nrm[2];
        }
        free(nb);
    }
}

/* ---- core/rodrigues.h ------------------------------------------------ */
void vo_hat(const double u[3], double M[9]) /* rodrigues.h:8-15 */
{
    M[0] = 0;     M[1] = -u[2]; M[2] = u[1];
    M[3] = u[2];  M[4] = 0;     M[5] = -u[0];
    M[6] = -u[1]; M[7] = u[0];  M[8] = 0;
}

void vo_vee(const double R[9], double v[3]) /* rodrigues.h:37-41 */
{
    v[0] = R[2 * 3 + 1] - R[1 * 3 + 2];
    v[1] = R[0 * 3 + 2] - R[2 * 3 + 0];
    v[2] = R[1 * 3 + 0] - R[0 * 3 + 1];
}

static const double DHAT[27] = { /* rodrigues.h:17-29, 9x3 */
    0, 0, 0,  0, 0, -1,  0, 1, 0,
    0, 0, 1,  0, 0, 0,  -1, 0, 0,
    0, -1, 0, 1, 0, 0,   0, 0, 0};
static const double DVEE[27] = { /* rodrigues.h:43-49, 3x9 */
    0, 0, 0, 0, 0, -1, 0, 1, 0,
    0, 0, 1, 0, 0, 0, -1, 0, 0,
    0, -1, 0, 1, 0, 0, 0, 0, 0};

void vo_rodrigues(const double w[3], double R[9], double *dR_dw)
{
    /* rodrigues.h:143-182 */
    double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double what[9];
    if (th < 1e-8) { /* :152-160: R = I + hat(w) */
        vo_hat(w, what);
        for (int a = 0; a < 9; a++) R[a] = what[a];
        R[0] += 1; R[4] += 1; R[8] += 1;
        if (dR_dw) memcpy(dR_dw, DHAT, sizeof(DHAT));
        return;
    }
    double inv_th = 1.0 / th;
    double u[3] = {w[0] * inv_th, w[1] * inv_th, w[2] * inv_th};
    double s = sin(th), c = cos(th);
    double uh[9], uh2[9];
    vo_hat(u, uh);
    mat3_mul(uh, uh, uh2);
    for (int a = 0; a < 9; a++) R[a] = uh[a] * s + uh2[a] * (1 - c); /* :168 */
    R[0] += 1; R[4] += 1; R[8] += 1;
    if (!dR_dw) return;
    /* D = dAB_dA(uh,uh) + dAB_dB(uh,uh), 9x9 (rodrigues.h:87-141) */
    double D[81];
    memset(D, 0, sizeof(D));
    for (int n = 0; n < 3; n++)
        for (int p = 0; p < 3; p++)
            for (int m = 0; m < 3; m++) {
                D[(n * 3 + p) * 9 + (n * 3 + m)] += uh[m * 3 + p]; /* dAB_dA */
                D[(n * 3 + p) * 9 + (m * 3 + p)] += uh[n * 3 + m]; /* dAB_dB */
            }
    double dR_du[27]; /* :170-171 */
    for (int r = 0; r < 9; r++)
        for (int k = 0; k < 3; k++) {
            double a = 0.0;
            for (int j = 0; j < 9; j++) a += D[r * 9 + j] * DHAT[j * 3 + k];
            dR_du[r * 3 + k] = s * DHAT[r * 3 + k] + (1 - c) * a;
        }
    double du_dw[9]; /* :172 */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            du_dw[i * 3 + j] = inv_th * ((i == j ? 1.0 : 0.0) - u[i] * u[j]);
    double dR_dth[9]; /* :173-174 row-major vec */
    for (int a = 0; a < 9; a++) dR_dth[a] = uh[a] * c + uh2[a] * s;
    for (int r = 0; r < 9; r++) /* :176 */
        for (int k = 0; k < 3; k++) {
            double a = 0.0;
            for (int j = 0; j < 3; j++) a += dR_du[r * 3 + j] * du_dw[j * 3 + k];
            dR_dw[r * 3 + k] = a + dR_dth[r] * u[k];
        }
}

void vo_invrodrigues(const double R[9], double w[3], double *dw_dR)
{
    /* rodrigues.h:184-226 */
    double tmp = 0.5 * (R[0] + R[4] + R[8] - 1);
    double v[3];
    vo_vee(R, v);
    if (tmp > 1.0 - 1e-10) { /* :195-202 */
        for (int a = 0; a < 3; a++) w[a] = 0.5 * v[a];
        if (dw_dR) for (int a = 0; a < 27; a++) dw_dR[a] = 0.5 * DVEE[a];
        return;
    }
    double th = acos(tmp), s = sin(th), is = 1.0 / s;
    double u[3] = {0.5 * v[0] * is, 0.5 * v[1] * is, 0.5 * v[2] * is};
    for (int a = 0; a < 3; a++) w[a] = th * u[a];
    if (!dw_dR) return;
    double dth_dtmp = -1.0 / sqrt(1 - tmp * tmp);
    double dth_dR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int a = 0; a < 9; a++) dth_dR[a] *= 0.5 * dth_dtmp;
    double cth = cos(th);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 9; j++) {
            double du = 0.5 * (DVEE[i * 9 + j] * is - v[i] * cth * is * is * dth_dR[j]);
            dw_dR[i * 9 + j] = u[i] * dth_dR[j] + th * du; /* :222 */
        }
}

/* core/se3.h:96-100 */
void vo_se3_compose(const double Ra[9], const double ta[3], const double Rb[9],
                    const double tb[3], double R[9], double t[3])
{
    double Rr[9], tr[3];
    mat3_mul(Ra, Rb, Rr);
    for (int i = 0; i < 3; i++)
        tr[i] = Ra[i * 3] * tb[0] + Ra[i * 3 + 1] * tb[1] + Ra[i * 3 + 2] * tb[2] + ta[i];
    memcpy(R, Rr, sizeof(Rr));
    memcpy(t, tr, sizeof(tr));
}

/* core/se3.h:103-106 */
void vo_se3_act(const double R[9], const double t[3], const double v[3],
                double out[3])
{
    double o[3];
    for (int i = 0; i < 3; i++)
        o[i] = R[i * 3] * v[0] + R[i * 3 + 1] * v[1] + R[i * 3 + 2] * v[2] + t[i];
    memcpy(out, o, sizeof(o));
}

/* core/se3.h:108-110 */
void vo_se3_inv(const double R[9], const double t[3], double Ri[9],
                double ti[3])
{
    double Rt[9], tt[3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rt[i * 3 + j] = R[j * 3 + i];
    for (int i = 0; i < 3; i++)
        tt[i] = -(Rt[i * 3] * t[0] + Rt[i * 3 + 1] * t[1] + Rt[i * 3 + 2] * t[2]);
    memcpy(Ri, Rt, sizeof(Rt));
    memcpy(ti, tt, sizeof(tt));
}

/* ------------------------------------------------------------------ */
/* (B) kernel specification                                            */
/* ------------------------------------------------------------------ */
static inline void vk_transform_f32(const float T[12], const float *s,
                                    float p[3])
{
    p[0] = fmaf(T[0], s[0], fmaf(T[1], s[1], fmaf(T[2], s[2], T[3])));
    p[1] = fmaf(T[4], s[0], fmaf(T[5], s[1], fmaf(T[6], s[2], T[7])));
    p[2] = fmaf(T[8], s[0], fmaf(T[9], s[1], fmaf(T[10], s[2], T[11])));
}

static inline float vk_d2_f32(const float p[3], const float *q)
{
    float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

int64_t vk_nn_pass_f32(const float *src, int64_t ns, int sstride,
                       const float *tgt, int64_t nt, int tstride,
                       const float T32[12], float r2f, int32_t *idx, float *d2)
{
    int64_t k = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(+ : k)
#endif
    for (int64_t i = 0; i < ns; i++) {
        float p[3];
        vk_transform_f32(T32, src + (size_t)i * sstride, p);
        float best = r2f;
        int32_t bi = -1;
        for (int64_t j = 0; j < nt; j++) {
            float d = vk_d2_f32(p, tgt + (size_t)j * tstride);
            if (d < best) { best = d; bi = (int32_t)j; }
        }
        idx[i] = bi;
        if (d2) d2[i] = best;
        if (bi >= 0) k++;
    }
    return k;
}

int64_t vk_nn_pass_f32_grid(const float *src, int64_t ns, int sstride,
                            const float *tgt, int64_t nt, int tstride,
                            const float T32[12], float r2f, int32_t *idx,
                            float *d2)
{
    if (nt <= 0 || !(r2f > 0.0f)) {
        for (int64_t i = 0; i < ns; i++) { idx[i] = -1; if (d2) d2[i] = r2f; }
        return 0;
    }
    double *tx = (double *)malloc((size_t)nt * 3 * sizeof(double));
    double *ty = tx + nt, *tz = ty + nt;
    for (int64_t j = 0; j < nt; j++) {
        tx[j] = tgt[(size_t)j * tstride];
        ty[j] = tgt[(size_t)j * tstride + 1];
        tz[j] = tgt[(size_t)j * tstride + 2];
    }
    vo_grid g;
    /* fp32 d2 < r2f can hold for a true distance a few ulp above sqrt(r2f):
     * widen the cell by 1e-4 relative + an absolute pad */
    grid_build(&g, tx, ty, tz, nt, sqrt((double)r2f) * (1.0 + 1e-4) + 1e-30);
    free(tx);
    int64_t k = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : k)
#endif
    for (int64_t i = 0; i < ns; i++) {
        float p[3];
        vk_transform_f32(T32, src + (size_t)i * sstride, p);
        int c[3];
        for (int a = 0; a < 3; a++) c[a] = grid_cell_coord(&g, (double)p[a], a);
        float best = r2f;
        int32_t bi = -1;
        for (int dz = -1; dz <= 1; dz++) {
            int z = c[2] + dz;
            if (z < 0 || z >= g.dim[2]) continue;
            for (int dy = -1; dy <= 1; dy++) {
                int y = c[1] + dy;
                if (y < 0 || y >= g.dim[1]) continue;
                for (int dx = -1; dx <= 1; dx++) {
                    int x = c[0] + dx;
                    if (x < 0 || x >= g.dim[0]) continue;
                    int64_t cc = ((int64_t)z * g.dim[1] + y) * g.dim[0] + x;
                    for (int32_t s = g.start[cc]; s < g.start[cc + 1]; s++) {
                        int32_t j = g.order[s];
                        float d = vk_d2_f32(p, tgt + (size_t)j * tstride);
                        if (d < best || (d == best && bi >= 0 && j < bi)) {
                            best = d; bi = j;
                        }
                    }
                }
            }
        }
        idx[i] = bi;
        if (d2) d2[i] = best;
        if (bi >= 0) k++;
    }
    grid_free(&g);
    return k;
}

void vk_reduce_stats(const float *src, int64_t ns, int sstride,
                     const float *tgt, int tstride, const int32_t *idx,
                     const double T64[12], const double *offset,
                     double stats[VK_NSTATS])
{
    const double off[3] = {offset ? offset[0] : 0.0, offset ? offset[1] : 0.0,
                           offset ? offset[2] : 0.0};
    double JTJ[36], JTr[6], qp[9], r2 = 0.0;
    memset(JTJ, 0, sizeof(JTJ));
    memset(JTr, 0, sizeof(JTr));
    memset(qp, 0, sizeof(qp));
    int64_t k = 0;
    for (int64_t i = 0; i < ns; i++) {
        if (idx[i] < 0) continue;
        const float *s = src + (size_t)i * sstride;
        const float *qf = tgt + (size_t)idx[i] * tstride;
        double p[3], q[3];
        for (int a = 0; a < 3; a++) {
            p[a] = T64[a * 4] * (double)s[0] + T64[a * 4 + 1] * (double)s[1] +
                   T64[a * 4 + 2] * (double)s[2] + T64[a * 4 + 3] + off[a];
            q[a] = (double)qf[a] + off[a];
        }
        for (int a = 0; a < 3; a++) {
            double n[3] = {0, 0, 0}, J[6];
            n[a] = 1.0;
            cross3(p, n, J);
            J[3] = n[0]; J[4] = n[1]; J[5] = n[2];
            jtj_accumulate(J, p[a] - q[a], JTJ, JTr, &r2);
            for (int b = 0; b < 3; b++) qp[a * 3 + b] += q[a] * p[b];
        }
        k++;
    }
    stats[0] = (double)k;
    stats[1] = r2;
    int o = 2;
    for (int a = 0; a < 6; a++)
        for (int b = a; b < 6; b++) stats[o++] = JTJ[a * 6 + b];
    for (int a = 0; a < 6; a++) stats[o++] = JTr[a];
    for (int a = 0; a < 9; a++) stats[o++] = qp[a];
}

static void stats_unpack(const double stats[VK_NSTATS], double JTJ[36],
                         double JTr[6], double qp[9])
{
    int o = 2;
    for (int a = 0; a < 6; a++)
        for (int b = a; b < 6; b++) {
            JTJ[a * 6 + b] = stats[o];
            JTJ[b * 6 + a] = stats[o];
            o++;
        }
    for (int a = 0; a < 6; a++) JTr[a] = stats[o++];
    for (int a = 0; a < 9; a++) qp[a] = stats[o++];
}

void vk_solve_kabsch_from_stats(const double stats[VK_NSTATS], int with_scaling,
                                double T[16])
{
    double K = stats[0];
    if (!(K > 0)) { mat4_identity(T); return; }
    double JTJ[36], JTr[6], qp[9];
    stats_unpack(stats, JTJ, JTr, qp);
    /* top-right block of JTJ is hat(sum p) */
    double P[3] = {JTJ[2 * 6 + 4], JTJ[0 * 6 + 5], JTJ[1 * 6 + 3]};
    double Q[3] = {P[0] - JTr[3], P[1] - JTr[4], P[2] - JTr[5]};
    double sum_p2 = 0.5 * (JTJ[0] + JTJ[7] + JTJ[14]); /* tr(|p|^2 I - pp^T)=2|p|^2 */
    double pm[3], qm[3], sig[9];
    for (int a = 0; a < 3; a++) { pm[a] = P[a] / K; qm[a] = Q[a] / K; }
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) sig[a * 3 + b] = qp[a * 3 + b] / K - qm[a] * pm[b];
    double var = sum_p2 / K - (pm[0] * pm[0] + pm[1] * pm[1] + pm[2] * pm[2]);
    umeyama_from_moments(pm, qm, sig, var, with_scaling, T);
}

int vk_solve_gn_from_stats(const double stats[VK_NSTATS], double T[16])
{
    if (!(stats[0] > 0)) { mat4_identity(T); return 0; }
    double JTJ[36], JTr[6], qp[9];
    stats_unpack(stats, JTJ, JTr, qp);
    return vo_solve_jacobian_system(JTJ, JTr, T);
}

int vk_registration_icp(const double *src, int64_t ns, const double *tgt,
                        int64_t nt, double max_dist, const double init[16],
                        int with_scaling, double rel_fitness, double rel_rmse,
                        int max_iter, int use_grid, vo_result *out,
                        int32_t *idx_out, double *trace)
{
    memset(out, 0, sizeof(*out));
    memcpy(out->T, init, 16 * sizeof(double));
    if (max_dist <= 0.0) return -1;
    /* centre on the target centroid: sequential f64 sum in index order */
    double c[3] = {0, 0, 0};
    for (int64_t j = 0; j < nt; j++)
        for (int a = 0; a < 3; a++) c[a] += tgt[3 * j + a];
    if (nt > 0) for (int a = 0; a < 3; a++) c[a] /= (double)nt;
    float *s32 = (float *)malloc((size_t)(ns > 0 ? ns : 1) * 4 * sizeof(float));
    float *t32 = (float *)malloc((size_t)(nt > 0 ? nt : 1) * 4 * sizeof(float));
    int32_t *idx = (int32_t *)malloc((size_t)(ns > 0 ? ns : 1) * sizeof(int32_t));
    for (int64_t i = 0; i < ns; i++) {
        for (int a = 0; a < 3; a++) s32[4 * i + a] = (float)(src[3 * i + a] - c[a]);
        s32[4 * i + 3] = 0.0f;
    }
    for (int64_t j = 0; j < nt; j++) {
        for (int a = 0; a < 3; a++) t32[4 * j + a] = (float)(tgt[3 * j + a] - c[a]);
        t32[4 * j + 3] = 0.0f;
    }
    /* centred transform: x' = x - c  =>  t' = R c + t - c */
    double Tc[16];
    memcpy(Tc, init, sizeof(Tc));
    for (int a = 0; a < 3; a++)
        Tc[a * 4 + 3] = init[a * 4] * c[0] + init[a * 4 + 1] * c[1] +
                        init[a * 4 + 2] * c[2] + init[a * 4 + 3] - c[a];
    const float r2f = (float)(max_dist * max_dist);
    double stats[VK_NSTATS], fit = 0, rmse = 0;
    int64_t k = 0;
    double Tw[16]; /* uncentred */
#define UNCENTRE()                                                             \
    do {                                                                       \
        memcpy(Tw, Tc, sizeof(Tw));                                            \
        for (int a = 0; a < 3; a++)                                            \
            Tw[a * 4 + 3] = Tc[a * 4 + 3] - (Tc[a * 4] * c[0] + Tc[a * 4 + 1] * c[1] + \
                                             Tc[a * 4 + 2] * c[2]) + c[a];     \
    } while (0)
#define PASS()                                                                 \
    do {                                                                       \
        float T32[12];                                                         \
        for (int a = 0; a < 12; a++) T32[a] = (float)Tc[a];                    \
        k = use_grid ? vk_nn_pass_f32_grid(s32, ns, 4, t32, nt, 4, T32, r2f, idx, NULL) \
                     : vk_nn_pass_f32(s32, ns, 4, t32, nt, 4, T32, r2f, idx, NULL);    \
        vk_reduce_stats(s32, ns, 4, t32, 4, idx, Tc, NULL, stats);                   \
        if (k == 0) { fit = 0; rmse = 0; }                                     \
        else { fit = (double)k / (double)ns; rmse = sqrt(stats[1] / (double)k); } \
    } while (0)
    PASS();
    UNCENTRE();
    trace_row(trace, 0, Tw, fit, rmse, k);
    int it = 0;
    for (int i = 0; i < max_iter; i++) {
        double upd[16];
        vk_solve_kabsch_from_stats(stats, with_scaling, upd);
        mat4_mul(upd, Tc, Tc);
        double bfit = fit, brmse = rmse;
        PASS();
        it = i + 1;
        UNCENTRE();
        trace_row(trace, it, Tw, fit, rmse, k);
        if (fabs(bfit - fit) < rel_fitness && fabs(brmse - rmse) < rel_rmse) break;
    }
#undef PASS
    UNCENTRE();
#undef UNCENTRE
    memcpy(out->T, Tw, sizeof(Tw));
    out->fitness = fit; out->rmse = rmse; out->k = k; out->iters = it;
    if (idx_out) memcpy(idx_out, idx, (size_t)ns * sizeof(int32_t));
    free(s32); free(t32); free(idx);
    return 0;
}
