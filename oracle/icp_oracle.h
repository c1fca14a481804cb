/*
 * icp_oracle.h -- CPU oracle for the VISMA orientation-constrained ICP path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker.  The product path (visma_amd/, include/)
 * never links, imports or calls this file.
 *
 * Two groups of functions:
 *
 *  (A) vo_*  : f64 restatement of the reference algorithm, function by
 *              function, each citing the reference file:line it follows
 *              (paths relative to /root/reference; O3D = thirdparty/Open3D/src).
 *              Pinned against the compiled reference (oracle/_ref) and against
 *              the golden fixtures in tests/golden/.
 *
 *  (B) vk_*  : "kernel specification" -- the exact fp32 arithmetic the HIP
 *              kernels are required to perform (operation order, fma
 *              placement, tie rule), so GPU correspondences can be checked
 *              bit-for-bit, plus the f64 statistics the reduction kernel must
 *              reproduce to summation-order accuracy.
 *
 * All 4x4 matrices are ROW-MAJOR double[16].  Point arrays are AoS xyz.
 */
#ifndef VISMA_ICP_ORACLE_H
#define VISMA_ICP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- (A) reference-semantics, f64 ------------------------------------- */

/* O3D/Core/Geometry/PointCloud.cpp:75-87  (w row ignored; normals use w=0) */
void vo_transform_points(double *xyz, int64_t n, const double T[16]);
void vo_transform_normals(double *nxyz, int64_t n, const double T[16]);

/* O3D/Core/Registration/Registration.cpp:41-96 + KDTreeFlann.cpp:164-189.
 * Exact radius-limited 1-NN: accept iff d2 < (double)(float)(r*r) (strict).
 * idx[i] = target index or -1; d2[i] = squared distance (f64) or 0.
 * Returns K.  *err2 = sum of accepted d2.  Brute force, lowest index on ties. */
int64_t vo_nn_pass(const double *src, int64_t ns, const double *tgt, int64_t nt,
                   double max_dist, int32_t *idx, double *d2, double *err2);
/* Same result via a uniform grid with cell = max_dist (fast for big clouds). */
int64_t vo_nn_pass_grid(const double *src, int64_t ns, const double *tgt,
                        int64_t nt, double max_dist, int32_t *idx, double *d2,
                        double *err2);

/* O3D/Core/Geometry/PointCloud.cpp:122-142 (unbounded 1-NN distance). */
void vo_nn_distance(const double *src, int64_t ns, const double *tgt,
                    int64_t nt, double *dist);

/* src/constrained_ICP.cpp:13-23 == O3D TransformationEstimation.cpp:35-45.
 * corr = K pairs (src index, tgt index). */
double vo_compute_rmse(const double *src, const double *tgt,
                       const int32_t *corr, int64_t k);

/* src/constrained_ICP.cpp:25-37 -> Eigen/src/Geometry/Umeyama.h:93-162.
 * Two-pass demeaned form exactly as Umeyama.h.  Identity if k == 0. */
void vo_umeyama(const double *src, const double *tgt, const int32_t *corr,
                int64_t k, int with_scaling, double T[16]);

/* O3D/Core/Utility/Eigen.cpp:137-182 driven by the point-to-plane row of
 * O3D TransformationEstimation.cpp:82-92: r=(vs-vt).nt, J=[vs x nt | nt].
 * JTJ row-major 6x6, JTr 6, *r2 = sum r^2. */
void vo_jtj_jtr_point_to_plane(const double *src, const double *tgt,
                               const double *tgt_normals, const int32_t *corr,
                               int64_t k, double JTJ[36], double JTr[6],
                               double *r2);
/* Point-to-point written as three plane rows n = e_x, e_y, e_z per pair. */
void vo_jtj_jtr_point_to_point(const double *src, const double *tgt,
                               const int32_t *corr, int64_t k, double JTJ[36],
                               double JTr[6], double *r2);

/* O3D/Core/Utility/Eigen.cpp:35-56,58-68,88-106.  Returns 1 on success,
 * 0 (and Identity) if |det|<1e-6 / nan / inf. */
int vo_solve_jacobian_system(const double JTJ[36], const double JTr[6],
                             double T[16]);
void vo_vector6d_to_matrix4d(const double x[6], double T[16]);

/* O3D TransformationEstimation.cpp:75-103 (point-to-plane update). */
void vo_point_to_plane_update(const double *src, const double *tgt,
                              const double *tgt_normals, const int32_t *corr,
                              int64_t k, double T[16]);

typedef struct {
    double T[16];      /* transformation_ (row-major)                      */
    double fitness;    /* fitness_                                         */
    double rmse;       /* inlier_rmse_                                     */
    int64_t k;         /* correspondence_set_.size()                       */
    int32_t iters;     /* solves performed (<= max_iter)                   */
} vo_result;

enum { VO_EST_POINT_TO_POINT = 1, VO_EST_POINT_TO_PLANE = 2 };

/* O3D/Core/Registration/Registration.cpp:141-186.  idx_out (ns, may be NULL)
 * receives the final correspondence per source point (-1 = none).
 * trace (may be NULL) receives (max_iter+1) rows of 19 doubles:
 * [T(16), fitness, rmse, K] after NN pass #0..#iters.  use_grid selects the
 * NN implementation (identical results).  Returns 0, or -1 on bad arguments
 * (then out->T = init and the rest 0, as Registration.cpp:148-157). */
int vo_registration_icp(const double *src, int64_t ns, const double *tgt,
                        int64_t nt, const double *tgt_normals, double max_dist,
                        const double init[16], int estimator, int with_scaling,
                        double rel_fitness, double rel_rmse, int max_iter,
                        int use_grid, vo_result *out, int32_t *idx_out,
                        double *trace);

/* src/annotation.cpp:29-64: sweep `level` yaw initialisations about +Y, full
 * ICP from each, keep the first result with strictly most correspondences. */
int vo_register_model_to_scene(const double *model, int64_t ns,
                               const double *scene, int64_t nt,
                               const double *scene_normals, int level,
                               double max_dist, int point_to_plane,
                               double rel_fitness, double rel_rmse,
                               int max_iter, vo_result *best, int *best_level);

/* core/rodrigues.h (row-major vec(R) convention, as VISMA builds with
 * EIGEN_DEFAULT_TO_ROW_MAJOR): hat :8-15, vee :37-41, rodrigues :143-182,
 * invrodrigues :184-226.  dR_dw is 9x3 row-major, dw_dR 3x9 (may be NULL). */
void vo_hat(const double u[3], double M[9]);
void vo_vee(const double R[9], double v[3]);
void vo_rodrigues(const double w[3], double R[9], double *dR_dw);
void vo_invrodrigues(const double R[9], double w[3], double *dw_dR);
/* core/se3.h:96-110: compose, act, inverse on (R row-major 3x3, t). */
void vo_se3_compose(const double Ra[9], const double ta[3], const double Rb[9],
                    const double tb[3], double R[9], double t[3]);
void vo_se3_act(const double R[9], const double t[3], const double v[3],
                double out[3]);
void vo_se3_inv(const double R[9], const double t[3], double Ri[9],
                double ti[3]);

/* O3D/Core/Geometry/DownSample.cpp:179-220 (VoxelDownSample): voxel index
 * floor((p - (min_bound - voxel/2)) / voxel) in f64, per-voxel mean of points /
 * colours, normalised sum of the non-NaN normals, accumulation in input order.
 * The reference emits voxels in std::unordered_map iteration order; here they
 * come sorted by (ix, iy, iz).  normals / colors may be NULL.  Returns the number
 * of voxels (0 for voxel_size <= 0 or voxel_size * INT_MAX < extent). Outputs
 * must hold n rows. */
int64_t vo_voxel_down_sample(const double *xyz, const double *normals, const double *colors,
                             int64_t n, double voxel_size, double *out_xyz,
                             double *out_normals, double *out_colors);

/* feh::SamplePointCloudFromMesh (include/geometry.h:29-64).  The reference draws
 * from a TIME-seeded std::knuth_b, so only its mapping (uniforms -> points) can be
 * restated: sample i uses (r, a, b) = u[3i..3i+2].  quirks != 0 reproduces the
 * reference exactly -- face k is chosen for r in [cdf[k], cdf[k+1]) (geometry.h:53-54:
 * one face late, the last face never, r < cdf[0] yields NO point) and the point is
 * v0 + a(v1-v0) + b(v2-v0) with a,b in [0,1) (a parallelogram, :55-57).
 * quirks == 0: face k for r in [cdf[k-1], cdf[k]) and (a,b) reflected into the
 * triangle.  F = nf x 3 int32.  Returns the number of points written (<= n). */
int64_t vo_sample_mesh(const double *V, const int32_t *F, int64_t nf, const double *u,
                       int64_t n, int quirks, double *out_xyz);

/* Closest point on a triangle (Ericson, Real-Time Collision Detection 5.1.5 -- the
 * algorithm behind igl::point_simplex_squared_distance used by
 * feh::MeasureSurfaceError, include/geometry.h:117-141). */
double vo_point_triangle_sqdist(const double p[3], const double a[3], const double b[3],
                                const double c[3], double closest[3]);
/* For every query point the squared distance to the mesh, the face and the closest
 * point (brute force over all faces; lowest face index on exact ties). */
void vo_point_mesh_sqdist(const double *P, int64_t np, const double *V, const int32_t *F,
                          int64_t nf, double *d2, int32_t *face, double *closest);
/* feh::ComputeErrorMetric (include/geometry.h:85-101): out = mean, std, median
 * (errors[n >> 1] of the sorted list), min, max. */
void vo_error_metric(const double *errors, int64_t n, double out[5]);
void vo_estimate_normals(const double *xyz, int64_t n, const double *normals_in, int search_type,
                         int knn, double radius, double *out);

/* 3x3 SVD helper (one-sided Jacobi), exposed for tests. A = U diag(s) V^T,
 * s descending, row-major. */
void vo_svd3(const double A[9], double U[9], double s[3], double V[9]);

/* ---- (B) kernel specification, fp32 search + f64 statistics ------------ */

#define VK_NSTATS 38
/* Layout of the 38 statistics (the C ABI's visma_icp_reduce uses the same):
 *  [0]      K
 *  [1]      sum |p-q|^2
 *  [2..22]  upper triangle of JTJ (6x6, row by row: 00 01 .. 05 11 12 .. 55)
 *  [23..28] JTr
 *  [29..37] sum q p^T (3x3 row-major: q is the row index)
 * with rows J = [p x e_k | e_k], r_k = (p-q).e_k, k = x,y,z (Open3D order
 * x=[alpha beta gamma tx ty tz]). p = T*s evaluated in f64 from the fp32
 * source, q = fp32 target widened. */

/* Fused transform + brute-force NN exactly as the HIP kernel must do it:
 *   p = fmaf(r0,sx, fmaf(r1,sy, fmaf(r2,sz, t)))   per row, fp32
 *   d2 = fmaf(dz,dz, fmaf(dy,dy, dx*dx)),  d = q - p, fp32
 *   accept iff d2 < r2f, r2f = (float)(max_dist*max_dist); lowest index wins.
 * src/tgt are fp32 arrays with `stride` floats per point (3 or 4).
 * T32 = 12 floats (row-major 3x4).  idx = -1 if none, d2 then undefined (=r2f). */
int64_t vk_nn_pass_f32(const float *src, int64_t ns, int sstride,
                       const float *tgt, int64_t nt, int tstride,
                       const float T32[12], float r2f, int32_t *idx, float *d2);
int64_t vk_nn_pass_f32_grid(const float *src, int64_t ns, int sstride,
                            const float *tgt, int64_t nt, int tstride,
                            const float T32[12], float r2f, int32_t *idx,
                            float *d2);

/* Statistics over the accepted pairs (idx >= 0). T64 = row-major 3x4 f64.
 * offset (3 doubles, may be NULL = 0) shifts the frame: p+offset, q+offset. */
void vk_reduce_stats(const float *src, int64_t ns, int sstride,
                     const float *tgt, int tstride, const int32_t *idx,
                     const double T64[12], const double *offset,
                     double stats[VK_NSTATS]);

/* Closed-form Kabsch/Umeyama update from the statistics (design rule R1). */
void vk_solve_kabsch_from_stats(const double stats[VK_NSTATS], int with_scaling,
                                double T[16]);
/* Single Gauss-Newton step from the statistics (Open3D Euler update). */
int vk_solve_gn_from_stats(const double stats[VK_NSTATS], double T[16]);

/* The full GPU-shaped loop on the CPU: centre on the target centroid in f64,
 * round to fp32, apply TOTAL T to the pristine source each pass, compose in
 * f64 (design rule R2).  Same result layout as vo_registration_icp. */
int vk_registration_icp(const double *src, int64_t ns, const double *tgt,
                        int64_t nt, double max_dist, const double init[16],
                        int with_scaling, double rel_fitness, double rel_rmse,
                        int max_iter, int use_grid, vo_result *out,
                        int32_t *idx_out, double *trace);

/* Threads the OpenMP build will use (1 if built without OpenMP). */
int vo_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
