/*
 * visma_icp.h -- C ABI of the MI355X-native ICP registration path.
 *
 * This is the drop-in boundary under the C++ shim (include/constrained_ICP.h,
 * include/Core/Registration/Registration.h).  The reference has no C ABI or
 * FFI for this path (it is a C++ virtual plugin called from a C++ free
 * function), so each entry point cites the reference C++ interface it
 * replaces.  Paths are relative to the reference root;
 * O3D = thirdparty/Open3D/src.
 *
 * Conventions: every function returns a visma_icp_status (0 = OK) and never
 * throws; 4x4 matrices are ROW-MAJOR double[16]; point arrays are AoS with a
 * caller-given stride in elements; the caller owns every pointer it passes
 * and the library copies what it keeps.  One ctx per host thread per GPU
 * (thread-compatible, not thread-safe).
 */
#ifndef VISMA_ICP_H
#define VISMA_ICP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define VISMA_ICP_API __attribute__((visibility("default")))
#else
#define VISMA_ICP_API
#endif

typedef struct visma_icp_ctx visma_icp_ctx;

typedef enum {
    VISMA_ICP_OK = 0,
    VISMA_ICP_ERR_INVALID = 1,   /* bad argument                              */
    VISMA_ICP_ERR_NO_DEVICE = 2, /* no usable gfx950 device / HIP unavailable */
    VISMA_ICP_ERR_HIP = 3,       /* a HIP call failed (see last_error)        */
    VISMA_ICP_ERR_RCCL = 4,      /* RCCL unavailable or a collective failed   */
    VISMA_ICP_ERR_STATE = 5,     /* call order violated (e.g. no clouds set)  */
    VISMA_ICP_ERR_ENGINE = 6     /* an injected engine callback failed        */
} visma_icp_status;

/* Per-iteration solve.  KABSCH is the reference's arithmetic
 * (src/constrained_ICP.cpp:25-37 -> Eigen umeyama); the GN modes are single
 * Gauss-Newton steps on the 6x6 normal equations
 * (O3D/Core/Utility/Eigen.cpp:88-106; exp-map update via core/rodrigues.h:143). */
typedef enum {
    VISMA_ICP_SOLVER_KABSCH = 0,
    VISMA_ICP_SOLVER_GN_EULER = 1,
    VISMA_ICP_SOLVER_GN_EXPMAP = 2
} visma_icp_solver;

/* Nearest-neighbour search implementation (identical results at equal search
 * precision, see visma_icp_set_search_precision). */
typedef enum {
    VISMA_ICP_NN_AUTO = 0,
    VISMA_ICP_NN_BRUTE = 1, /* brute force over the whole target           */
    VISMA_ICP_NN_GRID = 2   /* radius-cell uniform grid (exact, radius-limited) */
} visma_icp_nn_mode;

#define VISMA_ICP_NSTATS 38
/* Layout of the reduced statistics (one ICP iteration's normal equations):
 *   [0]      K            number of correspondences
 *   [1]      sum |p-q|^2
 *   [2..22]  upper triangle of J^T J (6x6), row by row
 *   [23..28] J^T r
 *   [29..37] sum q p^T   (3x3 row-major, q = row)
 * rows J = [p x e_k | e_k], r_k = (p-q).e_k (k = x,y,z), parameter order
 * x = [alpha beta gamma tx ty tz] -- the convention of
 * O3D/Core/Registration/TransformationEstimation.cpp:82-92 and
 * O3D/Core/Utility/Eigen.cpp:137-182 (ComputeJTJandJTr). */

/* Mirrors open3d::RegistrationResult (O3D/Core/Registration/Registration.h:81-94)
 * plus bookkeeping. */
typedef struct {
    double transformation[16]; /* transformation_  (row-major)              */
    double fitness;            /* fitness_                                  */
    double inlier_rmse;        /* inlier_rmse_                              */
    int64_t num_correspondences; /* correspondence_set_.size()              */
    int32_t iterations;        /* solves performed                          */
    int32_t nn_passes;         /* NN passes performed (= iterations + 1)    */
} visma_icp_result;

/* Kernel timing accumulated since the last reset (HIP events on the ctx's
 * stream; only collected while profiling is enabled). */
typedef struct {
    double nn_ms;        /* total time in the NN-correspondence kernel       */
    int64_t nn_launches;
    double reduce_ms;    /* total time in the Jacobian/residual reduction    */
    int64_t reduce_launches;
    double aux_ms;       /* grid build / refine / other kernels              */
    int64_t aux_launches;
    double grid_candidates; /* target points examined by the grid search (sum) */
    double grid_candidates_27cell; /* ... points in the full 3x3x3 cell blocks (before row pruning) */
    /* streamed (LDS-tile) search, profiled launches only: */
    double tile_workgroups;          /* workgroups of the streamed search */
    double tile_fallback_workgroups; /* parts (see tile_parts) searched from global memory: footprint larger than the tile */
    double tile_parts;               /* parts the chunks were searched in (1 per workgroup unless a footprint had to be split) */
    double tile_points;              /* target points streamed into LDS (sum over workgroups) */
    double tile_rows;                /* (y,z) rows of the footprints (sum over workgroups) */
    double f64_reranks;              /* queries re-ranked in f64 (runner-up or radius inside the rounding band) */
    double tile_phase_cycles[7];     /* shader cycles per workgroup phase, summed over workgroups: source + transform,
                                        run bounds + footprint rows + scan, tile streaming, search + re-rank,
                                        winner fetch, moments + row; [6] unused */
    double grid_certified;           /* queries of profiled passes whose previous winner was CERTIFIED unchanged (no search:
                                        the warm-started kernel's triangle-inequality test, visma_amd/csrc/grid_coop.hip) */
    /* persistent launches (one launch of the certificate kernel running several passes of a host loop, the next
     * transform handed over through mapped host memory): nn_ms holds their whole duration -- the time the launch
     * waits for the host included -- and nn_launches counts their PASSES, so that nn_ms / nn_launches stays the
     * time of one pass whichever way it ran; these two say how many launches and passes that were. */
    double persist_launches;
    double persist_passes;
    double persist_ms;               /* their share of nn_ms */
    double persist_aborts;           /* persistent launches that ended by themselves (no command in time); counted always */
} visma_icp_timing;

/* ---- lifetime ---------------------------------------------------------- */

/* Create a context on HIP device `device`.  Fails with NO_DEVICE (never falls
 * back to the CPU) when there is no GPU.  Replaces the per-call state of
 * open3d::RegistrationICP (KD-tree + source copy, Registration.cpp:159-165). */
VISMA_ICP_API int visma_icp_create(visma_icp_ctx **out, int device);
VISMA_ICP_API int visma_icp_destroy(visma_icp_ctx *ctx);
/* Message of the last failure on ctx (or of the last failed create if NULL). */
VISMA_ICP_API const char *visma_icp_last_error(const visma_icp_ctx *ctx);
VISMA_ICP_API const char *visma_icp_version(void);
/* Number of HIP devices this process sees (0 without a GPU or a driver); no context needed. */
VISMA_ICP_API int visma_icp_device_count(void);

/* ---- clouds ------------------------------------------------------------ */

/* Upload both clouds from the reference's own storage: AoS float64 xyz
 * (open3d::PointCloud::points_, O3D/Core/Geometry/PointCloud.h:86; stride 3).
 * Centres both on the target centroid in f64, rounds to fp32 (design rule R2),
 * and remembers the centre so every transform in this API stays in the
 * caller's frame.  Replaces KDTreeFlann::SetGeometry + `PointCloud pcd = source`
 * (Registration.cpp:160-162). */
VISMA_ICP_API int visma_icp_set_clouds_f64(visma_icp_ctx *ctx, const double *src_xyz,
                                           int64_t ns, int src_stride,
                                           const double *tgt_xyz, int64_t nt,
                                           int tgt_stride);
/* The same with target = open3d::VoxelDownSample(scene, voxel_size) (O3D/Core/Geometry/DownSample.cpp:179-220):
 * the step both callers run on the scene right before RegistrationICP (src/evaluation.cpp:258-271,
 * src/annotation.cpp:112), fused with the target upload.  The scene crosses PCIe once, is down-sampled on the
 * device (the reference's point values bit for bit; voxels in ascending voxel-index order instead of the
 * reference's hash-map iteration order) and becomes the target where it lies; *nt_out = its point count.
 * Target indices of the results refer to that order; visma_icp_get_voxel_target copies the down-sampled points
 * out (nt x 3 f64) for a caller that wants them too.  The registration that follows equals, bit for bit, the one
 * after visma_icp_voxel_down_sample + visma_icp_set_clouds_f64. */
VISMA_ICP_API int visma_icp_set_clouds_f64_voxel_target(visma_icp_ctx *ctx, const double *src_xyz, int64_t ns,
                                                        int src_stride, const double *scene_xyz, int64_t n_scene,
                                                        int scene_stride, double voxel_size, int64_t *nt_out);
VISMA_ICP_API int visma_icp_get_voxel_target(visma_icp_ctx *ctx, double *xyz_out, int64_t nt);
/* The max_correspondence_distance the NEXT registration on this context will use (RegistrationICP's third argument,
 * Registration.h:102-107), told before the clouds are uploaded: visma_icp_set_clouds_f64 then builds the search
 * structure on the GPU while the host is still staging the source (C4: 0.7 ms of a 3.7 ms registration hidden).
 * One-shot: the upload that uses the hint clears it.  Without a hint the radius of the context's previous
 * registration is assumed; 0 clears a pending hint.  A wrong value
 * only costs the wasted build: results never depend on it. */
VISMA_ICP_API int visma_icp_set_radius_hint(visma_icp_ctx *ctx, double max_correspondence_distance);
/* feh::ICPRefinement's clouds (src/evaluation.cpp:248-271) made on the device, source side too:
 *   for every model: SamplePointCloudFromMesh(V, F, samples) (include/geometry.h:29-64; visma_icp_sample_mesh's
 *   draws: mesh k uses the stream `seed + k`, reference_quirks as there), PointCloud::Transform(model_to_scene)
 *   (O3D/Core/Geometry/PointCloud.cpp:75-80; row-major 4x4, NULL = identity), `*scene_est += *model`;
 *   scene = VoxelDownSample(scene, voxel_size)  (voxel_size == 0: the scene as it is).
 * The sampled points never cross PCIe: they are sampled, moved, concatenated, ordered and widened where the search
 * reads them.  *ns_out / *nt_out = the sizes of the two clouds; source indices of the results refer to the
 * concatenation in mesh order, visma_icp_get_mesh_source copies it (ns x 3 doubles) to the host.  Same values as
 * visma_icp_sample_mesh + a host transform + visma_icp_set_clouds_f64[_voxel_target] (tests/test_mesh_source.py). */
typedef struct {
    const double *V; int64_t nv;        /* vertices, nv x 3 */
    const int32_t *F; int64_t nf;       /* faces, nf x 3 */
    int64_t samples;                    /* options["samples_per_model"] */
    const double *model_to_scene;       /* 16 doubles, row-major, or NULL */
} visma_icp_mesh_source;
VISMA_ICP_API int visma_icp_set_clouds_meshes_f64(visma_icp_ctx *ctx, const visma_icp_mesh_source *meshes, int n_meshes,
                                                  int reference_quirks, uint64_t seed, const double *scene_xyz,
                                                  int64_t n_scene, int scene_stride, double voxel_size,
                                                  int64_t *ns_out, int64_t *nt_out);
VISMA_ICP_API int visma_icp_get_mesh_source(visma_icp_ctx *ctx, double *xyz_out, int64_t ns);
/* fp32 uploads, no centring (the caller's coordinates are used as they are; the exact search
 * takes the fp32 values as its f64 coordinates).  FRAME RULE: visma_icp_set_clouds_f64 centres
 * BOTH clouds on one point; these setters (and the _device ones) upload in the caller's frame.
 * The two frames cannot be combined: the first uncentred upload after a centred one INVALIDATES
 * the other cloud -- set it again (the next run fails with VISMA_ICP_ERR_STATE "clouds not set"
 * otherwise). */
VISMA_ICP_API int visma_icp_set_target(visma_icp_ctx *ctx, const float *xyz, int64_t nt,
                                       int stride_floats);
VISMA_ICP_API int visma_icp_set_source(visma_icp_ctx *ctx, const float *xyz, int64_t ns,
                                       int stride_floats);
/* Same from DEVICE memory (float4 xyzw per point, w ignored), copied D2D. */
VISMA_ICP_API int visma_icp_set_target_device(visma_icp_ctx *ctx, const void *d_xyzw,
                                              int64_t nt);
VISMA_ICP_API int visma_icp_set_source_device(visma_icp_ctx *ctx, const void *d_xyzw,
                                              int64_t ns);
/* Target normals (AoS f64, stride 3) for the point-to-plane estimator. */
VISMA_ICP_API int visma_icp_set_target_normals_f64(visma_icp_ctx *ctx,
                                                   const double *nxyz, int64_t nt,
                                                   int stride);

/* ---- the three kernels, individually ----------------------------------- */

/* Fused PointCloud::Transform (O3D/Core/Geometry/PointCloud.cpp:75-80) +
 * GetRegistrationResultAndCorrespondences (Registration.cpp:41-96): apply T
 * (caller frame) to the pristine source and find, per source point, the
 * nearest target point with d^2 < (float)(max_dist^2).  Results stay on the
 * device. */
VISMA_ICP_API int visma_icp_nn_pass(visma_icp_ctx *ctx, const double T[16],
                                    double max_dist);
/* Per-correspondence Jacobian/residual + wavefront-shuffle reduction to the
 * statistics above for the last nn_pass (replaces ComputeJTJandJTr,
 * O3D/Core/Utility/Eigen.cpp:137-182, and the gathers of
 * src/constrained_ICP.cpp:30-35).  Statistics are in the CENTRED frame. */
VISMA_ICP_API int visma_icp_reduce(visma_icp_ctx *ctx,
                                   double out_stats[VISMA_ICP_NSTATS]);
/* Correspondences of the last nn_pass, sorted by source index
 * (RegistrationResult::correspondence_set_).  Buffers hold >= ns entries;
 * d2 may be NULL. */
VISMA_ICP_API int visma_icp_get_correspondences(visma_icp_ctx *ctx, int32_t *src_idx,
                                                int32_t *tgt_idx, float *d2,
                                                int64_t *k);

/* ---- host solves (pure functions; no ctx, no GPU) ----------------------- */

/* Update from the statistics.  KABSCH: closed-form Umeyama
 * (Eigen/src/Geometry/Umeyama.h:118-159) from the moments; GN_*: solve
 * J^T J x = -J^T r with the |det| < 1e-6 guard (Eigen.cpp:35-56) and map x to
 * SE(3) by Rz*Ry*Rx (Eigen.cpp:58-68) or by the exponential map
 * (core/rodrigues.h:143-182).  Identity when K = 0 or the solve is rejected
 * (src/constrained_ICP.cpp:29; TransformationEstimation.cpp:102). */
VISMA_ICP_API int visma_icp_solve_from_stats(const double stats[VISMA_ICP_NSTATS],
                                             int solver, int with_scaling,
                                             double T_update[16]);

/* ---- the full loop ------------------------------------------------------ */

/* open3d::RegistrationICP (O3D/Core/Registration/Registration.h:102-107,
 * .cpp:141-186) with estimator = cicp::TransformationEstimationPointToPoint4DoF
 * (include/constrained_ICP.h:14-30).  Same iteration/termination semantics:
 * max_iter+1 NN passes at most, stop when |dfitness| < rel_fitness and
 * |drmse| < rel_rmse, result fields from the last NN pass.  max_dist <= 0
 * returns OK with transformation = init and everything else 0
 * (Registration.cpp:148-151). */
VISMA_ICP_API int visma_icp_run(visma_icp_ctx *ctx, const double init[16],
                                double max_dist, int max_iter, double rel_fitness,
                                double rel_rmse, int solver, int with_scaling,
                                visma_icp_result *out);
/* Exactly `steps` fixed ICP iterations with no stop test: each step = one NN
 * pass at T, one reduction, one solve, T <- update * T (the body of the loop
 * at Registration.cpp:169-178).  T_inout is updated in place; out (may be NULL)
 * reports fitness / rmse / K of the LAST pass (taken at the T before the
 * final update).  This is the unit bench.py times. */
VISMA_ICP_API int visma_icp_iterate(visma_icp_ctx *ctx, double T_inout[16], double max_dist,
                                    int steps, int solver, int with_scaling,
                                    visma_icp_result *out);
/* Point-to-plane estimator (TransformationEstimation.cpp:61-103) on the same
 * reduction; needs set_target_normals_f64, else returns OK with
 * transformation = init (Registration.cpp:152-157). */
VISMA_ICP_API int visma_icp_run_point_to_plane(visma_icp_ctx *ctx, const double init[16],
                                               double max_dist, int max_iter,
                                               double rel_fitness, double rel_rmse,
                                               visma_icp_result *out);

/* feh::RegisterModelToScene (include/tool.h:40-42, src/annotation.cpp:29-64):
 * `level` yaw initialisations R_y(2 pi i / level), a full ICP from each (all
 * in flight together on the GPU), keep the first with strictly the most
 * correspondences.  per_level (may be NULL) receives all `level` results. */
VISMA_ICP_API int visma_icp_run_yaw_sweep(visma_icp_ctx *ctx, int level, double max_dist,
                                          int max_iter, double rel_fitness,
                                          double rel_rmse, int solver,
                                          visma_icp_result *best, int *best_level,
                                          visma_icp_result *per_level);

/* ---- batched small problems (AnnotationTool loop, src/annotation.cpp:103-168) */

typedef struct {
    const double *src_xyz; int64_t ns; /* AoS f64, stride 3 */
    const double *tgt_xyz; int64_t nt;
    double init[16];
    double max_dist;
} visma_icp_problem;

/* The same sweep with the point-to-plane estimator (ICP.point_to_plane of cfg/tool.json,
 * src/annotation.cpp:45-50): needs target normals (visma_icp_set_target_normals_f64); without
 * them every start returns its initial transform, like Registration.cpp:152-157. */
VISMA_ICP_API int visma_icp_run_yaw_sweep_point_to_plane(visma_icp_ctx *ctx, int level,
                                                         double max_dist, int max_iter,
                                                         double rel_fitness, double rel_rmse,
                                                         visma_icp_result *best, int *best_level,
                                                         visma_icp_result *per_level);
/* n independent ICPs advanced in lock step, one grid launch per iteration. */
VISMA_ICP_API int visma_icp_run_batch(visma_icp_ctx *ctx, const visma_icp_problem *probs,
                                      int n, int max_iter, double rel_fitness,
                                      double rel_rmse, int solver,
                                      visma_icp_result *out);
/* The same batch over several WORKER contexts (any number on the same GPU -- each has its own stream -- and/or on
 * several GPUs): problems that pass the same target cloud stay together, the groups are dealt to the contexts by
 * size, every context runs its share as one visma_icp_run_batch on its own host thread, the shares side by side:
 * while one worker packs, uploads or solves, the others' searches have the GPU (config 3 on one MI355X: two workers
 * ~1.15x one).  out[] equals the single-context call's.  Returns the first error (message in errbuf). */
VISMA_ICP_API int visma_icp_run_batch_multi(visma_icp_ctx *const *ctxs, int n_ctx, const visma_icp_problem *probs, int n,
                                            int max_iter, double rel_fitness, double rel_rmse, int solver,
                                            visma_icp_result *out, char *errbuf, size_t errbuf_len);
/* The same with the point-to-plane estimator (TransformationEstimationPointToPlane,
 * O3D/Core/Registration/TransformationEstimation.cpp:101-134): tgt_normals[i] are the normals of
 * probs[i].tgt_xyz (AoS f64, stride 3; the same pointer wherever the same target pointer is
 * passed).  A problem whose target has no normals (NULL) returns its initial transform, like
 * Registration.cpp:152-157. */
VISMA_ICP_API int visma_icp_run_batch_point_to_plane(visma_icp_ctx *ctx, const visma_icp_problem *probs,
                                                     const double *const *tgt_normals, int n,
                                                     int max_iter, double rel_fitness, double rel_rmse,
                                                     visma_icp_result *out);

/* ---- the corpus: every (scene, CAD candidate) pair, one orientation-constrained registration each --------
 * The per-object loop of AnnotationTool (src/annotation.cpp:103-168: for each entry of objects.json ->
 * RegisterModelToScene(model, scan, config["ICP"])) over a whole corpus, as a native work queue: ONE HOST
 * THREAD PER CONTEXT (= per GPU) pulls chunks of `chunk` items from a counter, runs their `level` yaw starts
 * (src/annotation.cpp:35-39) as one batch (visma_icp_run_batch: clouds passed by several problems are uploaded
 * and gridded once) and keeps, per item, the first start with strictly the most correspondences
 * (src/annotation.cpp:59-61).  No collective, no exchange: replicas only.
 * `counter` (may be NULL: a private one) is advanced with atomic adds: ranks in OTHER processes can pull from
 * the same queue when it lives in shared memory (bench.py --workload c5 --gpus N: one process per GPU); it must
 * be 0 when the pass starts.  results[i].device is the index of the context that registered item i, -1 when
 * another process took it.  Returns the first error of any thread (message in errbuf). */
typedef struct {
    const double *model_xyz; int64_t n_model;   /* source: the sampled CAD model (AoS f64, stride 3) */
    const double *scene_xyz; int64_t n_scene;   /* target: the scan */
} visma_icp_corpus_item;
typedef struct {
    int level;               /* rotation_level (cfg/tool.json:17): yaw starts R_y(2 pi k / level) */
    double max_dist;         /* distance_threshold */
    int max_iter;            /* ICPConvergenceCriteria: 30 */
    double rel_fitness, rel_rmse;
    int solver;              /* VISMA_ICP_SOLVER_KABSCH = the reference's estimator */
    int chunk;               /* items per pull (<= 0: 8 -> 8 x 24 = 192 registrations in flight per launch) */
} visma_icp_corpus_params;
typedef struct {
    visma_icp_result best;   /* RegisterModelToScene's choice (transformation_ = what it returns) */
    int32_t best_level;      /* which start it was; -1: no start found a correspondence (best = identity) */
    int32_t device;          /* index into ctxs of the context that did it; -1: not done by this call */
    int64_t iterations_all_starts;   /* ICP iterations summed over the item's `level` registrations */
} visma_icp_corpus_result;
VISMA_ICP_API int visma_icp_run_corpus(visma_icp_ctx *const *ctxs, int n_ctx, const visma_icp_corpus_item *items,
                                       int64_t n_items, const visma_icp_corpus_params *params, int64_t *counter,
                                       visma_icp_corpus_result *results, char *errbuf, size_t errbuf_len);

/* ---- the gravity alignment and pose composition of feh::AnnotationTool (src/annotation.cpp:82-91, 111-153) --------
 * Host arithmetic (no context): the steps between the scan / the CAD model and RegisterModelToScene.
 *   visma_geom_find_plane_normal       feh::FindPlaneNormal (include/geometry.h:18-26): unit normal of the plane through
 *                                      n points = right singular vector of the smallest singular value of their
 *                                      covariance, WITH THE SIGN Eigen 3.3.2's JacobiSVD gives it (the reference turns
 *                                      this normal onto +Y: the sign decides which way up the scene ends)
 *   visma_geom_jacobi_svd3             Eigen::JacobiSVD<Matrix3d>(A, ComputeFullU | ComputeFullV), row-major 3x3
 *   visma_geom_rotation_between_vectors  feh::RotationBetweenVectors (core/utils.h:229-233) =
 *                                      Eigen::Quaterniond::FromTwoVectors(u, v).toRotationMatrix(), row-major 3x3
 *   visma_geom_centre_on_floor         (-mean_x, -min_y, -mean_z) of a cloud: the translation of T1 / T2
 *                                      (src/annotation.cpp:114-119, 128-132)
 *   visma_annot_total_pose             Ttot = (T1 T0)^-1 T3 T2 with the reference's rigid inverse (:147-153), row-major 4x4 */
VISMA_ICP_API int visma_geom_find_plane_normal(const double *xyz, int64_t n, double normal_out[3]);
VISMA_ICP_API int visma_geom_jacobi_svd3(const double A[9], double U[9], double S[3], double V[9]);
VISMA_ICP_API int visma_geom_rotation_between_vectors(const double u[3], const double v[3], double R[9]);
VISMA_ICP_API int visma_geom_centre_on_floor(const double *xyz, int64_t n, double t_out[3]);
VISMA_ICP_API int visma_annot_total_pose(const double T0[16], const double T1[16], const double T2[16], const double T3[16],
                                         double Ttot[16]);

/* ---- options / measurement --------------------------------------------- */
/* AUTO (default) uses the radius-cell grid whenever the target/radius make it
 * worthwhile and the LDS-tiled brute-force kernel otherwise; BRUTE / GRID force
 * one.  The grid is (re)built on the GPU when the target or the radius changes. */
VISMA_ICP_API int visma_icp_set_nn_mode(visma_icp_ctx *ctx, int nn_mode);
/* Arithmetic of the nearest-neighbour search.
 *   0            fp32 only: fp32 distances on the fp32-rounded clouds centred on the target
 *                centroid (the round-1 kernels).  Near-ties between two candidates, and
 *                candidates within ~1e-6 of the radius, can be decided differently from the
 *                reference's f64 KD-tree (about one query in 1e5); one flipped pair among K
 *                moves the update by ~(pair spacing)/K.
 *   1 (default)  exact: candidates are ranked in fp32, the best two are kept, and whenever
 *                the runner-up or the radius lies within the rounding band of the best
 *                (2.4e-7 (|p|_1 + r) + 4.8e-7 r on the distance, both operands fp32-rounded
 *                f64 coordinates) the candidates concerned are re-ranked in f64 with the
 *                reference's arithmetic: the f64 sum of squares of FLANN L2<double>, the
 *                strict d2 < (double)(float)(r*r) test, lowest index on exact ties.  Three
 *                candidates inside the band: the query rescans its cells in f64.  Source
 *                transform and statistics in f64 from the caller's f64 coordinates (shifted by
 *                the target centroid in f64 on upload).  The
 *                correspondences are those of mode 2 (and of the reference) for every input;
 *                every path has this flavour -- grid (single, sweep, batch), brute force,
 *                target-sharded.  Clouds given as fp32 are promoted on the device.
 *   2            f64 search: every candidate distance in f64.  Same results as 1, slower
 *                (+30 % at 64k -> 256k); kept as the in-library check of mode 1.
 * Takes effect at the next cloud upload.  Mode 1 runs at the speed of mode 0 (measured on MI355X,
 * profiles/r02_probe_keepq.txt: 56.3 vs 58.6 us per iteration at C4, 30.7 vs 29.0 at 64k -> 1M).
 * visma_icp_get_search_precision_used reports what the last run executed (0 / 1 / 2). */
VISMA_ICP_API int visma_icp_set_search_precision(visma_icp_ctx *ctx, int mode);
/* What the last pass ran: 0 fp32 ranking only, 1 exact (fp32 ranking + f64 re-rank), 2 f64. */
VISMA_ICP_API int visma_icp_get_search_precision_used(visma_icp_ctx *ctx, int *is_f64);
/* Which search the last nn_pass used (VISMA_ICP_NN_BRUTE or VISMA_ICP_NN_GRID). */
VISMA_ICP_API int visma_icp_get_nn_mode_used(visma_icp_ctx *ctx, int *nn_mode);
/* Kernel timing with HIP events on the context's stream (read with
 * visma_icp_get_timing).  0 = off, 1 = every launch, n > 1 = every n-th ICP pass
 * (the event records themselves cost ~3 us each; sampling keeps a timed run close
 * to an untimed one).  Candidate counting of the grid kernel follows the same switch. */
VISMA_ICP_API int visma_icp_set_profiling(visma_icp_ctx *ctx, int enabled);
VISMA_ICP_API int visma_icp_get_timing(visma_icp_ctx *ctx, visma_icp_timing *out,
                                       int reset);
/* The same for callers compiled against another revision of this header: at most `struct_size` bytes of the
 * structure are written (the structure only ever grows at its end; round 4 added persist_* -- a caller built against
 * round 3's header passes ITS sizeof and is not overrun).  visma_icp_get_timing(ctx, out, reset) is
 * visma_icp_get_timing_sized(ctx, out, sizeof(visma_icp_timing) OF THE LIBRARY'S BUILD, reset). */
VISMA_ICP_API int visma_icp_get_timing_sized(visma_icp_ctx *ctx, void *out, size_t struct_size, int reset);

/* ---- the persistent launch of a host loop -------------------------------------------------------------------------
 * visma_icp_run / visma_icp_iterate on one GPU (and source-sharded ranks on their own GPUs) keep ONE launch of the
 * search kernel alive for the passes of their loop: the next transform goes to the launch through a command block, the
 * statistics come back as always -- same results as one launch per pass, bit for bit.  Such a launch
 * needs every one of its workgroups resident at once and SPINS between passes: while a loop runs, the compute units it
 * holds are not available to other streams or processes.  An integrator decides with these three calls (they replace
 * nothing in the reference, which has no device to share):
 *  visma_icp_set_persistent(ctx, enabled, timeout_ms): 1 (default) / 0 = one launch per pass on this context (also
 *    VISMA_ICP_PERSIST=0).  timeout_ms > 0: how long the launch waits for the host's next command before it ends by
 *    itself (default 200; the loop then carries on with ordinary launches, same results).  After such an abort the
 *    context COOLS DOWN: its next 8 host loops launch once per pass, then persistent launches are tried again by
 *    themselves; visma_icp_set_persistent(ctx, 1, ...) re-arms them at once.
 *  visma_icp_set_persistent_cu_share(share): PER PROCESS, 0 < share <= 1 (default 1; also VISMA_ICP_PERSIST_CU_SHARE):
 *    the largest part of a device's workgroup slots a persistent launch may hold.  A loop whose launch would need
 *    more runs one launch per pass, which other streams' kernels interleave with (a 262,144-point source needs all
 *    slots of an MI355X; 65,536 points a quarter).  Queue workers of visma_icp_run_corpus / batches never start
 *    persistent launches.
 *  visma_icp_get_persistent_info(ctx, out): what happened so far on this context. */
typedef struct {
    int struct_size;               /* in: sizeof(visma_icp_persistent_info) of the caller's build */
    int enabled;                   /* the next host loop may start a persistent launch: the user's setting AND not cooling
                                      down (0 during the 8 loops after an abort, although the setting is still 1) */
    int last_loop_persistent;      /* the last finished host loop ran (part of) its passes in a persistent launch */
    int last_loop_passes;          /* ... that many of them */
    double launches, passes;       /* persistent launches / passes inside them since the context was created */
    double aborts;                 /* launches that ended by themselves or whose host came back too late */
    double timeout_ms;             /* the patience in force */
    double cu_share;               /* the process-wide share in force */
    int device_slots;              /* workgroups of this kernel the device holds at once (0: not asked yet) */
    int reserved;
} visma_icp_persistent_info;
VISMA_ICP_API int visma_icp_set_persistent(visma_icp_ctx *ctx, int enabled, double timeout_ms);
VISMA_ICP_API int visma_icp_set_persistent_cu_share(double share);
VISMA_ICP_API int visma_icp_get_persistent_info(visma_icp_ctx *ctx, visma_icp_persistent_info *out);

/* ---- radii that are large against the target's point spacing --------------------------------------------------------
 * The grid search lists the 27 radius-sized cells around a query (KDTreeFlann.cpp:164-189 asks for the nearest point
 * within the radius).  When such a cell holds hundreds of points -- a radius of tens of point spacings -- the library builds
 * cells of a few point spacings instead and searches them in rings of rows around the query, nearest first, bounded by the
 * best candidate so far and, after the first pass, by the previous winner (grid_ring.hip): the same correspondences, bit
 * for bit; the cost of a query follows the number of points nearer than its nearest neighbour, not the radius.
 *  visma_icp_set_ring_search(ctx, mode): -1 (default) by the occupancy of the radius-sized cells (>= 48 points per
 *    occupied cell for yaw sweeps, whose far-off starts leave most queries without a partner -- the ring walk's worst
 *    case --, >= 20 for one registration at a time; VISMA_ICP_RING_OCCUPANCY sets the first), 0 never, 1 whenever the f64 views exist and a finer table fits (also
 *    VISMA_ICP_RING=0/1 when the context is created).  Takes effect at the next grid build (new target or radius).
 *  visma_icp_get_ring_search(ctx, ...): what the current grid is: *rings > 0 = ring search with that many rings at most,
 *    *cell = the cell edge, *occupancy = points per occupied radius-sized cell as counted (0 = not counted).  Any pointer may
 *    be NULL.  Replaces nothing in the reference (FLANN's KD-tree has no such regime change). */
VISMA_ICP_API int visma_icp_set_ring_search(visma_icp_ctx *ctx, int mode);
VISMA_ICP_API int visma_icp_get_ring_search(visma_icp_ctx *ctx, int *rings, double *cell, double *occupancy);

/* ---- multi-GPU (one process per GPU; source-sharded) -------------------- */

#define VISMA_ICP_UNIQUE_ID_BYTES 128
/* Rank 0 creates the id, the host program broadcasts it (any transport), every
 * rank calls comm_init.  After that visma_icp_reduce / visma_icp_run sum the
 * per-shard statistics with ONE ncclAllReduce(38 x f64) per iteration over
 * xGMI; every rank solves the same system and holds the same transform.
 * RCCL is loaded at run time (dlopen) -- a single-GPU build needs none. */
VISMA_ICP_API int visma_icp_comm_unique_id(void *out_id /* 128 bytes */);
VISMA_ICP_API int visma_icp_comm_init(visma_icp_ctx *ctx, int rank, int nranks,
                                      const void *unique_id);
/* Peer-to-peer alternative on one node (preferred: ~3 us per iteration against ~30 us for a
 * 304-byte ncclAllReduce).  Every rank exports the handle of its mailbox (uncached device
 * memory), the host program all-gathers the handles (any transport), every rank calls
 * comm_ipc_init with ALL of them (nranks x VISMA_ICP_IPC_HANDLE_BYTES, rank order).  From then on
 * each iteration's 38 statistics are exchanged inside the search launch (by the workgroup that
 * finishes the fold; one small extra launch on the brute-force path): remote 16-byte stores
 * {value, sequence tag} into the peers' mailboxes over xGMI, every rank sums in rank order
 * (identical transforms on all ranks, bit for bit), the result goes straight to the host.
 * nranks <= 16; ranks may share a device (tests).  comm_ipc_init is COLLECTIVE: it ends with a
 * handshake (one all-reduce of known values, every rank must enter within tens of seconds) and
 * fails with VISMA_ICP_ERR_HIP when a peer's stores do not arrive -- fall back to
 * visma_icp_comm_init on every rank then.  comm_ipc_export starts a NEW session: it drops the mappings
 * of an earlier one, clears the mailbox and restarts the exchange count (which lives in device memory and
 * advances only when an exchange runs), so a retry after a failed handshake begins with export on every
 * rank again. */
#define VISMA_ICP_IPC_HANDLE_BYTES 64
VISMA_ICP_API int visma_icp_comm_ipc_export(visma_icp_ctx *ctx, void *out_handle /* 64 bytes */);
VISMA_ICP_API int visma_icp_comm_ipc_init(visma_icp_ctx *ctx, int rank, int nranks,
                                          const void *all_handles);
/* Alternative to RCCL: the host supplies the all-reduce (used by the CPU
 * `gloo` tests).  fn must sum `n` doubles in place across ranks. */
typedef int (*visma_icp_allreduce_fn)(void *user, double *inout, int n);
VISMA_ICP_API int visma_icp_set_allreduce(visma_icp_ctx *ctx, visma_icp_allreduce_fn fn,
                                          void *user, int rank, int nranks);
/* TARGET-sharded ranks (the literal reading of "the target cloud is sharded
 * across the GPUs"; for targets that exceed one GPU).  Every rank holds ALL source
 * points and target points [global_offset, global_offset + nt) of a target of
 * global_nt points (< 2^31).  Summing per-shard accumulators directly would count a
 * source point once per shard that has a neighbour of it, which is not the reference
 * algorithm (Registration.cpp:53-85 keeps ONE nearest neighbour); the exact scheme is
 * two collectives per iteration: a MIN all-reduce of NS packed keys
 * (fp32 d2 bits << 32 | global index: smallest distance, lowest index on ties), then
 * the owner of each winner accumulates it and the 38 statistics are summed as in the
 * source-sharded mode.  Correspondence indices are global.  `centre` must be the SAME
 * point on every rank (e.g. the centroid of the whole target); call this BEFORE
 * visma_icp_set_clouds_f64.  global_nt = 0 switches the mode off.  Needs
 * visma_icp_comm_init (RCCL) or both host callbacks below; the host loop only. */
VISMA_ICP_API int visma_icp_set_target_shard(visma_icp_ctx *ctx, int64_t global_offset,
                                             int64_t global_nt, const double centre[3]);
/* Host-supplied MIN all-reduce of n uint64 keys, in place (tests / other transports). */
typedef int (*visma_icp_minreduce_fn)(void *user, uint64_t *inout, int64_t n);
VISMA_ICP_API int visma_icp_set_minreduce(visma_icp_ctx *ctx, visma_icp_minreduce_fn fn, void *user);
/* Total source points over all ranks (fitness denominator); 0 = local ns. */
VISMA_ICP_API int visma_icp_set_global_source_count(visma_icp_ctx *ctx, int64_t ns_total);

/* open3d::VoxelDownSample (O3D/Core/Geometry/DownSample.cpp:179-220), the step
 * both callers run right before ICP (src/annotation.cpp:112,
 * src/evaluation.cpp:258).  AoS f64 in (stride 3; normals / colors may be NULL),
 * AoS f64 out (buffers of n rows).  Every output value is bit-identical to the
 * reference's (same f64 voxel index expression, sums taken in input order); the
 * ORDER of the voxels is ascending (ix,iy,iz) here, hash-map order there.
 * voxel_size <= 0, or voxel_size * INT_MAX < extent, give *n_out = 0 like the
 * reference. */
VISMA_ICP_API int visma_icp_voxel_down_sample(visma_icp_ctx *ctx, const double *xyz, int64_t n,
                                              const double *normals, const double *colors,
                                              double voxel_size, double *out_xyz,
                                              double *out_normals, double *out_colors,
                                              int64_t *n_out);

/* feh::SamplePointCloudFromMesh (include/geometry.h:29-64), the step that builds
 * the ICP source from a CAD mesh (src/evaluation.cpp:252, src/annotation.cpp:126).
 * V: nv x 3 f64, F: nf x 3 int32.  Sample i uses three uniforms (r, a, b): from
 * `uniforms` (3n doubles) when given, else from a counter-based Philox4x32-10
 * keyed by `seed` (the reference seeds std::knuth_b from the clock, so only its
 * mapping uniforms -> points can be matched).  reference_quirks != 0 reproduces
 * that mapping exactly: face k for r in [cdf[k], cdf[k+1]) (one face late, never
 * the last face, NO point for r < cdf[0]) and v0 + a(v1-v0) + b(v2-v0) over the
 * whole parallelogram -- about half of those points lie off the surface.
 * reference_quirks == 0 samples the triangles themselves, area-uniformly.
 * out_xyz holds n rows; *n_out <= n. */
VISMA_ICP_API int visma_icp_sample_mesh(visma_icp_ctx *ctx, const double *V, int64_t nv,
                                        const int32_t *F, int64_t nf, int64_t n,
                                        int reference_quirks, uint64_t seed,
                                        const double *uniforms, double *out_xyz, int64_t *n_out);
/* open3d::EstimateNormals (O3D/Core/Geometry/EstimateNormals.cpp:114-153; PointCloud.h:140-146) on the
 * GPU: the normal of every point from the covariance of its neighbours, found by one of KDTreeFlann's
 * three searches (KDTreeFlann.cpp:114-189) --
 *   search_type 0  KDTreeSearchParamKNN(knn)            the knn nearest points (the point itself included)
 *               1  KDTreeSearchParamRadius(radius)      every point with d2 < (double)(float)(radius^2)
 *               2  KDTreeSearchParamHybrid(radius, knn) the knn nearest of those
 * -- fewer than 3 neighbours: (0,0,1); normals_in (may be NULL) are the cloud's existing normals, whose
 * sign is kept (EstimateNormals.cpp:133-146).  n x 3 f64 in, n x 3 f64 out.  Any knn /
 * max_nn: lists of up to 170 entries live in LDS, longer ones (and dense Radius searches) in a heap per point in
 * global memory; Radius results are summed in the order of the reference's result list too. */
VISMA_ICP_API int visma_icp_estimate_normals(visma_icp_ctx *ctx, const double *xyz, int64_t n,
                                             const double *normals_in, int search_type, int knn,
                                             double radius, double *normals_out);

/* Point -> triangle-mesh squared distance, face and closest point for np query
 * points: what igl::AABB::squared_distance returns inside feh::MeasureSurfaceError
 * (include/geometry.h:123-136).  face / closest may be NULL; exact ties go to the
 * lowest face index. */
VISMA_ICP_API int visma_icp_point_mesh_distance(visma_icp_ctx *ctx, const double *P, int64_t np,
                                                const double *V, int64_t nv, const int32_t *F,
                                                int64_t nf, double *d2, int32_t *face,
                                                double *closest);
/* feh::ComputeErrorMetric (include/geometry.h:85-101): out = mean, std, median
 * (sorted[n >> 1]), min, max.  Host only. */
VISMA_ICP_API int visma_icp_error_metric(const double *errors, int64_t n, double out[5]);
/* feh::MeasureSurfaceError (include/geometry.h:117-141): sample the source mesh,
 * distance of every sample to the target mesh, statistics of the distances. */
VISMA_ICP_API int visma_icp_measure_surface_error(visma_icp_ctx *ctx, const double *Vs, int64_t nvs,
                                                  const int32_t *Fs, int64_t nfs, const double *Vt,
                                                  int64_t nvt, const int32_t *Ft, int64_t nft,
                                                  int64_t num_samples, int reference_quirks,
                                                  uint64_t seed, double out[5]);


/* ---- SO(3) maps and their derivatives (host; the device versions are the same code) ----------
 * core/rodrigues.h of the reference on plain arrays.  3x3 matrices row-major; a derivative of
 * (or with respect to) a matrix indexes it by its ROW-MAJOR vectorisation, as the reference
 * (built with EIGEN_DEFAULT_TO_ROW_MAJOR) does:
 *   rodrigues      R = exp(hat(w)),  dR_dw[(3i+j)*3 + k] = dR(i,j)/dw(k)     (:143-182; th < 1e-8 -> I + hat(w))
 *   invrodrigues   w = log(R),       dw_dR[k*9 + 3i+j]   = dw(k)/dR(i,j)     (:184-226; tr -> 3 branch)
 *   project        U V^T of the SVD (projectSO3 :229-237, SO3Type::fitToSO3 core/se3.h:58-61)
 *   matrix_derivatives   dAB_dA, dAB_dB (:87-141), dAt_dA (:58-69), dhat (:17-35), dvee (:43-56);
 *                        any output may be NULL.
 * Jacobian arguments may be NULL. */
/* SE3Type of core/se3.h:79-169 as plain functions on g = [R | t], row-major 3x4: composition (:96-100),
 * action on a point (:103-106: what every search kernel applies to a source point), inverse (:108-110).
 * Host functions (visma_icp_testing.h: visma_icp_selftest_se3 runs the same code on the GPU).  Pinned by the outputs of
 * the reference header itself (tests/golden/se3.npz, oracle/ref_se3.cpp). */
VISMA_ICP_API int visma_se3_compose(const double a[12], const double b[12], double out[12]);
VISMA_ICP_API int visma_se3_act(const double g[12], const double v[3], double out[3]);
VISMA_ICP_API int visma_se3_inv(const double g[12], double out[12]);
VISMA_ICP_API int visma_so3_rodrigues(const double w[3], double R[9], double dR_dw[27]);
VISMA_ICP_API int visma_so3_invrodrigues(const double R[9], double w[3], double dw_dR[27]);
VISMA_ICP_API int visma_so3_project(const double A[9], double R[9]);
VISMA_ICP_API int visma_so3_matrix_derivatives(const double A[9], const double B[9],
                                               double dAB_dA[81], double dAB_dB[81],
                                               double dAt_dA[81], double dhat[27], double dvee[27]);

#ifdef __cplusplus
}
#endif
#endif /* VISMA_ICP_H */
