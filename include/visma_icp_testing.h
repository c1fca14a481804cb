/* visma_icp_testing.h -- (1) the TEST SEAM of the library's host driver and (2) the measurement / A-B knobs.  Neither
 * is part of the product ABI.  (1): libvisma_icp.so does not export visma_icp_create_with_engine; only the side build
 * (visma_amd/lib/libvisma_icp_experiments.so, -DVISMA_TEST_SEAMS, built by visma_amd.build.build_experiments)
 * does, and only tests load it. */
#ifndef VISMA_ICP_TESTING_H
#define VISMA_ICP_TESTING_H
#include "visma_icp.h"
#ifdef __cplusplus
extern "C" {
#endif

/* The driver (centring, loop, solve, stop test, sharding) runs over an
 * "engine" that owns the clouds and produces statistics.  visma_icp_create
 * installs the HIP engine, the only one the product ships.  This entry point
 * lets the CPU test-suite drive the same host logic with an engine of its
 * own; the product never calls it. */
typedef struct {
    int (*set_source)(void *user, const float *xyzw, int64_t ns);
    int (*set_target)(void *user, const float *xyzw, int64_t nt);
    int (*set_target_normals)(void *user, const float *nxyzw, int64_t nt);
    int (*nn_pass)(void *user, const double T_centred[16], double max_dist);
    /* statistics of p + frame_offset, q + frame_offset (0 = centred frame) */
    int (*reduce)(void *user, const double T_centred[16], const double frame_offset[3],
                  int point_to_plane, double stats[VISMA_ICP_NSTATS]);
    int (*get_correspondences)(void *user, int32_t *tgt_idx_per_src, float *d2);
} visma_icp_engine;
VISMA_ICP_API int visma_icp_create_with_engine(visma_icp_ctx **out,
                                               const visma_icp_engine *engine,
                                               void *user);

/* ---- measurement and A/B knobs ------------------------------------------------------------------------------------
 * Exported by libvisma_icp.so too (bench.py, tools/ and the tests reach them through ctypes), but NOT part of the
 * drop-in boundary: nothing of the reference corresponds to them and no caller of the path needs them (round 4: moved
 * here from visma_icp.h). */
/* Which kernel the last grid pass ran: 0 brute force, 1 the lane-serial grid search (first pass of a
 * registration: progressive pruning, nothing known about the queries), 2 the warm-started wave-cooperative
 * search (visma_amd/csrc/grid_coop.hip: every later pass; each query starts from its previous winner, which
 * bounds it before anything is gathered), 3 the ring search over cells smaller than the radius (grid_ring.hip: radii
 * that are large against the point spacing, visma_icp_set_ring_search).  Same results, bit for bit.
 * visma_icp_forget_winners drops what the passes so far remembered (the library does so itself whenever the
 * source, the target or the radius changes): the next pass then runs like the first of a new registration.
 * Replaces nothing in the reference (KDTreeFlann keeps no state between searches, KDTreeFlann.cpp:164-189). */
VISMA_ICP_API int visma_icp_get_search_kernel_used(visma_icp_ctx *ctx, int *kernel);
VISMA_ICP_API int visma_icp_forget_winners(visma_icp_ctx *ctx);

/* Where the ICP loop runs.  1: ON THE DEVICE (per-iteration solve, compose and
 * stop test in a one-thread kernel epilogue, the host reads the state back
 * every 8 passes); 0: on the host (statistics published to mapped host memory,
 * host spin-waits, solves in f64, relaunches); -1 (default): automatic -- host
 * loop for a single problem, device loop for visma_icp_run_yaw_sweep, whose
 * `level` problems then advance together with one set of launches per
 * iteration.  Results agree to rounding. */
VISMA_ICP_API int visma_icp_set_device_loop(visma_icp_ctx *ctx, int enabled);

/* (visma_icp_set_persistent, visma_icp_set_persistent_cu_share, visma_icp_get_persistent_info: product API since
 *  round 5, visma_icp.h.)
 * visma_icp_test_stall_command: (tests) the host sleeps `ms` before it posts its `nth` command to a persistent launch
 * from now (1 = the next) -- a stalled host thread, as seen from the device.  Replaces nothing in the reference. */
VISMA_ICP_API int visma_icp_test_stall_command(visma_icp_ctx *ctx, int nth, double ms);

/* (round 6, measurement) the persistent SWEEP launch of device-resident loops over shared clouds (yaw sweeps, the device
 * loop of one registration: every pass after the first inside ONE launch; VISMA_ICP_SWEEP_PERSIST=0 switches it off):
 * launches started / launches that gave up (a wait ran out; the loop carried on with one launch per pass) since the
 * context was created.  Either pointer may be NULL. */
VISMA_ICP_API int visma_icp_get_sweep_info(visma_icp_ctx *ctx, double *launches, double *aborts);

/* (round 6, host logic of the ring search -- visma_amd/csrc/grid_ring.hip --, no device needed)
 * visma_icp_plan_ring_grid: the cell table the library would plan for a target with bounding box [mn, mx], search radius
 *   max_dist and a wished cell edge `cell` < max_dist: cells per axis, the edge it ends up with (grown until the table fits
 *   64 M cells, 2048 per axis and 64 rings), *rings = the largest ring of rows a radius can reach + 1 (0: the ordinary plan
 *   with radius-sized cells -- the wished edge was not smaller than the radius).
 * visma_icp_ring_visiting_order: the rows' visiting order for `rings` rings, (2 rings + 1)^2 entries nearest first: offsets
 *   (dy, dz) and base = max(|dy| - 1, 0)^2 + max(|dz| - 1, 0)^2; writes min(capacity, *nrows) entries.  Any pointer may be NULL.
 * Replace nothing in the reference. */
VISMA_ICP_API int visma_icp_plan_ring_grid(const float mn[3], const float mx[3], double max_dist, double cell, int dims[3],
                                           double *cell_out, int *rings);
VISMA_ICP_API int visma_icp_ring_visiting_order(int rings, int capacity, short *dy, short *dz, float *base, int *nrows);

/* Compile-time tile constants, for roofline accounting: S_TILE source points
 * per workgroup, target chunk staged per LDS fill, workgroup size. */
VISMA_ICP_API int visma_icp_get_tile_config(int *s_tile, int *t_chunk, int *block);
/* Launch geometry of the last nn_pass: source tiles x target splits. */
VISMA_ICP_API int visma_icp_get_launch_config(visma_icp_ctx *ctx, int *src_tiles,
                                              int *tgt_splits);

/* Device time (HIP events) of the distance kernel, and of building the search
 * structure (Morton sort + BVH), of the last visma_icp_point_mesh_distance /
 * visma_icp_measure_surface_error call.  Either pointer may be NULL. */
VISMA_ICP_API int visma_icp_last_mesh_kernel_ms(visma_icp_ctx *ctx, double *query_ms, double *build_ms);
/* 0 = choose (default: BVH from 64 faces up), 1 = brute force over all faces,
 * 2 = BVH.  Both give the same minimum, face and closest point, bit for bit. */
VISMA_ICP_API int visma_icp_set_mesh_search(visma_icp_ctx *ctx, int method);

/* Device self-test of the SO(3) math the kernels are built on (restatement of
 * core/rodrigues.h:143-226 in visma_amd/csrc/so3.h): for n axis-angle vectors
 * w (3n doubles) computes, ON THE GPU, R = rodrigues(w) (9n) and
 * w_back = invrodrigues(R) (3n). */
VISMA_ICP_API int visma_icp_selftest_so3(const double *w, double *R, double *w_back, int n);
/* The same ON THE GPU with the derivatives and the projection: for n axis-angle vectors w,
 * R (9n), dR/dw (27n), w_back (3n), dw/dR (27n) and project_so3 of a sheared copy of R (9n). */
VISMA_ICP_API int visma_icp_selftest_so3_jac(const double *w, int n, double *R, double *dR_dw,
                                             double *w_back, double *dw_dR, double *proj);

/* visma_se3_compose / _act / _inv (visma_icp.h) run ON THE GPU for n elements */
VISMA_ICP_API int visma_icp_selftest_se3(const double *g, const double *h, const double *v, int n,
                                         double *gh, double *gv, double *g_inv);

#ifdef __cplusplus
}
#endif
#endif /* VISMA_ICP_TESTING_H */
