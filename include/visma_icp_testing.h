/* visma_icp_testing.h -- the TEST SEAM of the library's host driver.  Not part of the product ABI:
 * libvisma_icp.so does not export this entry point; only the side build
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

#ifdef __cplusplus
}
#endif
#endif /* VISMA_ICP_TESTING_H */
