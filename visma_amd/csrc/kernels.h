// kernels.h -- launch interface of the gfx950 ICP kernels (kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace visma {

// ---- compile-time tile constants (reported by visma_icp_get_tile_config) ----
constexpr int kBlock = 256;         // threads per workgroup (4 wave64)
constexpr int kSptLarge = 8;        // source points per thread, large clouds
constexpr int kSptSmall = 2;        // ... small clouds (more workgroups)
constexpr int kSTile = kBlock * kSptLarge;  // S_TILE = 2048 source points / workgroup
constexpr int kTChunk = 512;        // target points staged per LDS fill
constexpr int kSub = 64;            // sub-chunk whose winner id is tracked
constexpr int kReduceAcc = 29;      // per-thread f64 accumulators (max over modes)
constexpr int kNStats = 38;

struct Xform32 { float m[12]; };    // row-major 3x4, fp32 (NN search)
struct Xform64 { double m[12]; };   // row-major 3x4, f64  (statistics)
struct Offset64 { double v[3]; };   // frame shift applied to p and q in the statistics
// A point in f64 for the double-precision search (32 B: x, y, z, original index).
struct Pt64 {
    double x, y, z;
    unsigned long long w;
};

// State of one ICP problem advanced entirely on the device (icp_loop.hip): the
// NN / reduction kernels read the current transform from it, the one-thread
// solve kernel updates it, the host only reads it back at the end.
struct DevIcpState {
    double Tc[12];       // current transform, centred frame (row-major 3x4)
    double centre[3];    // cloud centre (frame offset of GN / point-to-plane statistics)
    double stats[kNStats];
    double fit, rmse, fit_prev, rmse_prev, K;
    double rel_fit, rel_rmse;
    long long ns_total;  // fitness denominator (all ranks)
    int iter;            // solves performed
    int passes;          // NN passes performed
    int active;          // 1 while the loop is still running
    int max_iter, solver, scaling, plane, world_frame, check_stop;
    float r2f;
    int have_prev;       // Tc_prev holds the transform of the previous NN pass (the certificate of grid_coop.hip)
    double Tc_prev[12];
};

// The fold of the partial rows inside the search launch (device_common.h: fused_fold).
// Rows per first-level group (launches of more than kFoldSingle rows fold in two levels).  16 since the end of round 4
// (32 before): the last arrivers of the groups -- the slowest workgroups of a launch by construction -- sum half as many
// rows, and the level-2 sum of 64 rows is still one batch of eight loads per thread: C4 22.2 vs 24.2 us per iteration
// in the persistent launch, 25.9 vs 27.8 with one launch per pass (64: 26.4 in the persistent launch; 8: as 16).
#ifndef VISMA_FOLD_GROUP
#define VISMA_FOLD_GROUP 16
#endif
constexpr int kFoldGroup = VISMA_FOLD_GROUP;
constexpr int kIpcMaxRanks = 16;
constexpr long long kIpcSpinLimit = 200000000ll;    // polls of the own mailbox before a peer counts as lost (minutes)
constexpr long long kIpcHandshakeSpins = 15000000ll;  // ... in the handshake of visma_icp_comm_ipc_init (tens of seconds)
struct IpcPeers { void *box[kIpcMaxRanks]; };     // box[r]: rank r's mailbox as mapped HERE (box[rank] = own)

struct FoldArgs {
    unsigned *tickets;            // ticket_stride words per problem, zero; NULL = fold in a separate launch
    double *partials2;            // ticket_stride rows of kReduceAcc per problem
    int ticket_stride;            // >= 1 + ceil(workgroups per problem / 32)
    double *stats_out;            // the 38 statistics of problem b at stats_out + b * stats_stride
    long long stats_stride;
    double *host_out;             // mapped host memory for tagged publication (or NULL)
    unsigned long long seq;
    // source-sharded ranks with peer-to-peer mailboxes (ipc_n > 1): the folding workgroup exchanges the 38
    // statistics with the peers itself before it publishes (no separate all-reduce launch)
    IpcPeers peers;
    int ipc_rank, ipc_n;
    unsigned long long *ipc_seq_dev;   // exchanges performed so far, in DEVICE memory: it advances only when an
                                       // exchange really runs (a launch that returns because its loop has ended
                                       // must not burn a number: the two mailbox halves alternate by it)
    int *ipc_flag;
    long long ipc_spins;
    void *const *peer_table;           // the same mailboxes as `peers`, as a table in DEVICE memory: what the persistent
                                       // kernel's fold reads (a by-value table indexed in a loop would live in scratch there)
    // the POLLED fold of persistent launches (device_common.h: polled_fold): rows and group rows as self-validating
    // granules {low half, tag, high half, tag}, 32 per row; NULL = the ticket fold
    void *rows_tagged, *rows2_tagged;
    unsigned long long *dead_flag;     // the launch's "somebody left" word (kernels.h: kPersistDead)
    long long poll_ticks;              // how long a reducer waits for a row (wall_clock64 ticks) before it gives the launch up
    // device-resident loops with the closed-form (Kabsch) update (round 5): the workgroup that completes problem b's fold
    // advances solve[b] right there (icp_state.h: advance_state<true>) -- no solve launch between two search launches.
    // NULL: the statistics are left in the state and solve_state_kernel advances it (Gauss-Newton modes, point-to-plane,
    // ranks that exchange first).  Honoured by the kernels batches and sweeps run (grid.hip, grid_wave.hip), NOT by
    // the certificate kernels of grid_coop.hip (launch_nn_coop refuses a fold that sets it for them).  Compiled in only
    // with -DVISMA_SOLVE_IN_FOLD=1 (kSolveInFold): it measured slower than the solve launch and its
    // inlined one-thread solve costs the default search kernels scratch and LDS -- default builds carry none of it.
    DevIcpState *solve;
    // (round 6) the PERSISTENT SWEEP launch (grid_wave.hip: nn_wave_kernel_sweep): the workgroup that completes problem b's
    // fold advances solve[b] and hands the next pass's transform to the problem's other workgroups through
    // sweep_relay + 32 b: kPersistWords 8-byte words {half of a double | tag << 32} + the command word (GO / STOP), every
    // word carrying sweep_tag -- the tag of the pass that is due next.  NULL: not that launch.
    unsigned long long *sweep_relay;
    unsigned sweep_tag;
    int sweep_passes;                  // DevIcpState::passes of solve[b] before this pass's update (validates the state read)
};
#ifndef VISMA_SOLVE_IN_FOLD
#define VISMA_SOLVE_IN_FOLD 0
#endif
constexpr bool kSolveInFold = VISMA_SOLVE_IN_FOLD != 0;

// The tag the granule rows of the polled fold validate themselves with: the pass's sequence number folded into
// 1 .. 2^32 - 1 -- never 0 (a cleared buffer validates nothing), the same value again only 2^32 - 1 passes later (the
// host clears the rows before a session whose numbers run through that wrap: hip_engine_passes.cpp).
constexpr unsigned long long kFoldTagPeriod = 0xFFFFFFFFull;
#if defined(__HIPCC__)
__host__ __device__
#endif
inline unsigned fold_row_tag(unsigned long long seq) { return (unsigned)(seq % kFoldTagPeriod) + 1u; }

// The PERSISTENT form of the certificate kernel (grid_coop.hip, round 4b): ONE launch runs up to max_passes ICP
// passes of one registration.  After a pass the workgroup that finished the fold (and published the statistics to
// the host) polls a command block in mapped host memory for the next transform, re-publishes it in device memory
// (relay) for the other workgroups, and everybody runs the next pass -- no relaunch, no dispatch ramp, source and
// state re-read from a warm L2.  Command block / relay: kPersistWords 8-byte words {low or high half of a double |
// tag << 32}, each word validating itself (no fence anywhere): words 0..23 = the 3 x 4 transform, word 24 = command.
// Every wait is bounded by the 100 MHz wall clock: a launch whose host went away ends by itself.
constexpr int kPersistWords = 25;
constexpr int kPersistPublished = 31;   // relay word: the tag of the pass whose statistics have been published
constexpr int kPersistDead = 29;        // relay word: nonzero once a workgroup of this launch has left early (the launch can never fold again)
constexpr int kPersistStarted = 30;     // relay word: workgroups of this launch that have begun (zeroed on the stream before the launch)
constexpr unsigned kPersistGo = 1u, kPersistStop = 2u, kPersistAbort = 3u;
struct PersistArgs {
    const unsigned long long *host_cmd;   // mapped, coherent host memory (device pointer), kPersistWords words -- or, `direct`,
                                          // fine-grained DEVICE memory the host stores into through the PCIe BAR: then every
                                          // workgroup polls it itself (no PCIe read per poll, no relay hop)
    int direct;
    unsigned long long *relay;            // device memory, kPersistWords words
    unsigned *host_flag;                  // mapped host memory: set to the pass count reached when a wait ran out
    int max_passes;                       // passes this launch may run (>= 1); the first needs no command
    unsigned tag0;                        // the command for pass p (1-based after the first) carries tag0 + p - 1
    long long wait_ticks;                 // how long a workgroup waits for a command AFTER the pass's statistics were published
                                          // (wall_clock64 ticks, 100 MHz).  The HOST never posts GO later than a quarter of
                                          // this after it saw the statistics (it posts STOP and carries on with ordinary
                                          // launches), so no command can arrive while some workgroups have given up already
    long long start_ticks;                // how long a workgroup waits for the launch's OTHER workgroups to begin (round 5:
                                          // 5 ms by default -- when another stream's kernels hold compute units the launch
                                          // cannot complete its residency, and waiting the four patiences of wait_ticks for
                                          // that cost 0.8 s per attempt); 0: wait_ticks
    long long hard_ticks;                 // ... and before that publication (the pass is still running somewhere): a cap that
                                          // only a lost workgroup could reach -- unless not every workgroup of the launch has
                                          // begun by then (another process's persistent launch holds the rest of the compute
                                          // units and waits for ours): then wait_ticks
    unsigned long long *timeline;         // measurement (VISMA_ICP_PERSIST_TIMELINE), else NULL: per pass and workgroup the
    int timeline_passes;                  // 100 MHz clock when the pass began and when its body (fold ticket included) was done
    int wait_first;                       // (round 6) the launch was queued BEHIND the registration's cold pass, before its
                                          // statistics were out: the transform of the launch's first pass is not in the
                                          // kernel arguments -- it comes as a command with tag0 like every later one (which
                                          // then carry tag0 + p instead of tag0 + p - 1), and the launch waits for it first
};

struct NNLaunch {
    int src_tiles;
    int tgt_splits;
    int spt;
};

// Pick the launch geometry for ns source points against nt_pad/kTChunk chunks.
NNLaunch nn_plan(int64_t ns, int64_t nt_pad);

// Fused transform + brute-force nearest neighbour.  tgt must be padded to a
// multiple of kTChunk with +inf points.  keys: [splits][ns_pad] 64-bit
// (d2 bits << 32 | sub-chunk id), no initialisation needed.
hipError_t launch_nn_brute(const float4 *src, int64_t ns, const float4 *tgt,
                           int64_t nt_pad, const Xform32 &T, float r2f,
                           unsigned long long *keys, int64_t ns_pad,
                           const NNLaunch &plan, const DevIcpState *st, hipStream_t stream,
                           const Pt64 *src64 = nullptr, const Xform64 *T64 = nullptr, float *second = nullptr);
// src64 + T64 + second ([splits][ns_pad] floats): the exact flavour (see nn_brute_kernel)

// Merge the per-split keys, recover the exact target index inside the winning
// sub-chunk, then accumulate the per-correspondence Jacobian/residual
// statistics; per-workgroup partials go to `partials`, `finalize` folds them
// (fixed order) into the 38 statistics at `stats_out` (device or mapped host).
// idx_out/d2_out (ns) receive the final correspondence per source point.
// the exact flavour of the brute-force path: f64 clouds (source in engine order, target in the caller's
// order) and the runner-up array written by launch_nn_brute
// Queries whose winner cannot be decided inside the winning sub-chunk (another sub-chunk reaches into the
// rounding band: about one in a thousand) are listed by the reduce kernel and resolved together by two
// passes of ALL workgroups over the target (brute_rescan_kernel), not by a scan of their own.
constexpr int kBrutePendCap = 4096;       // more pending queries than this: the wave scans in place (slow, exact)
struct BrutePend {
    int *count;                           // [0] pending queries (may exceed the capacity)
    float4 *q32;                          // (px, py, pz, L): fp32 query and its squared fp32 filter radius
    Pt64 *q64;                            // f64 query, w = source slot
    unsigned long long *best;             // f64 bits of the best d2 (order preserving), r2d bits = none
    unsigned *best_idx;                   // lowest index among the targets at that distance
};
struct BruteExact {
    const Pt64 *src64, *tgt64, *nrm64;
    const float *second;
    int64_t nt;
    BrutePend pend;                       // pend.count == NULL: every scan in place
};
hipError_t launch_reduce(const float4 *src, int64_t ns, const float4 *tgt,
                         const float4 *tgt_normals, const unsigned long long *keys,
                         int nsplits, int64_t ns_pad, const Xform32 &T32,
                         const Xform64 &T64, const double frame_offset[3], float r2f,
                         int point_to_plane,
                         int32_t *idx_out, float *d2_out, double *partials,
                         int max_partial_blocks, double *stats_out, const DevIcpState *st,
                         int *nblocks_out, hipStream_t stream, double *host_out = nullptr,
                         unsigned long long seq = 0, const BruteExact *ex = nullptr);

int reduce_max_blocks();
constexpr int kSortedSlack = 64;   // entries allocated past the end of a cell-sorted target: the exact grid
                                   // search loads whole batches (<= U*G slots) from a run's first slot
// one-shot all-reduce of the 38 statistics through IPC-mapped mailboxes (kernels.hip)
hipError_t launch_ipc_allreduce(const double *stats_in, double *stats_out, const IpcPeers &peers, int rank,
                                int nranks, unsigned long long *seq_dev, double *host_out, unsigned long long host_seq,
                                int *timeout_flag, hipStream_t stream, long long max_spins = kIpcSpinLimit);
// target-sharded ranks: keys of the local winners / moments of the global winners owned here
hipError_t launch_shard_keys(const int32_t *idx, const float *d2, int64_t ns, unsigned offset,
                             unsigned long long *keys, hipStream_t stream);
hipError_t launch_shard_accumulate(const float4 *src, int64_t ns, const unsigned long long *keys,
                                   const float4 *tgt, int64_t nt_local, unsigned offset,
                                   const float4 *tgt_normals, const Xform64 &T64,
                                   const double frame_offset[3], float r2f, int point_to_plane,
                                   int32_t *idx_out, float *d2_out, double *partials,
                                   int max_partial_blocks, int *nblocks_out, hipStream_t stream);
// ... the same in f64 (exact search on every shard): key 1 = bits of the local f64 d2 (MIN over ranks =
// the global nearest), key 2 = global index of the shards that hold it (MIN = lowest index on exact
// ties); the owner accumulates from the f64 coordinates
hipError_t launch_shard_keys64(const int32_t *idx, const double *d64, int64_t ns, unsigned long long *keys,
                               hipStream_t stream);
hipError_t launch_shard_claim64(const int32_t *idx, const double *d64, const unsigned long long *gkeys, int64_t ns,
                                unsigned offset, unsigned long long *claim, hipStream_t stream);
hipError_t launch_shard_accumulate64(const Pt64 *src64, int64_t ns, const unsigned long long *gkeys,
                                     const unsigned long long *claim, const Pt64 *tgt64, int64_t nt_local,
                                     unsigned offset, const float4 *tgt_normals, const Pt64 *nrm64,
                                     const Xform64 &T64, const double frame_offset[3], double r2d,
                                     int point_to_plane, int32_t *idx_out, float *d2_out, double *partials,
                                     int max_partial_blocks, int *nblocks_out, hipStream_t stream,
                                     const DevIcpState *st = nullptr);
// stats (device) -> host_out[0..37] (mapped host memory), then host_out[38] = seq (u64 bits)
hipError_t launch_publish_stats(const double *stats, double *host_out, unsigned long long seq,
                                hipStream_t stream);

// Fold `nblocks` partial rows into the 38 statistics (one workgroup, fixed order).
// host_out (mapped host memory, may be NULL): also publish the 38 statistics there
// and then raise host_out[38] = seq (u64 bits) -- see launch_publish_stats.
hipError_t launch_finalize(const double *partials, int nblocks, int point_to_plane,
                           double *stats_out, hipStream_t stream, double *host_out = nullptr,
                           unsigned long long seq = 0);
// On-device loop (icp_loop.hip): fold + solve + stop test, no host round trip.
//  launch_finalize_solve : single GPU (fold partials, then advance the state)
//  launch_finalize_state : fold partials into st->stats only (an all-reduce follows)
//  launch_solve_state    : advance the state from st->stats
// nprob problems: workgroup b folds rows [b*nblocks, (b+1)*nblocks) into st[b]
hipError_t launch_finalize_solve(const double *partials, int nblocks, DevIcpState *st, int plane,
                                 int nprob, hipStream_t stream);
hipError_t launch_finalize_state(const double *partials, int nblocks, DevIcpState *st, int plane,
                                 hipStream_t stream);
hipError_t launch_solve_state(DevIcpState *st, int nprob, hipStream_t stream);

// ---- radius-cell uniform grid (grid.hip) -------------------------------------

struct GridParams {
    float mn[3];      // lower corner of the target's bounding box
    float h, inv_h;   // cell edge along x (>= 1.001 * max_dist) and its reciprocal
    int dim[3];       // cells per axis (y and z counted in cells of edge h / sub)
    int sub;          // 1, or 2: rows (y,z) at half pitch -- 25 thinner rows instead of 9
    float hs, inv_hs; // h / sub and its reciprocal
    int ring;         // 0: cells at least as large as the search radius (27-cell neighbourhoods); > 0: cells SMALLER than
                      // the radius, searched in rings of rows (grid_ring.hip) -- the largest ring a radius can reach, + 1
                      // (in the 4 bytes the alignment of ncell left free: the structure every search kernel takes by value
                      //  has the size it had -- 8 bytes more cost the persistent launch 0.25 us per pass)
    int64_t ncell;
};
static_assert(sizeof(GridParams) == 56, "GridParams is a kernel argument of every search kernel");
// one row of the ring search's visiting order: its offset from the query's row and the squared distance (in cells) that
// every point of it is at least away from any point of the query's cell row: max(|dy| - 1, 0)^2 + max(|dz| - 1, 0)^2
struct RingRow { short dy, dz; float base; };
// ... the whole order for a grid with g.ring rings: (2 ring + 1)^2 entries, nearest first (device memory)
struct RingTable { const RingRow *rows; int nrows; };
constexpr int kRingMaxRings = 64;   // cells are enlarged until a radius spans no more rings than this
constexpr int64_t kGridMaxCells = 64ll * 1024 * 1024;        // at sub = 1
constexpr int64_t kGridMaxCellsFine = 256ll * 1024 * 1024;   // at sub = 2 (1 GiB table)
constexpr int kGridMaxDim = 2048;   // per axis at sub = 1: keeps the fp32 binning error < 1e-3 cell

// One problem of a batch with its OWN clouds (offsets into concatenated arrays).
struct ProbDesc {
    long long src_off, sorted_off, start_off, out_off;
    int ns, first_block, nblocks, pad_;
    GridParams g;
};

#ifdef VISMA_WITH_TILE   /* experiment, side build only (see HipEngine::use_tile) */
// ---- streamed radius-cell search with exact (f64) tie-breaks and fused fold (tile.hip) ------
struct TileArgs {
    const Pt64 *src64;            // source, f64 (centred), Morton order
    int ns;
    const float4 *sorted;         // cell-sorted target, fp32 view (w = original index)
    const Pt64 *sorted64;         // ... and the f64 points in the same slots
    const unsigned *start;        // cell table
    GridParams g;
    const float4 *nrm;            // target normals by original index (point-to-plane)
    const Pt64 *nrm64;
    Xform64 T64;
    Offset64 off;
    float r2f;
    int *idx_out;
    float *d2_out;
    double *partials;             // one row of kReduceAcc per workgroup
    double *partials2;            // ticket_stride rows per problem (level-2 rows of the fused fold)
    unsigned *tickets;            // ticket_stride words per problem, zero; NULL = no fused fold
    int ticket_stride;            // >= 1 + ceil(workgroups per problem / 32)
    double *stats_out;            // the 38 statistics of problem b at stats_out + b * stats_stride
    long long stats_stride;
    double *host_out;             // mapped host memory for tagged publication (or NULL)
    unsigned long long seq;
    unsigned long long *stats;    // profiling counters, 512 x 8 (or NULL)
    const DevIcpState *st;        // device loop state (or NULL)
    int bpp;                      // workgroups per problem (shared clouds)
    long long out_stride;
    const ProbDesc *descs;        // problems with their own clouds (or NULL)
    int nprob;
    int force_fallback;           // testing: search from global memory in every workgroup
};
// workgroup size of a tile configuration (queries per workgroup)
int tile_threads(int config);
// total_blocks = sum over problems of ceil(ns / tile_threads(config))
hipError_t launch_nn_tile_reduce(const TileArgs &a, int point_to_plane, int config, int total_blocks,
                                 hipStream_t stream);
#endif
// float4 (x, y, z, .) -> Pt64 (x, y, z, index): f64 view of clouds uploaded as fp32
hipError_t launch_promote_pt64(const float4 *src, Pt64 *dst, int64_t n, hipStream_t stream);
// order.hip: the source cloud (caller's f64 points, already on the device) into Morton order
size_t order_source_scratch_bytes(int64_t n);
hipError_t order_source_device(const double *d_xyz, int64_t n, const double c[3], float4 *d_src, Pt64 *d_src64,
                               int32_t *d_order, void *scratch, size_t scratch_bytes, hipStream_t stream);
// raw f64 xyz (3 doubles per point) -> f4[j] = ((float)(x - c), .., 0) and, when p8 != NULL, p8[j] = {x - c, .., j}
hipError_t launch_expand_f64(const double *xyz, int64_t n, const double c[3], float4 *f4, Pt64 *p8, hipStream_t stream,
                             int64_t index0 = 0);
// ... the same from values fp32 holds exactly (3 floats per point): p8[j].w = index0 + j
hipError_t launch_expand_f32(const float *xyz, int64_t n, int64_t index0, const double c[3], float4 *f4, Pt64 *p8,
                             hipStream_t stream);

hipError_t launch_grid_bbox(const float4 *tgt, int64_t nt, unsigned *box6, hipStream_t stream);
hipError_t launch_pack12(const float4 *src, float *dst, int64_t n, hipStream_t stream);   // (x,y,z,w) -> packed (x,y,z)
void grid_decode_bbox(const unsigned box6[6], float mn[3], float mx[3]);
GridParams grid_plan(const float mn[3], const float mx[3], double max_dist, int64_t max_cells, int max_sub = 1);
// the same table with cells of edge `cell` < max_dist (enlarged until the table fits), for the ring search (grid_ring.hip)
GridParams grid_plan_ring(const float mn[3], const float mx[3], double max_dist, double cell, int64_t max_cells);
int grid_scan_blocks(int64_t ncell);

// ---- mesh steps (mesh.hip): host arrays in, host arrays out -------------------
// method: 0 = choose, 1 = brute force (LDS-tiled scan of all faces), 2 = BVH
hipError_t point_mesh_distance_device(const double *h_P, int64_t np, const double *h_V, int64_t nv,
                                      const int32_t *h_F, int64_t nf, int method, double *h_d2, int32_t *h_face,
                                      double *h_closest, float *kernel_ms, float *build_ms, hipStream_t stream);
hipError_t sample_mesh_transformed_device(const double *h_V, int64_t nv, const int32_t *h_F, int64_t nf, int64_t n,
                                          int quirks, unsigned long long seed, const double *T16, double *d_out,
                                          int64_t room, int64_t *m_out, hipStream_t stream);
hipError_t sample_mesh_device(const double *h_V, int64_t nv, const int32_t *h_F, int64_t nf, int64_t n,
                              int quirks, unsigned long long seed, const double *h_uniforms,
                              double *h_out, int64_t *n_out, hipStream_t stream);
hipError_t surface_distances_device(const double *h_Vs, int64_t nvs, const int32_t *h_Fs, int64_t nfs,
                                    const double *h_Vt, int64_t nvt, const int32_t *h_Ft, int64_t nft, int64_t n,
                                    int quirks, unsigned long long seed, int method, double *h_dist,
                                    int64_t *n_out, float *kernel_ms, float *build_ms, hipStream_t stream);

// ---- normal estimation (normals.hip): host arrays in, host arrays out ----------
// open3d::EstimateNormals; search_type 0 KNN(knn) | 1 Radius(radius) | 2 Hybrid(radius, max_nn = knn)
constexpr int kNormalsMaxList = 170;     // longer result lists live in global memory (a heap per point) instead of LDS
hipError_t estimate_normals_device(const double *h_xyz, int64_t n, const double *h_nrm_in, int search_type,
                                   int knn, double radius, double *h_out, hipStream_t stream);

// ---- voxel down-sampling (voxel.hip): host arrays in, host arrays out ---------
hipError_t voxel_down_sample_device(const double *h_xyz, const double *h_nrm, const double *h_col,
                                    int64_t n, double voxel, double *h_out_xyz, double *h_out_nrm,
                                    double *h_out_col, int64_t *n_out, int *too_fine,
                                    hipStream_t stream);
// ... the device part alone (resident input, hipMalloc'ed resident output that the caller frees), and the centroid
// of resident points with the host's summation order (centroid_f64): what the device-resident caller pipeline uses
hipError_t voxel_down_sample_core(const double *d_xyz, const double *d_nrm, const double *d_col, int64_t n, double voxel,
                                  double **d_oxyz_out, double **d_onrm_out, double **d_ocol_out, int64_t *n_out,
                                  int *too_fine, hipStream_t stream);
hipError_t centroid_device(const double *d_xyz, int64_t n, int64_t chunk, double *d_part, double *d_out, hipStream_t stream);
// counting sort of the target by cell: start[ncell+1], sorted[nt] = (x,y,z, bits(orig index))
// tgt64 / sorted64 (both or neither): the f64 copy of the target is scattered in the same order
size_t exclusive_scan_u32_tmp_bytes(long long n);
hipError_t launch_exclusive_scan_u32_lib(const unsigned *in, long long n, unsigned *out, void *tmp, size_t tmp_bytes,
                                         hipStream_t stream);
// (cell_of: 2 * nt words; bsum: grid_scan_blocks(g.ncell) + 1 words)
hipError_t launch_grid_build(const float4 *tgt, int64_t nt, const GridParams &g,
                             unsigned *cell_of, unsigned *count, unsigned *bsum,
                             unsigned *start, float4 *sorted, hipStream_t stream,
                             const Pt64 *tgt64 = nullptr, Pt64 *sorted64 = nullptr);
// fused transform + grid NN + Jacobian/residual + reduction to partial rows
// (launch_finalize folds them); also writes idx_out / d2_out.
hipError_t launch_nn_grid_reduce(const float4 *src, int64_t ns, const float4 *sorted,
                                 const unsigned *start, const GridParams &g,
                                 const float4 *tgt_normals, const Xform32 &T32, const Xform64 &T64,
                                 const double frame_offset[3], float r2f, int point_to_plane,
                                 int32_t *idx_out, float *d2_out, double *partials,
                                 int max_partial_blocks, int *nblocks_out, int lanes_per_query,
                                 unsigned long long *cand_count, const DevIcpState *st,
                                 int nprob, int64_t out_stride, hipStream_t stream,
                                 const Pt64 *src64 = nullptr, const Pt64 *sorted64 = nullptr,
                                 double r2d = 0.0, const Pt64 *nrm64 = nullptr, int exact = 0,
                                 const FoldArgs *fold = nullptr, double *d64_out = nullptr,
                                 Pt64 *wst_io = nullptr, int warm = 0, const Xform64 *Tprev = nullptr,
                                 const PersistArgs *persist = nullptr, Pt64 *ru_io = nullptr,
                                 const RingTable *ring = nullptr);
// ring: the visiting order of a grid with g.ring > 0 (grid_ring.hip).
// persist: run the launch as the persistent certificate kernel (one problem, one query per lane, fused fold with
// host publication; hipErrorInvalidValue where that does not apply -- ask coop_persist_capacity first).
// wst_io (one Pt64 per query, laid out like idx_out): the exact searches leave their winners there (f64 point,
// original index | LB << 32 -- see grid_coop.hip; NaN coordinates = none, all bits set = nothing known); the
// warm-started search (kCoopLanes) reads them when `warm & 1`, and with Tprev (the transform of the pass that left
// them; device loops: from their DevIcpState) skips the search of queries whose winner provably cannot have changed.
// The exact search with the candidates of a wave flattened over its lanes (grid_coop.hip); selected by
// lanes_per_query = kCoopLanes (G = 1, U = 99) in the two launchers around it, which fall back to the
// lane-serial kernel where it does not apply (fp32-only / all-f64 search, half-pitch rows).
constexpr int kCoopLanes = 9901;
// (grid_wave.hip: round 3's kernel behind the same lanes code -- launch_nn_coop hands batches and sweeps to it)
hipError_t launch_nn_wave(int total_blocks, int bpp, int nprob, const ProbDesc *descs, int ns, const float *s12,
                          const unsigned *start, const GridParams &g, const float4 *nrm, const Pt64 *nrm64,
                          const Xform64 &T64, const Offset64 &off, float r2f, int point_to_plane, int one,
                          int32_t *idx_out, float *d2_out, double *partials, unsigned long long *cand_count,
                          const DevIcpState *st, long long out_stride, const Pt64 *src64, const Pt64 *sorted64,
                          const FoldArgs &fold, double *d64_out, Pt64 *prevq_io, int warm, hipStream_t stream);
// The persistent form of the batch / sweep search over SHARED clouds (grid_wave.hip, round 6): ONE launch runs up to
// max_passes warm passes of nprob problems of bpp workgroups each -- search, fold, closed-form update, compose and stop
// test inside the launch (fused_fold<..., LOOPED, SOLVE>), the next transform handed to the problem's workgroups through
// `relay` (32 words per problem, device memory, zeroed by the caller).  Every workgroup must be resident at once
// (nn_wave_sweep_capacity()).  `dead`: one word, zeroed by the caller; nonzero afterwards = a wait ran out (somebody's
// workgroups were not resident): the states hold what was completed, the caller carries on with ordinary launches.
struct SweepArgs {
    unsigned long long *relay;
    unsigned long long *dead;
    int max_passes;
    unsigned tag0;             // the command for pass p (p >= 1) carries tag0 + p; pass 0 reads the state itself
    int passes0;               // DevIcpState::passes of every problem when the launch begins (its first pass makes it passes0 + 1)
    long long wait_ticks;      // how long a workgroup waits for its problem's next transform (100 MHz ticks)
};
// ---- the ring search over cells smaller than the radius (grid_ring.hip; g.ring > 0): `nblocks` workgroups per problem,
// eight lanes per query; s12 (the packed fp32 copy of sorted64, 12 bytes per point) or NULL: candidates ranked in fp32 with
// the f64 re-rank of the rounding band, or in f64 throughout -- same results.  state_io: per query the winner's f64 point and original index (all bits
// set = none), read when `warm`, always written.
hipError_t launch_nn_ring(int lanes, int nblocks, int nprob, int ns, const Pt64 *src64, const Pt64 *sorted64, const float *s12, const unsigned *start,
                          const GridParams &g, const RingTable &tab, const float4 *nrm, const Pt64 *nrm64, const Xform64 &T64, const Offset64 &off,
                          float r2f, int point_to_plane, int32_t *idx_out, float *d2_out, double *d64_out, Pt64 *state_io,
                          int warm, double *partials, unsigned long long *cand_count, const DevIcpState *st,
                          long long out_stride, const FoldArgs &fold, hipStream_t stream);
// the visiting order of a grid with g.ring = rings (a new device buffer the caller owns: hipFree); *nrows = (2 rings + 1)^2
hipError_t build_ring_table(int rings, void **d_tab, int *nrows);
// ... the same order on the host (empty for rings outside 1 .. kRingMaxRings)
std::vector<RingRow> ring_visiting_order(int rings);
// *out (device, zeroed by the caller) += the number of non-zero entries of count[0 .. n)
hipError_t launch_count_occupied(const unsigned *count, int64_t n, unsigned long long *out, hipStream_t stream);
int nn_wave_sweep_capacity();
hipError_t launch_nn_wave_sweep(int bpp, int nprob, int ns, const float *s12, const unsigned *start, const GridParams &g,
                                float r2f, int32_t *idx_out, float *d2_out, double *partials, unsigned long long *cand_count,
                                DevIcpState *st, long long out_stride, const Pt64 *src64, const Pt64 *sorted64,
                                const FoldArgs &fold, Pt64 *prevq_io, const SweepArgs &sa, hipStream_t stream);

hipError_t launch_nn_coop(int total_blocks, int bpp, int nprob, const ProbDesc *descs, int ns, const float *s12,
                          const unsigned *start, const GridParams &g, const float4 *nrm, const Pt64 *nrm64,
                          const Xform64 &T64, const Offset64 &off, float r2f, int point_to_plane, int one,
                          int32_t *idx_out, float *d2_out, double *partials, unsigned long long *cand_count,
                          const DevIcpState *st, long long out_stride, const Pt64 *src64, const Pt64 *sorted64,
                          const FoldArgs &fold, double *d64_out, Pt64 *wst_io, int warm, hipStream_t stream,
                          const Xform64 *Tprev = nullptr, const PersistArgs *persist = nullptr, Pt64 *ru_io = nullptr);
// workgroups of the persistent kernel the current device holds at once (0: none -- do not launch it)
int coop_persist_capacity(int point_to_plane);
// Batch of problems with different clouds: `descs` (device) gives every problem's
// offsets / grid / workgroup range; total_blocks = sum of descs[].nblocks.
hipError_t launch_nn_grid_reduce_batch(const float4 *src, const float4 *sorted, const unsigned *start,
                                       const ProbDesc *descs, int nprob, int total_blocks,
                                       int32_t *idx_out, float *d2_out, double *partials,
                                       int lanes_per_query, int one_per_lane, const DevIcpState *st,
                                       hipStream_t stream, const Pt64 *src64 = nullptr,
                                       const Pt64 *sorted64 = nullptr, int exact = 0,
                                       const FoldArgs *fold = nullptr, unsigned long long *cand_count = nullptr,
                                       const float4 *nrm = nullptr, const Pt64 *nrm64 = nullptr,
                                       Pt64 *wst_io = nullptr, int warm = 0);
// (nrm / nrm64: target normals concatenated like the UNSORTED targets -> the point-to-plane estimator)
hipError_t launch_finalize_solve_batch(const double *partials, const ProbDesc *descs, DevIcpState *st,
                                       int nprob, hipStream_t stream, int plane = 0);

// fill n float4 with +inf (target padding)
hipError_t launch_fill_inf(float4 *dst, int64_t n, hipStream_t stream);
// AoS stride-s floats -> float4 (w = 0)
hipError_t launch_pack_float4(const float *src, int stride, float4 *dst, int64_t n,
                              hipStream_t stream);

// SO(3) self-test kernel: R = rodrigues(w), w2 = invrodrigues(R), v2 = g*v
hipError_t launch_so3_selftest(const double *w, double *R, double *w2, int n,
                               hipStream_t stream);
hipError_t launch_se3_selftest(const double *g, const double *h, const double *v, int n, double *gh, double *gv,
                               double *gi, hipStream_t stream);
hipError_t launch_so3_selftest_jac(const double *w, int n, double *R, double *dR, double *w2, double *dw, double *proj,
                                   hipStream_t stream);

}  // namespace visma
