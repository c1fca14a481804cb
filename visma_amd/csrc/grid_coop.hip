// grid_coop.hip -- the exact radius-cell search, WARM-STARTED and with the candidates of a WAVE flattened
// over its lanes (round 3; `lanes` code kCoopLanes; used for every pass after the first of a registration).
//
// Why.  nn_grid_reduce_kernel (grid.hip) gives every query one lane that walks its own rows: a load
// instruction of that kernel touches 64 different cache lines (one 12-byte candidate per lane), a wave
// executes max-over-its-64-queries batch trips (10.3 where the average query needs 5.2) and every trip
// is a dependent memory round trip because the next row is chosen from the best distance so far.  At C4
// (262,144 queries = ONE resident round of 4096 waves) the launch lasts as long as one wave's chain of
// ~17 round trips, and in throughput terms it is co-limited by VALU issue and the L1 address pipeline
// (one line per lane per load), both ~60 % busy (1 M queries against the same target: 117 us).
//
// Two changes, both exact:
//  * WARM START.  ICP asks the same queries again after a small motion.  The previous pass's winner
//    (wst_io: its f64 point, 32 bytes per query, streamed in with the source point; ranked on as fp32) gives
//    d_ub = |p - q_prev|^2 BEFORE anything is gathered, so ALL pruning happens up front and per CELL: only
//    the rows whose slab reaches within d_ub are looked up in the cell table (1.9 of 9), and of their cells
//    only those whose slab bound (same margins as the lane-serial kernel) does not exceed d_ub are listed
//    -- 19 candidates per query instead of 33 -- and no pruning decision depends on a load any more.  q_prev is a real target point, so everything
//    nearer than it (and it itself) lies in the listed cells: the result cannot differ from a full scan.
//    A query without a previous winner prunes against the radius.
//  * FLATTENING.  A wave still owns 64 queries (lane = query; the query -> lane map and the summation tree
//    of the lane-serial kernel, so the statistics are bit-identical), but their chunks (<= 8 consecutive
//    candidates = <= 96 contiguous bytes) go to a list in LDS that the wave works off eight lanes per
//    chunk, one candidate per lane: a load instruction touches <= 8 short segments instead of 64 lines, no
//    lane idles while another query's row is longer, and all loads are independent (4 chunks per lane
//    octet in flight).
// A query's chain is  source + previous winner -> row bounds -> chunks -> f64 winner:  4 dependent trips
// instead of ~17.
//  * CERTIFICATE (round 4).  The per-query state is the winner's f64 point (x, y, z, original index) and, in the
//    upper half of its last word, LB: a lower bound of the distance from the query AS IT STOOD to every target
//    point BUT the winner (to every target point when the query had no partner).  A full search leaves it behind:
//    the smallest of (a) the second-best examined candidate (second chunk minimum, and the best non-flagged
//    candidate of every chunk), (b) the slab bounds of the cells of the 27 that were not listed, (c) the distance
//    to the outside of the 27 cells (or to the grid's bounding box for a query far from it), less the rounding
//    band.  The next pass moves the query by delta = |T s - T_prev s| (computed per query), so every other target
//    point is still at least t = LB - delta away (triangle inequality); if the old winner's NEW distance --
//    the reference's f64 arithmetic -- is inside the radius and below t, it is the unique nearest neighbour again:
//    no cell-table row, no chunk, no gather -- source + state in (64 B), distance + state word out.  A query
//    without a partner stays without one when t exceeds the radius.  ICP's motion shrinks geometrically: after a
//    few iterations almost every query certifies and the pass streams.  LB <- t; anything else runs the search
//    below and renews LB.  Near-ties (more than one flagged candidate, a re-scan) leave LB = 0: never certified.
//
// Exactness without per-candidate top-2 bookkeeping: the eight lanes of a chunk reduce
// their fp32 d2 to the chunk minimum m (three DPP steps) and flag every candidate with d2 <= m + W, W an
// upper bound of band(m) - m for any m inside the radius (band(m) = (sqrt(m) + 2E)^2, E the rounding
// half-width of exact_band()).  The query keeps its two best CHUNKS (m, first slot, flag byte) and the
// third chunk minimum.  Any candidate within the rounding band of the global fp32 minimum g lies in a
// chunk with m <= g + W and is flagged there (d2 <= band(g) <= band(m) <= m + W), so ranking the flagged
// candidates of the kept chunks with m <= g + W in f64 (reference arithmetic, lowest original index on
// exact ties, strict d2 < (double)(float)r2) returns the reference's correspondence; a third chunk inside
// g + W, or more than four flagged candidates, sends the query to the f64 re-scan of its 27 cells (points
// given several times).
#include <cstdlib>

#include "device_common.h"
#include "grid_coop_probe.h"   // COOP_MARK: no-ops outside the measurement builds

namespace visma {

namespace {

struct P12 { float x, y, z; };                // fp32 rounding of a cell-sorted f64 target point
#ifndef VISMA_COOP_RU
#define VISMA_COOP_RU 0
#endif
#ifndef VISMA_COOP_CAP
#define VISMA_COOP_CAP (VISMA_COOP_RU ? 344 : 352)
#endif
// RUNNER-UP-AWARE certificates (round 5b).  What limits the bound LB a search leaves is the runner-up: in 82-85 % of the
// searched queries of a C4 registration the second-best examined candidate, on average 1.0 mm below what the cells not
// listed would allow (tools/lb_probe.py, profiles/r05_lb_probe.txt).  So the state also keeps the RUNNER-UP u (f64 point and
// index, 32 B per query in global memory: ru_io) and LB3, a lower bound of the distance to every target point but the
// winner w and u.  When the winner-unchanged certificate fails (d_w >= LB - delta), both distances are evaluated exactly
// (the reference's arithmetic): if the nearer of the two lies inside the radius and below LB3 - delta, it is the unique
// nearest neighbour (lowest original index on an exact tie, as rank()) -- no search; w and u change places if u won.
// MEASURED AND NOT ADOPTED (compiled out by default; -DVISMA_COOP_RU=1 builds it, and VISMA_ICP_RUNNER_UP=1 makes the
// contexts of such a build allocate and pass the runner-up buffer): it certifies what it promises -- 53 % of the queries at pass 2 of a C4 registration instead of 33 %,
// 71 % at pass 10 instead of 56 %, 87 % at pass 20 instead of 75 % (profiles/r05_cert_probe_runner_up.txt), same results
// bit for bit -- and the kernel gets SLOWER: it sits at the 128 registers four waves per SIMD allow, and the three best
// candidates per chunk (one more octet minimum, a ballot, a packed side word), the side word through the merge and the
// runner-up's choice behind the f64 winner cost 2.7 us per pass before a single query is certified by it (13 registers
// spilled instead of 8 -- the first build, which also fetched the runner-up's point beside the winner's, spilled 29 and
// lost 5 us); with it switched on, 35.7-36.0 us per iteration over 1..20 from the identity against 33.4-33.8 without the
// code, 22.8 against 21.2-21.5 at the converged pose (tools/ab_probe.py, same box, alternating runs).
constexpr bool kCoopRu = VISMA_COOP_RU != 0;
constexpr int kCoopCap = VISMA_COOP_CAP;      // chunk descriptors per wave and list window (1,408 per workgroup: 16.5 KiB with the
                                              // side array; a workgroup of 170 queued queries -- the second pass of a C4
                                              // registration -- lists ~850 chunks: at 768 it took two windows and 5-9 us more).
                                              // 384 until round 5: the persistent kernel keeps every query's source point and
                                              // state in LDS now (and re-derives the transformed query instead of keeping it)
#ifndef VISMA_COOP_DEPTH
#define VISMA_COOP_DEPTH 7
#endif
constexpr int kCoopDepth = VISMA_COOP_DEPTH;  // chunks per lane octet in flight
#ifndef VISMA_COOP_ROOM
#define VISMA_COOP_ROOM 0.04f
#endif
constexpr float kCoopRoom = VISMA_COOP_ROOM;  // listing margin beyond the previous winner's distance, in search radii

// minimum over the 8 lanes of an aligned lane octet: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror.
// (v_min_f32_dpp reads its first source through the permutation; a VALU result needs two wait states
// before a DPP read, which the compiler does not insert inside an asm block.)  d is never NaN.
__device__ __forceinline__ float octet_min(float d)
{
    float r;
    asm volatile("s_nop 1\n\t"
                 "v_min_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf"
                 : "=&v"(r)
                 : "v"(d));
    return r;
}

// the side word of a chunk result (s_sec): both values are lower bounds after the truncation (d2 >= 0: sign bit clear)
// (without the runner-up code the word is sec itself, unrounded)
__device__ __forceinline__ unsigned pack_sec(float sec, float third, unsigned lane)
{
    if constexpr (!kCoopRu) return __float_as_uint(sec);
    return (__float_as_uint(sec) & 0xFFFF0000u) | (((__float_as_uint(third) >> 18) & 0x1FFFu) << 3) | (lane & 7u);
}
__device__ __forceinline__ float sec_of(unsigned w) { return __uint_as_float(kCoopRu ? (w & 0xFFFF0000u) : w); }
[[maybe_unused]] __device__ __forceinline__ float third_of(unsigned w) { return __uint_as_float(((w >> 3) & 0x1FFFu) << 18); }

// inclusive prefix sum over the 64 lanes: DPP row shifts inside the 16-lane rows (absent lanes read 0), then the
// last lane of row 0 / 2 broadcast into row 1 / 3 and lane 31 into the upper half -- no LDS round trips
__device__ __forceinline__ unsigned wave_scan_incl(unsigned v, int)
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

}  // namespace

// The PERSISTENT launch keeps the state a query carries from pass to pass in LDS (round 5): the winner's f64 point,
// index and LB.  The first pass of a launch reads it from global memory, the passes after it do not touch it there: a
// certified query reads its source point (32 B, from a warm L2) and writes nothing, and nothing of a pass is written
// back until the launch ends (its last lines).  The one-pass kernels pass nullptrs.
// (Measured, round 5: the source points in LDS as well -- no global access at all for a certified query -- left no room
//  for the transformed queries, which every user then re-derived: 24 more scalar registers live across the search, 26 of
//  them spilled, and the chunk-list window had to shrink; 0.5-1 us slower per pass than this.)
struct CoopRes {
    double (*q64)[3];      // [NTH] state: the winner's f64 point (NaN: none)            } = the s_q64 / s_dprev slots the
    float *idx;            // [NTH] state: the winner's original index (bits; ~0: none)  } passes hand their results over in
    float *lb;             // [NTH] state: LB
    float *lb3;            // [NTH] state: LB3 (the runner-up's point and index stay in global memory: ru_io)
    int first;             // this pass is the launch's first: source and state come from global memory
    // (round 6) the lane's source point, asked for BEFORE the wait for this pass's command (the source never changes: the
    // load's round trip -- the first thing every pass waited for -- lies under the fold and the host's turn-around)
    double sx, sy, sz;
    int have_src;
};

// the query of thread `tid` of workgroup `lb` of a problem of `bpp` workgroups (the query -> lane map of
// nn_grid_reduce_kernel<G = 1>: XCD-aware chunking of the Morton order)
template <int NTH>
__device__ __forceinline__ void coop_query_range(int ns, int bpp, int lb, int tid, int &vb, int &per_group, long long &i_begin,
                                                 long long &i_end)
{
    vb = lb;
    if ((bpp & 7) == 0) vb = (lb & 7) * (bpp >> 3) + (lb >> 3);
    const int total_groups = bpp * NTH;
    per_group = (ns + total_groups - 1) / total_groups;
    const int gid = vb * NTH + tid;
    i_begin = (long long)gid * per_group;
    i_end = i_begin + per_group < ns ? i_begin + per_group : ns;
}

// Returns true on the one workgroup that finished the fold and published the statistics (fused fold only).
template <bool PLANE, bool ONE, int NTH, bool PERSIST = false>
__device__ __forceinline__ bool coop_body(
    int ns, const float *__restrict__ s12f, const unsigned *__restrict__ start, GridParams g,
    const float4 *__restrict__ nrm, Xform64 T64, Offset64 off, float r2f, int *__restrict__ idx_out,
    float *__restrict__ d2_out, double *__restrict__ partials, unsigned long long *__restrict__ cand_count,
    const DevIcpState *__restrict__ st, int bpp, long long out_stride, const ProbDesc *__restrict__ descs,
    int nprob, const Pt64 *__restrict__ src64, const Pt64 *__restrict__ sorted64,
    const Pt64 *__restrict__ nrm64, const FoldArgs &fold, double *__restrict__ d64_out,
    Pt64 *__restrict__ wst_io, int warm, Xform64 Tprev, Pt64 *__restrict__ ru_io, unsigned *work_out = nullptr,
    const CoopRes res = CoopRes{})
{
    constexpr int NACC = Acc<PLANE>::N;
    const P12 *s12 = reinterpret_cast<const P12 *>(s12f);
    int prob, lb;
    long long row0;
    if (descs) {
        // batch of problems with their own clouds (largest p with first_block <= blockIdx.x; wave-uniform)
        if (warm & 2) {
            // (the launcher put a workgroup -> problem map behind the descriptors)
            prob = reinterpret_cast<const int *>(descs + nprob)[blockIdx.x];
        } else {
            int lo = 0, hi = nprob - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (descs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
            }
            prob = lo;
        }
        const ProbDesc d = descs[prob];
        lb = (int)blockIdx.x - d.first_block;
        bpp = d.nblocks;
        src64 += d.src_off;
        ns = d.ns;
        s12 += d.sorted_off;
        sorted64 += d.sorted_off;
        if constexpr (PLANE) {
            if (nrm) nrm += d.sorted_off;
            if (nrm64) nrm64 += d.sorted_off;
        }
        row0 = d.first_block;
        start += d.start_off;
        g = d.g;
        out_stride = 0;
        idx_out += d.out_off;
        d2_out += d.out_off;
        wst_io += d.out_off;
        if (ru_io) ru_io += d.out_off;
    } else {
        prob = blockIdx.x / bpp;
        lb = blockIdx.x - prob * bpp;
        row0 = (long long)prob * bpp;
    }
    if (st) st += prob;
    {
        Xform32 T32_unused;
        if (!load_loop_state(st, T32_unused, T64, off, r2f)) return false;
        if (st) {
            // device-resident loop: the transform of the previous pass travels with the state (advance_state)
            warm &= ~4;
            if (st->have_prev && !(warm & 8)) {              // (warm & 8: certificates switched off, A/B timing)
                warm |= 4;
#pragma unroll
                for (int k = 0; k < 12; k++) Tprev.m[k] = st->Tc_prev[k];
            }
        }
        if (!(warm & 1)) warm &= ~4;
    }
    const double r2d = (double)r2f;                         // (double)(float)(r*r): KDTreeFlann.cpp:184-185
    COOP_PROBE_BEGIN();
    idx_out += (long long)prob * out_stride;
    d2_out += (long long)prob * out_stride;
    wst_io += (long long)prob * out_stride;
    if (ru_io) ru_io += (long long)prob * out_stride;
    double acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++) acc[a] = 0.0;

    // (inside the persistent kernel's loop: whatever derives from the thread's number alone -- LDS addresses, lane
    //  masks -- was hoisted out of the loop and spilled; opaque, it is computed where it is used)
    const int tid = thread_number<PERSIST, NTH>(), lane = tid & 63, wave = tid >> 6;
    const int oct = lane >> 3, l8 = lane & 7;
    constexpr int NW = NTH / 64;                             // waves of the workgroup
    constexpr unsigned kCapAll = (unsigned)(NW * kCoopCap);  // chunk descriptors of the workgroup's list window
    // the query -> lane map of nn_grid_reduce_kernel<G = 1> (XCD-aware chunking of the Morton order)
    int vb, per_group;
    long long i_begin, i_end;
    coop_query_range<NTH>(ns, bpp, lb, tid, vb, per_group, i_begin, i_end);

    __shared__ float4 s_qp[NTH];                            // (px, py, pz, W) of the query each SEARCHER thread took over
    __shared__ uint2 s_item[kCapAll];                       // the workgroup's chunk descriptors, completed by chunk results
    __shared__ unsigned s_sec[kCapAll];                     // per chunk: sec = its best candidate OUTSIDE the rounding band,
                                                            // third = its best candidate but the flagged ones and sec, and sec's
                                                            // lane: sec rounded DOWN to 7 mantissa bits << 16 | third rounded
                                                            // down to 5 mantissa bits << 3 | lane (pack_sec / sec_of / third_of)
    __shared__ double s_p64[NTH][3];                        // the transformed query of every HOME lane, f64 (what a searcher
                                                            // takes over instead of loading and transforming the source again)
    // the partner of every home lane's query: the certified winner (phase A) or what the query's searcher found
    // (phase B); and the home lane's squared fp32 distance to its previous winner (NaN: none) for the searcher, then the
    // partner's index (bits; ~0: none).  In the persistent launch these slots ARE the state the next pass starts from.
    __shared__ double s_q64_own[PERSIST ? 1 : NTH][3];
    __shared__ float s_dprev_own[PERSIST ? 1 : NTH];
    double (*s_q64)[3] = s_q64_own;
    float *s_dprev = s_dprev_own;
    if constexpr (PERSIST) { s_q64 = res.q64; s_dprev = res.idx; }
    __shared__ unsigned short s_home[NTH];                  // the home thread of the query each searcher thread took over
    __shared__ unsigned s_need[NW];                         // queries queued by each wave this round (see `round` below)
    __shared__ unsigned s_m[NW];                            // chunks listed by each wave's searchers
    __shared__ unsigned short s_queue[NW][64];              // the queued queries' home threads

    // outputs of one query: the correspondence and the state for the next pass
    // (`ht`: the query's home thread)
    auto emit = [&](long long i, unsigned ht, bool cert, double bd, unsigned bidx, bool found, double qx, double qy, double qz,
                    float lb_new) {
        if constexpr (PERSIST) {
            // the winner's point and index are handed over / stay in the home lane's slots; the rest of the state:
            // (the distance is re-derived when the launch ends: coop_flush)
            res.lb[ht] = lb_new;
            return;
        }
        const unsigned long long lbw = (unsigned long long)__float_as_uint(lb_new) << 32;
        idx_out[i] = found ? (int)bidx : -1;                 // (also when certified: later stages may reuse the array)
        if (cert) {
            // (winner unchanged: only the bound moves)
            reinterpret_cast<unsigned *>(&wst_io[i].w)[1] = __float_as_uint(lb_new);
        } else {
            Pt64 o8;
            o8.x = o8.y = o8.z = __longlong_as_double(-1ll);
            o8.w = 0xFFFFFFFFull | lbw;
            if (found) { o8.x = qx; o8.y = qy; o8.z = qz; o8.w = (unsigned long long)bidx | lbw; }
            wst_io[i] = o8;
        }
        d2_out[i] = (float)bd;
        if (d64_out) d64_out[i] = bd;                        // (target-sharded ranks compare shards in f64)
    };
    // the runner-up half of the state: u's f64 point, its index | LB3 << 32 (all bits set: none) -- written through to
    // global memory whenever it changes (a search, a change of places); LB3 also to the persistent launch's slot
    auto emit_ru = [&](long long i, unsigned ht, bool have, double ux, double uy, double uz, unsigned uidx, float lb3) {
        if constexpr (kCoopRu) {
            if (!ru_io) return;
            Pt64 o8;
            o8.x = o8.y = o8.z = __longlong_as_double(-1ll);
            o8.w = ~0ull;
            if (have) { o8.x = ux; o8.y = uy; o8.z = uz; o8.w = (unsigned long long)uidx | ((unsigned long long)__float_as_uint(lb3) << 32); }
            ru_io[i] = o8;
            if constexpr (PERSIST) res.lb3[ht] = have ? lb3 : __uint_as_float(0xFFFFFFFFu);
        }
    };
    // the transformed query of home thread h (f64)
    auto query_p = [&](unsigned h, double (&out)[3]) { out[0] = s_p64[h][0]; out[1] = s_p64[h][1]; out[2] = s_p64[h][2]; };
    // the Jacobian / residual moments of one correspondence (p, q), on the query's HOME lane
    auto moments = [&](double pxd, double pyd, double pzd, unsigned bidx, double qx, double qy, double qz) {
        double nx = 0.0, ny = 0.0, nz = 0.0;
        if (PLANE) {
            if (nrm64) { const Pt64 n8 = nrm64[bidx]; nx = n8.x; ny = n8.y; nz = n8.z; }
            else { const float4 n4 = nrm[bidx]; nx = n4.x; ny = n4.y; nz = n4.z; }
        }
        accumulate_pq_d<PLANE>(acc, pxd, pyd, pzd, qx, qy, qz, nx, ny, nz, off);
    };
    // ---- The slot ranges of the query at (px, py, pz) that can hold a candidate at or within the squared fp32 distance
    // bound0: xb[k] .. xe[k] of row k = (dy, dz) of the 3 x 3 x 3 cells around it, and lbgeo2, a lower bound (squared) of
    // the distance to every target point that lies in NONE of those ranges.  `want`: this lane looks its rows up at all.
    auto list_rows = [&](const float px, const float py, const float pz, const float E, const float bound0, const bool want,
                         unsigned (&xb)[9], unsigned (&xe)[9], float &lbgeo2, unsigned &looked) {
        const int cx = cell_coord(px, g.mn[0], g.inv_h, g.dim[0]);
        const int cy = cell_coord(py, g.mn[1], g.inv_hs, g.dim[1]);
        const int cz = cell_coord(pz, g.mn[2], g.inv_hs, g.dim[2]);
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dim[0] - 1);
        const int span = x1 + 1 - x0;                        // cells of a row that exist: <= 0, 1, 2 or 3
        // ---- starts of the cells x0 .. x0+3 of a row: one 16-byte load (absent row / lane: zeros = empty)
        typedef unsigned u4a __attribute__((ext_vector_type(4), aligned(4)));
        // (cell indices fit 32 bits: kGridMaxCells; which of the three y / z rows exist is tested once per axis)
        const bool yok[3] = {cy - 1 >= 0 && cy - 1 < g.dim[1], cy >= 0 && cy < g.dim[1], cy + 1 >= 0 && cy + 1 < g.dim[1]};
        const bool zok[3] = {cz - 1 >= 0 && cz - 1 < g.dim[2], cz >= 0 && cz < g.dim[2], cz + 1 >= 0 && cz + 1 < g.dim[2]};
        const int row_c = (cz * g.dim[1] + cy) * g.dim[0] + x0, pitch_y = g.dim[0], pitch_z = g.dim[1] * g.dim[0];
        auto load_row = [&](int k, bool wanted) {
            const bool ok = wanted && span > 0 && zok[k / 3] && yok[k % 3];
            u4a v = {0u, 0u, 0u, 0u};
            if (ok) v = *reinterpret_cast<const u4a *>(reinterpret_cast<const char *>(start) + (unsigned)(row_c + (k / 3 - 1) * pitch_z + (k % 3 - 1) * pitch_y) * 4u);
            return v;
        };
        // slab distances (in cells) of the neighbouring rows / cells, margins as in nn_grid_reduce_kernel
        const float fx = (px - g.mn[0]) * g.inv_h - (float)cx;
        const float fy = (py - g.mn[1]) * g.inv_hs - (float)cy;
        const float fz = (pz - g.mn[2]) * g.inv_hs - (float)cz;
        const float mgn = 1e-3f + 4.0f * E * g.inv_hs;
        const float lo_x = fmaxf(fx - mgn, 0.f), hi_x = fmaxf(1.0f - fx - mgn, 0.f);
        const float lo_y = fmaxf(fy - mgn, 0.f), hi_y = fmaxf(1.0f - fy - mgn, 0.f);
        const float lo_z = fmaxf(fz - mgn, 0.f), hi_z = fmaxf(1.0f - fz - mgn, 0.f);
        const float h2 = g.hs * g.hs * (1.0f - 1e-5f);
        // LB, geometric part: what everything outside the 27 cells is at least away (squared): one cell plus the
        // nearest face of the own cell; a query far outside the grid's bounding box (clamped cell coordinate): its
        // distance to the box, in cells (same fp32 binning, same margin).  The cells NOT listed: pruned2 below.
        float out2;
        {
            const float face = fminf(fminf(fminf(lo_x, hi_x), fminf(lo_y, hi_y)), fminf(lo_z, hi_z));
            const float ux = fx + (float)cx, uy = fy + (float)cy, uz = fz + (float)cz;
            const float far = fmaxf(fmaxf(fmaxf(-ux, ux - (float)g.dim[0]), fmaxf(-uy, uy - (float)g.dim[1])),
                                    fmaxf(-uz, uz - (float)g.dim[2])) - 2.0f * mgn;
            const float outc = fmaxf(1.0f + face, far);
            out2 = outc * outc * h2;
        }
        float exq[3];                                        // (a cell past the span: +inf -- never listed, bounds nothing)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int xi = x0 + j;
            const float ex = xi == cx ? 0.f : (xi < cx ? lo_x + (float)(cx - xi - 1) : hi_x + (float)(xi - cx - 1));
            exq[j] = j < span ? ex * ex * h2 : INFINITY;
        }
        const float ey2[3] = {lo_y * lo_y, 0.f, hi_y * hi_y}, ez2[3] = {lo_z * lo_z, 0.f, hi_z * hi_z};
        float rowB[9];                                       // squared slab bound of row k = (dy, dz); a row that does not exist: +inf
#pragma unroll
        for (int k = 0; k < 9; k++) rowB[k] = (zok[k / 3] && yok[k % 3]) ? (ey2[k % 3] + ez2[k / 3]) * h2 : INFINITY;
        // ---- only the rows whose slab can hold a point at or within bound0 are looked up at all (a converged
        // pass needs 1.9 of the 9 per query), all of them in ONE round of gathers: nothing depends on a load
        // from here to the candidates
        u4a cs[9];
        looked = 0u;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const bool w = want && !(rowB[k] > bound0);
            cs[k] = load_row(k, w);
            looked += (w && span > 0) ? 1u : 0u;             // (profiling)
        }
        // ---- the slots of each row that can hold a candidate at or within bound0: cells whose slab bound
        // (y, z and x slab distances, squared) does not exceed it.  A skipped cell holds nothing that could
        // win or tie (margins: fp32 binning + rounding band, as in the lane-serial kernel).
        // (B + exq[j] <= bound0 tested as B <= bound0 - exq[j]: the slab bounds carry a relative margin of ~1e-3, a
        //  rounding of the subtraction cannot drop a cell that matters; a cell that does not exist: -inf)
        const float lim0 = bound0 - exq[0];                  // (-inf for a cell past the span)
        const float lim1 = bound0 - exq[1];
        const float lim2 = bound0 - exq[2];
        float pruned2 = INFINITY;                            // smallest slab bound (squared) among the cells NOT listed
        const float thr0 = bound0 * (1.0f - 1e-6f);
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const float B = rowB[k];
            const bool n0 = B <= lim0, n1 = B <= lim1, n2 = B <= lim2;
            // (a cell is NOT listed iff B > fl(bound0 - exq[j]), which implies fl(B + exq[j]) > bound0 (1 - 2e-7): every
            //  such cell is in this minimum; a listed cell within 1e-6 of bound0 may be too -- its slab bound holds as well.
            //  Own comparisons, so that the 27 listing masks need not stay live)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const float S = B + exq[j];
                pruned2 = fminf(pruned2, S >= thr0 ? S : INFINITY);
            }
            const u4a v = cs[k];                             // (a row not looked up reads as zeros: empty)
            const unsigned b = n0 ? v.x : (n1 ? v.y : v.z);
            unsigned e = n2 ? v.w : (n1 ? v.z : (n0 ? v.y : b));
            if (!(n0 || n1 || n2)) e = b;
            xb[k] = b;
            xe[k] = e;
        }
        lbgeo2 = fminf(out2, pruned2);
    };
    // ---- One ROUND = one query per lane of the workgroup.
    //  A  every lane, its own ("home") query: source point + state in, the certificate; a certified query writes its
    //     outputs here, the others queue up (per-wave segments: the queue's order does not depend on arrival).
    //  B  the queued queries, COMPACTED over the threads of the workgroup: thread k searches the k-th queued query
    //     (whoever's it is) -- waves k/64 beyond the queue's length list nothing, so the search costs what the
    //     uncertified queries cost, not what the waves that hold one cost (after a few ICP iterations most waves hold
    //     a handful: without the compaction every one of them ran the whole search for those).  The searchers' chunks
    //     go to ONE list of the workgroup, which ALL its waves work off -- the waves without a searcher too: the
    //     gathers of a workgroup with 30 queued queries are one trip of all its lanes, not three of one wave's.
    //     A query's result does not depend on the thread that searched it or the lanes that ranked its chunks.
    //  C  home lanes again: the moments of the round's correspondence (from registers when certified, from the
    //     searcher's hand-over in LDS otherwise) -- the query -> lane map and the summation tree of the lane-serial
    //     kernel (for NTH = kBlock), so the 38 statistics stay bit-identical to its.
    // a searched query's runner-up on its way from the f64 array to the state (phase C -> behind the partial row)
    Pt64 ru_pending = Pt64{0.0, 0.0, 0.0, 0ull};
    float ru_lb3 = 0.f;
    bool ru_have = false, ru_go = false;
    int ru_it = 0;                                           // (the round: uniform; the query's index is re-derived -- kept
                                                             //  across the search it was two more registers to spill)
    auto ru_finish = [&]() {
        if constexpr (kCoopRu) {
            if (ru_go)
                emit_ru((long long)(vb * NTH + tid) * per_group + ru_it, (unsigned)tid, ru_have, ru_pending.x, ru_pending.y,
                        ru_pending.z, (unsigned)ru_pending.w, ru_lb3);
            ru_go = false;
        }
    };
    // (experiment, persistent launches: warm bits 32 / 64 = VISMA_ICP_PERSIST_PRIO 1 / 2.  The arbiter serves the oldest wave
    //  of a SIMD first and the pass waits for the youngest residency slot, DESIGN 4.1e: 1 = priorities reversed for the whole
    //  body -- youngest slot highest --, 2 = reversed in every other phase)
    auto set_phase_prio = [&](const int phase) {
        if constexpr (PERSIST) {
            const int mode = (warm >> 5) & 3;
            if (mode != 0) {
                const int slot = (int)((blockIdx.x * 4u) / gridDim.x);
                const int p = mode == 1 ? slot : ((phase & 1) ? slot : 3 - slot);
                if (p == 0) __builtin_amdgcn_s_setprio(0);
                else if (p == 1) __builtin_amdgcn_s_setprio(1);
                else if (p == 2) __builtin_amdgcn_s_setprio(2);
                else __builtin_amdgcn_s_setprio(3);
            }
        }
    };
    auto round = [&](const int it) {
        set_phase_prio(0);
        const long long i = i_begin + it;
        const bool active = i < i_end;
        // ---- A: the query: the reference's transform of a source point (PointCloud.cpp:75-80), in f64
        Pt64 s8 = Pt64{0.0, 0.0, 0.0, 0ull};
        // the state the previous pass left: its winner's f64 point (NaN = none), original index | LB << 32
        // (all bits set = nothing known: NaN coordinates, NaN bound)
        Pt64 w8;
        w8.x = w8.y = w8.z = __longlong_as_double(-1ll);
        w8.w = ~0ull;
        if (active) {
            if constexpr (PERSIST) {
                if (res.have_src) { s8.x = res.sx; s8.y = res.sy; s8.z = res.sz; }
                else s8 = src64[i];
            } else {
                s8 = src64[i];
            }
            if constexpr (PERSIST) {
                if (res.first) {
                    if (warm & 1) w8 = wst_io[i];
                } else {
                    // the state the pass before left in this lane's slots
                    w8.x = s_q64[tid][0]; w8.y = s_q64[tid][1]; w8.z = s_q64[tid][2];
                    w8.w = (unsigned long long)__float_as_uint(s_dprev[tid]) | ((unsigned long long)__float_as_uint(res.lb[tid]) << 32);
                }
            } else {
                if (warm & 1) w8 = wst_io[i];
            }
        }
        // (se3_act: the restatement of SE3Type's action on a point, core/se3.h:103-106 -- same products, same order)
        double hp[3];
        {
            const double sv[3] = {s8.x, s8.y, s8.z};
            se3_act(T64.m, sv, hp);
        }
        // (the runner-up of the state: asked for now, with the source point -- 32 B per query from a warm L2 -- so that a
        //  failed winner-unchanged certificate does not wait for it)
        Pt64 u8;
        u8.x = u8.y = u8.z = __longlong_as_double(-1ll);
        u8.w = ~0ull;
        if constexpr (kCoopRu) {
            if (ru_io && active && (warm & 4)) u8 = ru_io[i];
        }
        // ---- the certificate: has the query moved by less than the room its previous result left?
        unsigned widx = (unsigned)w8.w;
        bool has_w = w8.x == w8.x;                           // (NaN: no previous winner)
        bool cert = false;
        if (warm & 4) {
            const float hx = (float)hp[0], hy = (float)hp[1], hz = (float)hp[2];
            const float rup = sqrtf(r2f) * (1.0f + 2.4e-7f);
            const float E = 2.4e-7f * (fabsf(hx) + fabsf(hy) + fabsf(hz) + rup) + 4.8e-7f * rup;
            const float tlim = rup + 2.0f * E;
            // delta = |T s - T_prev s|: same products, same order as the transform itself, so the two computed
            // points are the ones the bounds speak about; rounded up
            double pp[3];
            bool pp_known = false;
            if constexpr (PERSIST) {
                // (round 6) inside a persistent launch the pass before left exactly that point in this lane's slot -- the
                // same products in the same order, from the transform that is Tprev now: read back instead of 18 more
                // f64 operations and Tprev's 24 scalar registers live through phase A
#ifndef VISMA_PERSIST_PP_LDS
#define VISMA_PERSIST_PP_LDS 1
#endif
                if (VISMA_PERSIST_PP_LDS && !res.first) { pp[0] = s_p64[tid][0]; pp[1] = s_p64[tid][1]; pp[2] = s_p64[tid][2]; pp_known = true; }
            }
            if (!pp_known) {
                const double sv[3] = {s8.x, s8.y, s8.z};
                se3_act(Tprev.m, sv, pp);
            }
            const double ex = hp[0] - pp[0], ey = hp[1] - pp[1], ez = hp[2] - pp[2];
            const float del = sqrtf((float)(ex * ex + ey * ey + ez * ez)) * (1.0f + 1e-6f);
            // (the bound of an unknown state is NaN: every comparison below is false.  The two ulps taken off cover the
            //  rounding of this subtraction and the ~1e-16 |p| by which the f64 difference above can be off -- the band
            //  E itself was taken off once, when the bound was formed: taking it off every pass would wear a 1 mm gap
            //  down in a few hundred certified passes and send the query back to the search for nothing)
            const float lb_old = __uint_as_float((unsigned)(w8.w >> 32));
            const float t = (lb_old - del) * (1.0f - 2.4e-7f);
            double d2w = r2d;                                // the previous winner's distance now (reference arithmetic)
            if (has_w) {
                // flann L2 (dist.h:159-176), as rank() below
                const double dx = w8.x - hp[0], dy = w8.y - hp[1], dz = w8.z - hp[2];
                double d = dx * dx;
                d += dy * dy;
                d += dz * dz;
                d2w = d;
                cert = active && t > 0.f && d < r2d && d < (double)t * (double)t * (1.0 - 1e-6);
            } else {
                cert = active && widx == 0xFFFFFFFFu && t > tlim * (1.0f + 1e-6f);
            }
            if (cert) emit(i, (unsigned)tid, true, has_w ? d2w : r2d, widx, has_w, w8.x, w8.y, w8.z, t);
            if constexpr (kCoopRu) {
                // ---- the runner-up's turn.  LB3 bounds every target point but w and u as the query stood; it moved by
                // del, so everything else is still at least t3 away.  (lb_old > 0: the state was left by a search of this
                // kernel or by a certificate -- the lane-serial and wave kernels write LB = 0 and no runner-up.)
                const bool has_u = u8.x == u8.x;
                float lb3 = __uint_as_float((unsigned)(u8.w >> 32));
                if constexpr (PERSIST) { if (!res.first) lb3 = res.lb3[tid]; }
                const float t3 = (lb3 - del) * (1.0f - 2.4e-7f);
                if (ru_io && active && has_w && has_u) {
                    if (cert) {
                        // (winner unchanged: LB3 moves like LB)
                        if constexpr (PERSIST) res.lb3[tid] = t3;
                        else reinterpret_cast<unsigned *>(&ru_io[i].w)[1] = __float_as_uint(t3);
                    } else if (lb_old > 0.f && t3 > 0.f) {
                        const unsigned uidx = (unsigned)u8.w;
                        const double dx = u8.x - hp[0], dy = u8.y - hp[1], dz = u8.z - hp[2];
                        double d2u = dx * dx;                // flann L2 (dist.h:159-176), as for the winner
                        d2u += dy * dy;
                        d2u += dz * dz;
                        const bool u_wins = d2u < d2w || (d2u == d2w && uidx < widx);   // (lowest original index on a tie: rank())
                        const double dm = u_wins ? d2u : d2w, dother = u_wins ? d2w : d2u;
                        if (dm < r2d && dm < (double)t3 * (double)t3 * (1.0 - 1e-6)) {
                            // the nearer of the two is the unique nearest neighbour; everything but it is at least
                            // min(the other's distance, t3) away
                            cert = true;
                            const float lbn = fminf(t3, sqrtf((float)dother) * (1.0f - 1e-6f));
                            if (u_wins) {
                                // they change places
                                const Pt64 ow = w8;
                                w8.x = u8.x; w8.y = u8.y; w8.z = u8.z;
                                w8.w = (unsigned long long)uidx | ((unsigned long long)__float_as_uint(lbn) << 32);
                                widx = uidx;
                                emit_ru(i, (unsigned)tid, true, ow.x, ow.y, ow.z, (unsigned)ow.w, t3);
                                if constexpr (!PERSIST) wst_io[i] = w8;
                            } else {
                                if constexpr (PERSIST) res.lb3[tid] = t3;
                                else reinterpret_cast<unsigned *>(&ru_io[i].w)[1] = __float_as_uint(t3);
                            }
                            emit(i, (unsigned)tid, true, dm, widx, true, w8.x, w8.y, w8.z, lbn);
                        }
                    }
                } else if (ru_io && active && cert) {
                    if constexpr (PERSIST) res.lb3[tid] = __uint_as_float(0xFFFFFFFFu);
                }
            }
        }
        const bool need = active && !cert;
        const bool cfound = cert && has_w;                   // certified WITH a partner: (hp, w8) is the correspondence
        // the transformed query, for its searcher (phase B) and for this lane itself in phase C
        s_p64[tid][0] = hp[0]; s_p64[tid][1] = hp[1]; s_p64[tid][2] = hp[2];
        if (need) {
            // (fp32 view of the previous winner: the value the candidate array holds for that point; NaN = none)
            const float dp = sqdist_f32(make_float4((float)w8.x, (float)w8.y, (float)w8.z, 0.f), (float)hp[0], (float)hp[1], (float)hp[2]);
            // ROOM for the certificates to come: cells are listed out to the previous winner's distance PLUS a margin
            // (any bound at or above that distance is a valid one).  Listing exactly to the winner's distance leaves
            // the cells just beyond it un-examined, and their slab bounds -- which can lie a hair above the winner's
            // distance -- then cap the LB this search leaves behind: one query in eight failed its certificate for
            // that reason alone, pass after pass.  The margin costs the candidates of a thin shell.
            const float sq = sqrtf(dp) + kCoopRoom * sqrtf(r2f);
            s_dprev[tid] = sq * sq;                          // (NaN stays NaN)
        } else {
            // certified (or no query): the correspondence of phase C is known now
            s_q64[tid][0] = w8.x; s_q64[tid][1] = w8.y; s_q64[tid][2] = w8.z;
            s_dprev[tid] = __uint_as_float(cfound ? widx : 0xFFFFFFFFu);
        }
        if (cand_count) {
            // (profiling) queries certified: no search
            const unsigned cc = (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(cert));
            if (lane == 0 && cc) atomicAdd(cand_count + 2 * 4096 + (blockIdx.x & 4095), (unsigned long long)cc);
        }
        // ---- queue the others: position = queries queued by the lower waves + by the lower lanes of this wave
        const unsigned long long needers = __builtin_amdgcn_ballot_w64(need);
        const unsigned rank = (unsigned)__builtin_popcountll(needers & ((1ull << lane) - 1ull));
        if (need) s_queue[wave][rank] = (unsigned short)tid;
        if (lane == 0) s_need[wave] = (unsigned)__builtin_popcountll(needers);
        COOP_MARK(9);                                        // phase A done
        __syncthreads();
        COOP_MARK(10);
        set_phase_prio(1);
        unsigned nq_all = 0;                                 // (uniform values: scalar registers)
        unsigned home = 0;                                   // the home thread of the query this thread searches
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const unsigned c = (unsigned)__builtin_amdgcn_readfirstlane((int)s_need[w]);
            if ((unsigned)tid >= nq_all && (unsigned)tid < nq_all + c) home = s_queue[w][(unsigned)tid - nq_all];
            nq_all += c;
        }
        // ---- B: thread k < nq_all searches the k-th queued query
        const bool searching = (unsigned)(wave * 64) < nq_all;           // (wave-uniform)
        const bool sact = (unsigned)tid < nq_all;                        // this thread has a query to search
        if (nq_all != 0u) {                                  // (workgroup-uniform: the stages below hold barriers)
            // -- B1 (searchers): the query taken over, its rows' slot ranges, its chunk count
            unsigned xb[9], xe[9];
            float lbgeo2 = 0.f;                              // LB, geometric part (squared)
            unsigned nq = 0, off_q = 0;                      // this query's chunks; where its run begins in its wave's part
            if (searching) {
                s_home[tid] = (unsigned short)home;          // (read back at the end: nothing is carried across the search)
                // the query as its home lane transformed it (phase A), and its distance to the previous winner
                double hq[3];
                query_p(home, hq);
                const double pxd = hq[0], pyd = hq[1], pzd = hq[2];
                const float dprev = sact ? s_dprev[home] : __uint_as_float(~0u);
                const float px = (float)pxd, py = (float)pyd, pz = (float)pzd;
                COOP_MARK(0);                                // the query taken over from its home lane
                // rounding band (exact_band): E bounds |d64 - sqrt(d2_32)|; L = squared fp32 distance at or beyond
                // which a candidate cannot be accepted in f64; W >= band(m) - m for every m < L
                const float rup = sqrtf(r2f) * (1.0f + 2.4e-7f);
                const float E = 2.4e-7f * (fabsf(px) + fabsf(py) + fabsf(pz) + rup) + 4.8e-7f * rup;
                const float tlim = rup + 2.0f * E;
                const float L = tlim * tlim * (1.0f + 6e-7f);
                const float W = (4.0f * E * tlim + 4.0f * E * E) * (1.0f + 1e-6f) + L * 5e-7f;
                s_qp[tid] = make_float4(px, py, pz, W);
                // what nothing nearer than can be missed by: the previous winner's distance now, or the radius
                float bound0 = L;
                if (dprev < L) bound0 = dprev;               // (NaN = no previous winner: the radius)
                unsigned looked;
                list_rows(px, py, pz, E, bound0, sact, xb, xe, lbgeo2, looked);
                // (pin the value here: left to itself the compiler sinks the whole slab arithmetic down to the bound's
                //  only use at the end of the query and spills its 40 inputs across the chunk phase instead)
                asm volatile("" : "+v"(lbgeo2));
                if (cand_count) {
                    // profiling only (one uniform branch): candidates listed, cell-table rows looked up -- summed over
                    // the wave and added to the launch's counters right here (counters carried to the end of the
                    // kernel were spilled across the search)
                    unsigned long long c = 0, ca = looked;
#pragma unroll
                    for (int k = 0; k < 9; k++) c += xe[k] - xb[k];
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        c += __shfl_down(c, o, 64);
                        ca += __shfl_down(ca, o, 64);
                    }
#ifndef VISMA_COOP_LB_PROBE
                    if (lane == 0 && ca) {
                        unsigned long long *slot = cand_count + 2 * (blockIdx.x & 4095);
                        atomicAdd(slot, c);
                        atomicAdd(slot + 1, ca);
                    }
#endif
                }
                COOP_MARK(1);                                // row bounds asked for, rows pruned
#pragma unroll
                for (int k = 0; k < 9; k++) nq += (xe[k] - xb[k] + 7u) >> 3;
                const unsigned incl = wave_scan_incl(nq, lane);
                off_q = incl - nq;
                if (lane == 63) s_m[wave] = incl;
            } else {
                if (lane == 0) s_m[wave] = 0u;
            }
            __syncthreads();
            unsigned m_all = 0;                              // chunks of the workgroup (uniform)
#pragma unroll
            for (int w = 0; w < NW; w++) {
                const unsigned c = (unsigned)__builtin_amdgcn_readfirstlane((int)s_m[w]);
                if (w < wave) off_q += c;                    // ... in the workgroup's list
                m_all += c;
            }
            if (work_out) *work_out = nq_all | (m_all << 12);    // (measurement: queued queries, chunks listed)
            // the two best chunks (minimum, first slot, flag byte) and the third chunk minimum
            float gh0 = INFINITY, gh1 = INFINITY, gh2 = INFINITY;
            float gsec = INFINITY;                           // best candidate outside the rounding band of its chunk's minimum
            unsigned gb0 = 0xFFFFFFFFu, gb1 = 0xFFFFFFFFu, gm0 = 0u, gm1 = 0u;
            unsigned gs0 = 0x7F800000u | (0x1FE0u << 3);     // the side word of the best chunk (sec = third = +inf)
            auto chunk_insert = [&](float m, unsigned b, unsigned flags, unsigned side) {
                const bool c1 = m < gh0, c2 = m < gh1;
                gh2 = __builtin_amdgcn_fmed3f(gh1, gh2, m);
                gh1 = __builtin_amdgcn_fmed3f(gh0, gh1, m);
                gh0 = fminf(gh0, m);
                gb1 = c2 ? b : gb1; gm1 = c2 ? flags : gm1;
                gb1 = c1 ? gb0 : gb1; gm1 = c1 ? gm0 : gm1;
                gb0 = c1 ? b : gb0; gm0 = c1 ? flags : gm0;
                if constexpr (kCoopRu) gs0 = c1 ? side : gs0;
            };
            // -- B2: the list, one window at a time (one window unless the cloud is very dense).  A query's chunks take
            // CONSECUTIVE entries, row after row: the owner reads its results back as one short run.
            for (unsigned w0 = 0; w0 < m_all; w0 += kCapAll) {
                if (searching) {
                    // descriptor: (the chunk's first slot, owner's query in s_qp | candidates << 16)
                    const unsigned own = (unsigned)tid << 4;
                    unsigned j = off_q - w0;                 // (a run that begins before the window wraps: never < cap)
#pragma unroll
                    for (int k = 0; k < 9; k++) {
                        unsigned b = xb[k];
                        while (b < xe[k]) {
                            if (j < kCapAll) s_item[j] = make_uint2(b, own | (min(xe[k] - b, 8u) << 16));
                            j++;
                            b += 8u;
                        }
                    }
                    COOP_MARK(2);                            // chunk list written
                }
                __syncthreads();
                set_phase_prio(2);
                // ---- ALL waves: the window's chunks, eight lanes per chunk, one candidate per lane, kCoopDepth chunks
                // per lane octet in flight (every load of the list is independent)
                const unsigned mw = min(m_all - w0, kCapAll);
                for (unsigned tb = 0; tb < mw; tb += (unsigned)(NW * 8 * kCoopDepth)) {            // (uniform trip count)
                    const unsigned t = tb + (unsigned)(wave * 8 + oct);
                    P12 c4[kCoopDepth];
                    unsigned meta[kCoopDepth];
#pragma unroll
                    for (int u = 0; u < kCoopDepth; u++) {
                        const unsigned c = t + (unsigned)(u * NW * 8);
                        uint2 dsc = make_uint2(0u, 0u);      // (past the window's end: a null descriptor, count 0)
                        if (c < mw) dsc = s_item[c];
                        meta[u] = dsc.y;
                        // scalar base + 32-bit byte offset (the launcher keeps 12 * slots below 2^32; the array carries
                        // kSortedSlack entries of slack)
                        c4[u] = *reinterpret_cast<const P12 *>(reinterpret_cast<const char *>(s12) + (((dsc.x * 3u) << 2) + (unsigned)l8 * 12u));
                    }
#pragma unroll
                    for (int u = 0; u < kCoopDepth; u++) {
                        const unsigned c = t + (unsigned)(u * NW * 8);
                        const unsigned cnt = meta[u] >> 16;
                        const float4 p = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_qp) + (meta[u] & 0xFFF0u));
                        float d = sqdist_f32(make_float4(c4[u].x, c4[u].y, c4[u].z, 0.f), p.x, p.y, p.z);
                        const bool mine = (unsigned)l8 < cnt;            // (lane 0 of the octet: the chunk exists)
                        d = mine ? d : INFINITY;
                        const float m = octet_min(d);
                        // (a lane past the chunk's end holds +inf: never within m + W of a finite minimum; a null
                        //  descriptor's result is not stored)
                        const bool fl = d <= m + p.w;
                        const unsigned long long bal = __builtin_amdgcn_ballot_w64(fl);
                        const unsigned flags = (unsigned)(bal >> (oct * 8)) & 0xFFu;
                        // the chunk's best candidate that is NOT flagged (for the LB the query leaves behind), its lane, and
                        // the best of the rest (the runner-up's chunk: what bounds everything but winner and runner-up)
                        const float dn = fl ? INFINITY : d;
                        const float sec = octet_min(dn);
                        unsigned secl = 0u;
                        float third = INFINITY;
                        if constexpr (kCoopRu) {
                            const unsigned long long bs = __builtin_amdgcn_ballot_w64(dn == sec && sec < INFINITY);
                            const unsigned b8 = (unsigned)(bs >> (oct * 8)) & 0xFFu;
                            secl = b8 ? (unsigned)__builtin_ctz(b8) : 0u;
                            third = octet_min((b8 != 0u && (unsigned)l8 == secl) ? INFINITY : dn);
                        }
                        // the result takes the place of the descriptor's second word: the chunk minimum rounded DOWN to
                        // 16 mantissa bits | the flag byte (the first word, the chunk's position, stays)
                        if (l8 == 0 && mine) {
                            s_item[c].y = (__float_as_uint(m) & 0xFFFFFF00u) | flags;
                            s_sec[c] = pack_sec(sec, third, secl);
                        }
                    }
                }
                if (searching) COOP_MARK(3);                 // chunks worked off (this wave's share)
                __syncthreads();
                set_phase_prio(3);
                if (searching) {
                    // the owner's run of results, four reads in flight (a lane past its run inserts +inf: no effect)
                    for (unsigned c0 = 0; __builtin_amdgcn_ballot_w64(c0 < nq) != 0ull; c0 += 4u) {
                        uint2 r[4];
                        unsigned rs[4];
                        bool in[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const unsigned j = off_q - w0 + c0 + (unsigned)u;
                            in[u] = c0 + (unsigned)u < nq && j < kCapAll;
                            r[u] = s_item[in[u] ? j : 0u];
                            rs[u] = s_sec[in[u] ? j : 0u];
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            chunk_insert(in[u] ? __uint_as_float(r[u].y & 0xFFFFFF00u) : INFINITY, r[u].x, r[u].y & 0xFFu, rs[u]);
                            gsec = fminf(gsec, in[u] ? sec_of(rs[u]) : INFINITY);
                        }
                    }
                    COOP_MARK(4);                            // chunk results merged per query
                }
                if (w0 + kCapAll < m_all) __syncthreads();   // (the list is re-used by the next window)
            }
            // (runner-up builds: the searchers hand slot and bound over through the chunk list's memory -- not before every
            //  owner has read its last results from it)
            if constexpr (kCoopRu) __syncthreads();
            // -- B3 (searchers): the f64 decision -- flagged candidates of the kept chunks inside g + W.
            // The query itself comes back from where it lies in LDS -- its f64 point from its home lane's slot, its
            // fp32 view and W from the searcher's, the band re-derived -- instead of occupying 17 registers across the
            // listing and the chunk phase, where the kernel sits at the 128 it may use (4 waves per SIMD).
            if (searching) {
                const bool need = sact, active = sact;
                float lb_new = 0.f;                          // what this pass leaves as LB
                double bd = r2d;                             // best d2 so far (strictly below r2d once set)
                unsigned bidx = 0xFFFFFFFFu, bpos = 0xFFFFFFFFu;
                Pt64 bq = Pt64{0.0, 0.0, 0.0, 0ull};
                auto tail = [&](const double pxd, const double pyd, const double pzd, const float px, const float py, const float pz,
                                const float W) {
                const float rup = sqrtf(r2f) * (1.0f + 2.4e-7f);
                const float E = 2.4e-7f * (fabsf(px) + fabsf(py) + fabsf(pz) + rup) + 4.8e-7f * rup;
                const float tlim = rup + 2.0f * E;
                const float L = tlim * tlim * (1.0f + 6e-7f);
                auto rank = [&](const Pt64 &c8, unsigned pos) {
                    // flann L2 (dist.h:159-176): result += diff * diff over x, y, z
                    const double dx = c8.x - pxd, dy = c8.y - pyd, dz = c8.z - pzd;
                    double d = dx * dx;
                    d += dy * dy;
                    d += dz * dz;
                    const unsigned id = (unsigned)c8.w;
                    const bool lt = d < bd || (d == bd && id < bidx && bidx != 0xFFFFFFFFu);
                    bd = lt ? d : bd;
                    bidx = lt ? id : bidx;
                    bpos = lt ? pos : bpos;
                    bq.x = lt ? c8.x : bq.x; bq.y = lt ? c8.y : bq.y; bq.z = lt ? c8.z : bq.z; bq.w = lt ? c8.w : bq.w;
                };
                bool slow = false;                                   // needs every listed candidate ranked in f64
                // (the kept chunk minima were rounded down by < 2^-15 relative: g_up bounds the fp32 minimum from above, and
                //  the tests below stay on the safe side -- a chunk or candidate more is ranked in f64, never one less)
                const float g_up = gh0 * (1.0f + 6.2e-5f);
                int n = 0;                                           // candidates ranked in f64
                // the runner-up the state will keep (kCoopRu): its slot, the squared bound of everything but it and the winner
                unsigned rupos = 0xFFFFFFFFu;
                float lb3_2 = 0.f;
                // (chunk minima at or beyond L cannot be accepted in f64: such chunks only serve the LB)
                if (need && gh0 < L) {
                    const float thr = g_up + W;
                    unsigned c[4] = {0u, 0u, 0u, 0u};
                    slow = gh2 <= thr;                               // a third chunk reaches into the band
                    auto add = [&](unsigned b, unsigned flags) {
                        while (flags) {
                            const unsigned pos = b + (unsigned)__builtin_ctz(flags);
                            flags &= flags - 1u;
                            if (n == 0) c[0] = pos; else if (n == 1) c[1] = pos; else if (n == 2) c[2] = pos; else if (n == 3) c[3] = pos;
                            else slow = true;
                            n++;
                        }
                    };
                    add(gb0, gm0);
                    if (gh1 <= thr) add(gb1, gm1);
                    if constexpr (kCoopRu) {
                        // One flagged candidate in all (the usual case): it is THE candidate of the best chunk.  The
                        // runner-up is the best of (A) that chunk's best candidate outside the band -- then every other
                        // chunk is at least gh1 away and the rest of this one at least `third` -- and (B) the second-best
                        // chunk's minimum, if that chunk flagged one candidate only -- then the rest of every chunk is at
                        // least gsec away and every other chunk gh2.  (All values rounded down: lower bounds.)
                        if (ru_io && n == 1 && !slow) {
                            const float secA = sec_of(gs0);
                            if (secA < INFINITY && secA <= gh1) {
                                rupos = gb0 + (gs0 & 7u);
                                lb3_2 = fminf(gh1, third_of(gs0));
                            } else if (gh1 < INFINITY && __builtin_popcount(gm1) == 1) {
                                rupos = gb1 + (unsigned)__builtin_ctz(gm1);
                                lb3_2 = fminf(gh2, gsec);
                            }
                            lb3_2 = fminf(lb3_2, lbgeo2);
                        }
                    }
                    // (a second flagged candidate: one query in a thousand; a third: duplicated points)
                    Pt64 c8a = Pt64{0.0, 0.0, 0.0, 0ull}, c8b = c8a;
                    if (n > 0) c8a = sorted64[c[0]];
                    if (n > 1) c8b = sorted64[c[1]];
                    if (n > 0) rank(c8a, c[0]);
                    if (n > 1) rank(c8b, c[1]);
                    if (n > 2) rank(sorted64[c[2]], c[2]);
                    if (n > 3) rank(sorted64[c[3]], c[3]);
                }
                // ---- the re-scan, by the WHOLE WAVE for one such query at a time (a few per launch at C4, and the launch
                // lasts as long as its slowest wave: one lane walking its 27 cells alone -- ~90 dependent loads -- put 6 us
                // on the tail of every launch).  The query's listed slot ranges (everything that can win or tie lies in
                // them, see the pruning above) are flattened over the lanes: one fp32 filter load, one f64 load, a
                // butterfly over (d2, original index), the winner's coordinates handed to the owner lane.
                for (unsigned long long rem = __builtin_amdgcn_ballot_w64(slow); rem; rem &= rem - 1ull) {
                    const int q = (int)__builtin_ctzll(rem);         // wave-uniform
                    auto bcast_u = [&](unsigned v) { return (unsigned)__builtin_amdgcn_readlane((int)v, q); };
                    auto bcast_f = [&](float v) { return __uint_as_float(bcast_u(__float_as_uint(v))); };
                    auto bcast_d = [&](double v) {
                        const unsigned long long u = (unsigned long long)__double_as_longlong(v);
                        return __longlong_as_double((long long)(((unsigned long long)bcast_u((unsigned)(u >> 32)) << 32) | bcast_u((unsigned)u)));
                    };
                    const float qx = bcast_f(px), qy = bcast_f(py), qz = bcast_f(pz);
                    const double qxd = bcast_d(pxd), qyd = bcast_d(pyd), qzd = bcast_d(pzd);
                    const float qrup = rup, qE = bcast_f(E);
                    const float sl = fminf(sqrtf(bcast_f(g_up)), qrup) + 2.0f * qE;
                    const float Ls = sl * sl * (1.0f + 6e-7f);       // fp32 distances beyond it cannot win or tie in f64
                    // the query's slot ranges, listed again (its lane alone looks the rows up) against Ls: nothing at or within
                    // it is left out, and nothing was kept in registers for this rare path
                    unsigned qb[9], pre[10];
                    pre[0] = 0u;
                    {
                        unsigned sb[9], se[9], looked_unused;
                        float geo_unused;
                        list_rows(px, py, pz, E, Ls, lane == q, sb, se, geo_unused, looked_unused);
        #pragma unroll
                        for (int k = 0; k < 9; k++) {
                            qb[k] = bcast_u(sb[k]);
                            pre[k + 1] = pre[k] + (bcast_u(se[k]) - qb[k]);
                        }
                    }
                    double ld = r2d;
                    unsigned lid = 0xFFFFFFFFu, lpos = 0xFFFFFFFFu;
                    Pt64 lq = Pt64{0.0, 0.0, 0.0, 0ull};
                    for (unsigned f0 = 0; f0 < pre[9]; f0 += 128u) {  // (one trip unless the rows are very dense)
                        unsigned j[2];
                        bool in[2];
                        P12 t[2];
        #pragma unroll
                        for (int u = 0; u < 2; u++) {
                            const unsigned f = f0 + (unsigned)u * 64u + (unsigned)lane;
                            in[u] = f < pre[9];
                            unsigned jj = 0u;
        #pragma unroll
                            for (int k = 0; k < 9; k++)
                                if (f >= pre[k] && f < pre[k + 1]) jj = qb[k] + (f - pre[k]);
                            j[u] = jj;
                            t[u] = P12{0.f, 0.f, 0.f};
                            if (in[u]) t[u] = s12[jj];
                        }
                        Pt64 c8[2];
        #pragma unroll
                        for (int u = 0; u < 2; u++) {
                            in[u] = in[u] && sqdist_f32(make_float4(t[u].x, t[u].y, t[u].z, 0.f), qx, qy, qz) <= Ls;
                            c8[u] = Pt64{0.0, 0.0, 0.0, 0ull};
                            if (in[u]) c8[u] = sorted64[j[u]];
                        }
        #pragma unroll
                        for (int u = 0; u < 2; u++)
                            if (in[u]) {
                                // flann L2 (dist.h:159-176), as rank() above
                                const double dx = c8[u].x - qxd, dy = c8[u].y - qyd, dz = c8[u].z - qzd;
                                double d = dx * dx;
                                d += dy * dy;
                                d += dz * dz;
                                const unsigned id = (unsigned)c8[u].w;
                                const bool lt = d < ld || (d == ld && id < lid && lid != 0xFFFFFFFFu);
                                if (lt) { ld = d; lid = id; lpos = j[u]; lq = c8[u]; }
                            }
                    }
                    // minimum over the lanes by (d2, original index); lanes without a candidate hold (r2d, none)
                    double rd = ld;
                    unsigned rid = lid;
        #pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        const double od = __shfl_xor(rd, o, 64);
                        const unsigned oid = (unsigned)__shfl_xor((int)rid, o, 64);
                        const bool lt = oid != 0xFFFFFFFFu && (od < rd || (od == rd && oid < rid));
                        rd = lt ? od : rd;
                        rid = lt ? oid : rid;
                    }
                    if (rid != 0xFFFFFFFFu) {                        // (wave-uniform)
                        const unsigned long long holders = __builtin_amdgcn_ballot_w64(lid == rid && ld == rd);
                        const int wl = (int)__builtin_ctzll(holders);
                        auto from_w = [&](unsigned v) { return (unsigned)__builtin_amdgcn_readlane((int)v, wl); };
                        auto from_w64 = [&](unsigned long long u) { return ((unsigned long long)from_w((unsigned)(u >> 32)) << 32) | from_w((unsigned)u); };
                        Pt64 w8;
                        w8.x = __longlong_as_double((long long)from_w64((unsigned long long)__double_as_longlong(lq.x)));
                        w8.y = __longlong_as_double((long long)from_w64((unsigned long long)__double_as_longlong(lq.y)));
                        w8.z = __longlong_as_double((long long)from_w64((unsigned long long)__double_as_longlong(lq.z)));
                        w8.w = from_w64(lq.w);
                        const unsigned wpos = from_w(lpos);
                        if (lane == q) {
                            const bool lt = rd < bd || (rd == bd && rid < bidx && bidx != 0xFFFFFFFFu);
                            if (lt) { bd = rd; bidx = rid; bpos = wpos; bq = w8; }
                        }
                    }
                }
                // ---- LB: what every target point but the winner (every target point, without a winner) is at least away
                if (need) {
                    float lb2 = fminf(lbgeo2, fminf(gh1, gsec));
                    if (bpos == 0xFFFFFFFFu) lb2 = fminf(lb2, gh0);  // nothing accepted: the best candidate bounds like the rest
                    // (examined candidates: |d64 - sqrt(d2_32)| <= E inside the radius, <= 2E out to the corners of the 27 cells)
                    float lb = sqrtf(lb2) * (1.0f - 1e-6f) - 4.0f * E;
                    // a near-tie (several candidates ranked in f64, a re-scan): the runner-up was not bounded
                    if (slow || n > 1) lb = 0.f;
                    lb_new = fminf(fmaxf(lb, 0.f), 3.0e38f);
#ifdef VISMA_COOP_LB_PROBE
                    // (tools/lb_probe.py, measurement build) what limits the bound a search leaves: the runner-up among the
                    // examined candidates, or the cells that were not listed?  counter 0: sum over the candidate-limited
                    // queries of (geometric bound - candidate bound) in micrometres; counter 1: their number
                    if (cand_count && bpos != 0xFFFFFFFFu && !(slow || n > 1)) {
                        const float cb = fminf(gh1, gsec);
                        if (cb < lbgeo2) {
                            unsigned long long *slot = cand_count + 2 * (blockIdx.x & 4095);
                            atomicAdd(slot, (unsigned long long)((sqrtf(fminf(lbgeo2, 4.0f * r2f)) - sqrtf(cb)) * 1e6f));
                            atomicAdd(slot + 1, 1ull);
                        }
                    }
#endif
                }
                const bool found = bpos != 0xFFFFFFFFu;
                COOP_MARK(5);                                // f64 winner arrived and ranked
                if (active) {
                    const long long i = (long long)(vb * NTH + (int)s_home[tid]) * per_group + it;   // its home thread's query of the round
                    emit(i, (unsigned)s_home[tid], false, bd, bidx, found, bq.x, bq.y, bq.z, lb_new);
                    if constexpr (kCoopRu) {
                        // (examined candidates and slab bounds alike: the margins of LB)
                        // The runner-up's f64 point is NOT fetched here, where the kernel has no register to spare (eight more
                        // live across the re-scan loop spilled 21: ~3 us of scratch round trips per pass): slot and bound go
                        // to the home lane through the chunk list's memory, which nobody needs any more this round, and
                        // the home lane fetches the point in phase C, behind its moments.
                        const float lb3 = sqrtf(lb3_2) * (1.0f - 1e-6f) - 4.0f * E;
                        const bool have = found && rupos != 0xFFFFFFFFu && !(slow || n > 1) && lb3 > 0.f && lb_new > 0.f;
                        if (ru_io) s_item[s_home[tid]] = make_uint2(have ? rupos : 0xFFFFFFFFu, __float_as_uint(fminf(lb3, 3.0e38f)));
                    }
                }
                        // the hand-over to the query's home lane (phase C): the winner's f64 point (NaN: none) and index
                if (active) {
                    const unsigned h = s_home[tid];
                    const double none = __longlong_as_double(-1ll);
                    s_q64[h][0] = found ? bq.x : none; s_q64[h][1] = found ? bq.y : none; s_q64[h][2] = found ? bq.z : none;
                    s_dprev[h] = __uint_as_float(found ? bidx : 0xFFFFFFFFu);
                }
                };   // tail
                const float4 me = s_qp[tid];
                const unsigned h = s_home[tid];
                double hq[3];
                query_p(h, hq);
                tail(hq[0], hq[1], hq[2], me.x, me.y, me.z, me.w);
                COOP_MARK(11);                               // search done
            }
        }
        __syncthreads();
        COOP_MARK(12);
        set_phase_prio(4);
        // ---- C: the moments of the round's correspondence, on the home lane: query and partner from LDS (nothing of
        // them was kept across the search, where the kernel sits at the 128 registers it may use: 4 waves per SIMD)
        if constexpr (kCoopRu) {
            // (a searched query's runner-up: asked for now, stored behind the workgroup's partial row -- ru_finish)
            ru_go = false;
            if (ru_io && ((needers >> lane) & 1ull)) {
                const uint2 h = s_item[tid];
                ru_go = true;
                ru_it = it;
                ru_lb3 = __uint_as_float(h.y);
                ru_have = h.x != 0xFFFFFFFFu;
                if (ru_have) ru_pending = sorted64[h.x];
            }
        }
        {
            const unsigned pidx = __float_as_uint(s_dprev[tid]);
            if (active && pidx != 0xFFFFFFFFu) {
                double hq[3];
                query_p((unsigned)tid, hq);
                moments(hq[0], hq[1], hq[2], pidx, s_q64[tid][0], s_q64[tid][1], s_q64[tid][2]);
            }
        }
        if constexpr (!ONE) {
            ru_finish();
            __syncthreads();                                 // (the next round re-uses the slots, the list and the queue)
        }
    };
    if constexpr (ONE) {
        round(0);
        COOP_MARK(13);                                       // phase C done
    } else {
        for (int it = 0; it < per_group; it++) round(it);    // (workgroup-uniform trip count: the rounds hold barriers)
    }
    COOP_MARK(6);                                            // outputs + moments
    COOP_WAVE_DONE();
    bool published = false;
    if constexpr (PERSIST) {
        // persistent launch: rows as tagged granules, the fold by polling (device_common.h: polled_fold; the launcher
        // refuses a persistent launch without the granule buffers)
        const unsigned tag = fold_row_tag(fold.seq);
        block_reduce_store<NACC, NW, true>(acc, partials, true, fold.rows_tagged, tag);
        ru_finish();
        COOP_MARK(7);
        published = polled_fold<PLANE, NTH>(fold, lb, bpp, tag);
    } else {
        block_reduce_store<NACC, NW, false>(acc, partials, fold.tickets != nullptr);
        if constexpr (ONE) ru_finish();
        COOP_MARK(7);                                        // workgroup's partial row stored
        if (fold.tickets) published = fused_fold<PLANE, NTH, false, false>(fold, partials, row0, lb, bpp, prob);   // (no SOLVE epilogue here: the
        // one-thread solve spilled 43 registers of this 128-register kernel; FoldArgs::solve is for grid.hip / grid_wave.hip)
    }
    COOP_MARK(8);                                            // fold (most workgroups: just the ticket)
    return published;
}

#define VISMA_COOP_PARAMS                                                                                         \
    int ns, const float *__restrict__ s12f, const unsigned *__restrict__ start, GridParams g,                    \
        const float4 *__restrict__ nrm, Xform64 T64, Offset64 off, float r2f, int *__restrict__ idx_out,         \
        float *__restrict__ d2_out, double *__restrict__ partials, unsigned long long *__restrict__ cand_count,  \
        const DevIcpState *__restrict__ st, int bpp, long long out_stride, const ProbDesc *__restrict__ descs,   \
        int nprob, const Pt64 *__restrict__ src64, const Pt64 *__restrict__ sorted64,                            \
        const Pt64 *__restrict__ nrm64, const FoldArgs fold, double *__restrict__ d64_out,                       \
        Pt64 *__restrict__ wst_io, int warm, Xform64 Tprev, Pt64 *__restrict__ ru_io
#define VISMA_COOP_ARGS                                                                                          \
    ns, s12f, start, g, nrm, T64, off, r2f, idx_out, d2_out, partials, cand_count, st, bpp, out_stride, descs,   \
        nprob, src64, sorted64, nrm64, fold, d64_out, wst_io, warm, Tprev, ru_io
// One query per lane: C4's 262,144 queries are 4096 waves, all resident at once only at 4 waves per SIMD
// (<= 128 VGPRs; the kernel needs 104).  Several queries per lane: the 23 / 29 f64 moments stay live across
// the queries, so the compiler gets the registers it asks for (2 waves per SIMD; such launches have more
// waves than the chip holds anyway).
template <bool PLANE>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void nn_coop_kernel_one(VISMA_COOP_PARAMS)
{
    (void)coop_body<PLANE, true, kBlock>(VISMA_COOP_ARGS);
}
template <bool PLANE>
__global__ __launch_bounds__(kBlock) void nn_coop_kernel_many(VISMA_COOP_PARAMS)
{
    (void)coop_body<PLANE, false, kBlock>(VISMA_COOP_ARGS);
}

// ---- The PERSISTENT launch (kernels.h: PersistArgs): up to pa.max_passes passes of ONE registration in one launch.
// Every workgroup of the launch must be resident at once (coop_persist_capacity; the launcher refuses otherwise): the
// fold of a pass needs the partial row of every workgroup, and a workgroup that is not running cannot write its.
// Between two passes
//   * the workgroup that finished the fold -- it has just sent the statistics to the host -- polls the command block in
//     host memory with its first wave (24 lanes: the 3 x 4 transform as 8-byte words {half | tag}, one lane: the
//     command), one PCIe round trip per poll, and stores the words it accepted to the relay in device memory;
//   * the first wave of every other workgroup polls the relay (agent scope: served by the fabric, not by PCIe);
//   * the other waves sleep in the barrier.
// A word is accepted when it carries the tag of the pass that is due: nothing orders the words among themselves, so
// no fence (an agent- or system-scope release would write the launch's megabytes of dirty state back first).
// The HOST keeps the patience: a thread that comes back later than VISMA_ICP_PERSIST_TIMEOUT_MS after it saw the
// statistics posts STOP instead of the transform and runs the pass as an ordinary launch.  The launch itself only
// protects the GPU against a host that never comes back: after the publication of a pass's statistics every
// workgroup waits four times that patience for a command (so no GO can arrive while some have already left), then
// leaves; before the publication -- the pass is still running somewhere -- it waits as long as the pass takes (a cap
// of a minute that only a lost workgroup could reach).  A launch that ended that way leaves the host a pass without
// statistics: it forgets the winners, re-arms the fold and carries on with ordinary launches.
__device__ __forceinline__ unsigned long long persist_wait(const PersistArgs &pa, unsigned tag, bool poller, int pass, int lane)
{
    const bool mine = lane < kPersistWords;
    unsigned long long w = 0ull;
    const bool direct = pa.direct != 0;                      // the command block lies in device memory: everybody reads it
    // the workgroup that published says so (the others' patience counts from there: until then the pass is still
    // running somewhere, for as long as it takes)
    if (poller && lane == 0)
        __hip_atomic_store(pa.relay + kPersistPublished, (unsigned long long)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long long t_ref = (long long)wall_clock64();
    bool published = poller;
    for (;;) {
        if (mine) {
            if (poller || direct) w = __hip_atomic_load(pa.host_cmd + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else w = __hip_atomic_load(pa.relay + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool ok = !mine || (unsigned)(w >> 32) == tag;
        if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) break;
        const long long now = (long long)wall_clock64();
        bool all_here = true, dead = false;
        if (!published) {
            const unsigned long long m = __hip_atomic_load(pa.relay + kPersistPublished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)m == tag) { published = true; t_ref = now; }
            else {
                all_here = __hip_atomic_load(pa.relay + kPersistStarted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned long long)gridDim.x;
                // (somebody left early -- its workgroups were not all running in time --: no fold of this launch can
                //  complete any more, whatever the others of us still do)
                dead = __hip_atomic_load(pa.relay + kPersistDead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull;
            }
        }
        // (published: a command is due within the host's patience x 4; not everybody has begun: the launch may never
        //  be complete -- somebody else holds compute units -- and says so soon; else the pass is still running somewhere)
        const long long limit = published ? pa.wait_ticks : (!all_here ? (pa.start_ticks > 0 ? pa.start_ticks : pa.wait_ticks) : pa.hard_ticks);
        if (dead || now - t_ref > limit) {
            // nothing came: everybody leaves (the command word decides; the transform words are not looked at)
            w = ((unsigned long long)tag << 32) | (lane == kPersistWords - 1 ? kPersistAbort : 0u);
            if (lane == 0) {
                __hip_atomic_store(pa.relay + kPersistDead, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(pa.host_flag, (unsigned)pass, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            break;
        }
#ifndef VISMA_PERSIST_SLEEP
#define VISMA_PERSIST_SLEEP 1        /* (round 6: 1 instead of 3 -- 64 instead of 192 cycles between two polls of the relay: 0.3 us per pass) */
#endif
        if (poller) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(VISMA_PERSIST_SLEEP);
    }
    if (poller && mine && !direct) __hip_atomic_store(pa.relay + lane, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return w;
}

// Everything the launch is given, as ONE by-value argument: the kernel reads it from the kernarg segment at the top of
// EVERY pass through a pointer the compiler cannot see through.  With ordinary arguments the loop around the pass had
// every invariant of the body hoisted out of it -- uniform and per-lane values that then stayed live across the
// search: 225 scalar registers spilled, and the vector lanes they were spilled to spilled to scratch (584 bytes).
struct CoopPersistParams {
    int ns; const float *s12f; const unsigned *start; GridParams g; const float4 *nrm; Xform64 T64; Offset64 off; float r2f;
    int *idx_out; float *d2_out; double *partials; unsigned long long *cand_count; int bpp; const Pt64 *src64;
    const Pt64 *sorted64; const Pt64 *nrm64; FoldArgs fold; Pt64 *wst_io; int warm; Xform64 Tprev; PersistArgs pa;
    Pt64 *ru_io;
};

// a (member of a) kernel argument read through the laundered kernarg pointer, word by word
template <class T>
__device__ __forceinline__ T ld_karg(const T __attribute__((address_space(4))) *p)
{
    static_assert(sizeof(T) % 4 == 0, "words");
    constexpr int N = (int)(sizeof(T) / 4);
    union U { T v; unsigned w[N]; __device__ U() {} } u;
    typedef const unsigned __attribute__((address_space(4))) *WordPtr;
    const WordPtr pw = (WordPtr)p;
#pragma unroll
    for (int k = 0; k < N; k++) u.w[k] = pw[k];
    return u.v;
}

template <bool PLANE>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void nn_coop_kernel_persist(const CoopPersistParams P)
{
    // The transforms of the pass that is due and of the one before it live in LDS (32-bit halves of the 2 x 12
    // doubles, two slots used alternately) and become scalars at the top of every pass.
    __shared__ unsigned s_tw[2][24];
    __shared__ unsigned s_cmdw;
    // what the queries carry from pass to pass (CoopRes): read from global memory by the first pass, written back once
    __shared__ double r_q64[kBlock][3];
    __shared__ float r_idx[kBlock], r_lb[kBlock], r_lb3[kCoopRu ? kBlock : 1];
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(P.pa.relay + kPersistStarted, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (this workgroup runs)
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const unsigned long long c = (unsigned long long)__double_as_longlong(P.T64.m[k]);
            const unsigned long long q = (unsigned long long)__double_as_longlong(P.Tprev.m[k]);
            s_tw[0][2 * k] = (unsigned)c; s_tw[0][2 * k + 1] = (unsigned)(c >> 32);
            s_tw[1][2 * k] = (unsigned)q; s_tw[1][2 * k + 1] = (unsigned)(q >> 32);
        }
    }
    __syncthreads();
    typedef const CoopPersistParams __attribute__((address_space(4))) *KernargPtr;
    // the lane's source point for the NEXT pass, asked for before the wait for its command (VISMA_PERSIST_PREFETCH=0: not)
#ifndef VISMA_PERSIST_PREFETCH
#define VISMA_PERSIST_PREFETCH 1
#endif
    double nsx = 0.0, nsy = 0.0, nsz = 0.0;
    // (round 6) a launch queued behind the cold pass of its registration, while the host still waits for that pass's
    // statistics: the transform of its first pass arrives like every later one, as a command (tag0) -- the dispatch, the
    // ramp of a thousand workgroups and the host's turn-around overlap instead of following each other
    const int tag_shift = P.pa.wait_first ? 1 : 0;
    if (P.pa.wait_first) {
        const int tidx = thread_number<true>();
        if (tidx < 64) {
            const unsigned long long cw = persist_wait(P.pa, P.pa.tag0, blockIdx.x == 0, 0, tidx);
            if (tidx < 24) s_tw[0][tidx] = (unsigned)cw;
            if (tidx == kPersistWords - 1) s_cmdw = (unsigned)cw;
        }
        __syncthreads();
        // (STOP, or nobody came: no pass has run, nothing to write back)
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)s_cmdw) != kPersistGo) return;
    }
    for (int pass = 1;; pass++) {
        KernargPtr kp = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));                         // (nothing read through it is loop-invariant to the compiler)
        // (member by member: scalar loads straight into registers)
#define VISMA_KARG(F_) ld_karg(&kp->F_)
        const unsigned long long t_begin = VISMA_KARG(pa.timeline) ? wall_clock64() : 0ull;
        unsigned work = 0u;
        const int cur = (pass - 1) & 1;
        Xform64 Tc, Tp;
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)s_tw[cur][2 * k]);
            const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)s_tw[cur][2 * k + 1]);
            Tc.m[k] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
            Tp.m[k] = 0.0;
        }
        // (the transform before this pass's: only the launch's first pass needs it as numbers -- later passes find the
        //  point it gave, which is all the certificate wants of it, where the pass before left it: coop_body, phase A)
        if (pass == 1 || !VISMA_PERSIST_PP_LDS) {
#pragma unroll
            for (int k = 0; k < 12; k++) {
                const unsigned plo = (unsigned)__builtin_amdgcn_readfirstlane((int)s_tw[cur ^ 1][2 * k]);
                const unsigned phi = (unsigned)__builtin_amdgcn_readfirstlane((int)s_tw[cur ^ 1][2 * k + 1]);
                Tp.m[k] = __longlong_as_double((long long)(((unsigned long long)phi << 32) | plo));
            }
        }
        // (after the first pass: the state the pass before left is there, and so is its transform)
        int w = VISMA_KARG(warm);
        if (pass > 1) { w |= 1; if (!(w & 8)) w |= 4; }
        FoldArgs f{};
        f.tickets = VISMA_KARG(fold.tickets); f.partials2 = VISMA_KARG(fold.partials2);
        f.ticket_stride = VISMA_KARG(fold.ticket_stride); f.stats_out = VISMA_KARG(fold.stats_out);
        f.stats_stride = VISMA_KARG(fold.stats_stride); f.host_out = VISMA_KARG(fold.host_out);
        f.seq = VISMA_KARG(fold.seq) + (unsigned long long)(pass - 1);
        // (source-sharded ranks, one per GPU: the folding workgroup exchanges with the peers through their mailboxes, read
        //  from the table in device memory)
        f.ipc_n = VISMA_KARG(fold.ipc_n); f.ipc_rank = VISMA_KARG(fold.ipc_rank); f.ipc_seq_dev = VISMA_KARG(fold.ipc_seq_dev);
        f.ipc_flag = VISMA_KARG(fold.ipc_flag); f.ipc_spins = VISMA_KARG(fold.ipc_spins); f.peer_table = VISMA_KARG(fold.peer_table);
        f.rows_tagged = VISMA_KARG(fold.rows_tagged); f.rows2_tagged = VISMA_KARG(fold.rows2_tagged);
        f.dead_flag = VISMA_KARG(fold.dead_flag); f.poll_ticks = VISMA_KARG(fold.poll_ticks);
        const bool published = coop_body<PLANE, true, kBlock, true>(
            VISMA_KARG(ns), VISMA_KARG(s12f), VISMA_KARG(start), VISMA_KARG(g), VISMA_KARG(nrm), Tc, VISMA_KARG(off), VISMA_KARG(r2f),
            VISMA_KARG(idx_out), VISMA_KARG(d2_out), VISMA_KARG(partials), VISMA_KARG(cand_count), nullptr, VISMA_KARG(bpp), 0ll,
            nullptr, 1, VISMA_KARG(src64), VISMA_KARG(sorted64), VISMA_KARG(nrm64), f, nullptr, VISMA_KARG(wst_io), w, Tp,
            VISMA_KARG(ru_io), t_begin ? &work : nullptr, CoopRes{r_q64, r_idx, r_lb, r_lb3, pass == 1 ? 1 : 0, nsx, nsy, nsz,
                                                                        (VISMA_PERSIST_PREFETCH && pass > 1) ? 1 : 0});
        const PersistArgs pa = VISMA_KARG(pa);
        if (pa.timeline && pass <= pa.timeline_passes && thread_number<true>() == 0) {
            // (measurement runs only) [pass][workgroup]{begin, body done}; the begin of pass 1 is the launch's
            unsigned long long *slot = pa.timeline + ((unsigned long long)(pass - 1) * gridDim.x + blockIdx.x) * 2ull;
            slot[0] = t_begin;
            // (the clock's low 44 bits -- 48 hours at 100 MHz -- and above them: queued queries, chunks listed)
            slot[1] = (wall_clock64() & 0xFFFFFFFFFFFull) | ((unsigned long long)work << 44);
        }
        bool last = pass >= pa.max_passes;
        nsx = nsy = nsz = 0.0;                               // (dead across the body: assigned on every path)
        if (VISMA_PERSIST_PREFETCH && !last) {
            // the next pass's source point: the load is in flight while the fold completes and the host turns around
            // (pinned below: the compiler would otherwise sink it to its use, behind the wait)
            const int tidx = thread_number<true>();
            const int bpp = VISMA_KARG(bpp);
            int vb, per_group;
            long long i_begin, i_end;
            coop_query_range<kBlock>(VISMA_KARG(ns), bpp, (int)blockIdx.x % bpp, tidx, vb, per_group, i_begin, i_end);
            if (i_begin < i_end) {
                const Pt64 s8 = VISMA_KARG(src64)[i_begin];
                nsx = s8.x; nsy = s8.y; nsz = s8.z;
            }
        }
        if (!last) {
            const int tidx = thread_number<true>();
            if (tidx < 64) {
                const bool poller = __builtin_amdgcn_readfirstlane((int)published) != 0;     // (the same on every lane: scalar)
                const unsigned long long cw = persist_wait(pa, pa.tag0 + (unsigned)(pass - 1 + tag_shift), poller, pass, tidx);
                // the new transform takes the slot of the one before the pass that just ran
                if (tidx < 24) s_tw[cur ^ 1][tidx] = (unsigned)cw;
                if (tidx == kPersistWords - 1) s_cmdw = (unsigned)cw;
            }
            __syncthreads();
            asm volatile("" : "+v"(nsx), "+v"(nsy), "+v"(nsz));
            const unsigned cmd = (unsigned)__builtin_amdgcn_readfirstlane((int)s_cmdw);
            last = cmd != kPersistGo;
        }
        if (last) {
            // The launch ends (its last pass, STOP, nobody came): what the last completed pass left goes back to memory --
            // correspondence and state of every query, exactly what a one-pass launch of that pass would have written.
            const int tidx = thread_number<true>();
            const int bpp = VISMA_KARG(bpp);
            int vb, per_group;
            long long i_begin, i_end;
            coop_query_range<kBlock>(VISMA_KARG(ns), bpp, (int)blockIdx.x % bpp, tidx, vb, per_group, i_begin, i_end);
            if (i_begin < i_end) {
                const unsigned pid = __float_as_uint(r_idx[tidx]);
                Pt64 o8;
                o8.x = r_q64[tidx][0]; o8.y = r_q64[tidx][1]; o8.z = r_q64[tidx][2];      // (NaN, all bits set: no partner)
                o8.w = (unsigned long long)pid | ((unsigned long long)__float_as_uint(r_lb[tidx]) << 32);
                // the squared distance of the last pass's correspondence, as that pass computed it: its transform (still
                // in its slot), the products of se3_act, flann's sum (dist.h:159-176)
                float d2 = VISMA_KARG(r2f);
                if (pid != 0xFFFFFFFFu) {
                    double Tl[12];
#pragma unroll
                    for (int k = 0; k < 12; k++)
                        Tl[k] = __longlong_as_double((long long)(((unsigned long long)s_tw[cur][2 * k + 1] << 32) | s_tw[cur][2 * k]));
                    const Pt64 s8 = VISMA_KARG(src64)[i_begin];
                    const double sv[3] = {s8.x, s8.y, s8.z};
                    double hp[3];
                    se3_act(Tl, sv, hp);
                    const double dx = o8.x - hp[0], dy = o8.y - hp[1], dz = o8.z - hp[2];
                    double d = dx * dx;
                    d += dy * dy;
                    d += dz * dz;
                    d2 = (float)d;
                }
                VISMA_KARG(idx_out)[i_begin] = (int)pid;                                // (~0 = -1: none)
                VISMA_KARG(d2_out)[i_begin] = d2;
                VISMA_KARG(wst_io)[i_begin] = o8;
                if constexpr (kCoopRu) {
                    // (the runner-up's point and index were written through when they changed; LB3 moved with every pass)
                    Pt64 *ru = VISMA_KARG(ru_io);
                    if (ru) reinterpret_cast<unsigned *>(&ru[i_begin].w)[1] = __float_as_uint(r_lb3[tidx]);
                }
            }
            break;
        }
#undef VISMA_KARG
    }
}
#undef VISMA_COOP_PARAMS
#undef VISMA_COOP_ARGS

#define VISMA_COOP_LAUNCH(KERNEL_)                                                                               \
    hipLaunchKernelGGL(KERNEL_, dim3(total_blocks), dim3(kBlock), 0, stream, ns, s12, start, g, nrm, T64, off,   \
                       r2f, idx_out, d2_out, partials, cand_count, st, bpp, out_stride, descs, nprob, src64,     \
                       sorted64, nrm64, fold, d64_out, wst_io, warm, Tp, ru_io)

// The warm-started, flattened exact search.  Shared clouds: `nprob` problems of `bpp` workgroups each
// (descs == NULL); own clouds: descs[nprob], total_blocks workgroups.  `one`: at most one query per lane.
// warm & 2: a workgroup -> problem map (int per workgroup) follows descs[nprob].
// wst_io (one Pt64 per query, laid out like idx_out): read when `warm & 1` (the winners of the previous pass
// over the SAME source order and target: f64 point, original index | LB << 32; NaN coordinates = none, all bits
// set = nothing known), always written.  warm & 4: Tprev is the transform of the pass that left the state
// (device loops take it from their DevIcpState instead) -- queries whose winner provably cannot have changed skip
// the search (the certificate above).
hipError_t launch_nn_coop(int total_blocks, int bpp, int nprob, const ProbDesc *descs, int ns, const float *s12,
                          const unsigned *start, const GridParams &g, const float4 *nrm, const Pt64 *nrm64,
                          const Xform64 &T64, const Offset64 &off, float r2f, int point_to_plane, int one,
                          int32_t *idx_out, float *d2_out, double *partials, unsigned long long *cand_count,
                          const DevIcpState *st, long long out_stride, const Pt64 *src64, const Pt64 *sorted64,
                          const FoldArgs &fold, double *d64_out, Pt64 *wst_io, int warm, hipStream_t stream,
                          const Xform64 *Tprev, const PersistArgs *persist, Pt64 *ru_io)
{
    if (!src64 || !sorted64 || !s12 || !wst_io) return hipErrorInvalidValue;
    if (persist) {
        // (the certificate kernels carry no solve epilogue: a fold that asks for one would leave the state where it is,
        //  silently -- refuse it here rather than trust the caller's predicate)
        if (fold.solve) return hipErrorInvalidValue;
        // one registration, one query per lane, the fold and its publication inside the launch, everybody resident
        if (descs || nprob != 1 || !one || st || !fold.rows_tagged || !fold.rows2_tagged || !fold.dead_flag || !fold.host_out ||
            (fold.ipc_n > 1 && !fold.peer_table) || d64_out ||
            persist->max_passes < 1 || !persist->host_cmd || !persist->relay || !persist->host_flag ||
            total_blocks > coop_persist_capacity(point_to_plane))
            return hipErrorInvalidValue;
        CoopPersistParams P{};
        P.ns = ns; P.s12f = s12; P.start = start; P.g = g; P.nrm = nrm; P.T64 = T64; P.off = off; P.r2f = r2f;
        P.idx_out = idx_out; P.d2_out = d2_out; P.partials = partials; P.cand_count = cand_count; P.bpp = bpp;
        P.src64 = src64; P.sorted64 = sorted64; P.nrm64 = nrm64; P.fold = fold; P.wst_io = wst_io;
        if (Tprev) { P.Tprev = *Tprev; warm |= 4; } else warm &= ~4;
        P.warm = warm; P.pa = *persist; P.ru_io = ru_io;
        if (point_to_plane) hipLaunchKernelGGL(nn_coop_kernel_persist<true>, dim3(total_blocks), dim3(kBlock), 0, stream, P);
        else hipLaunchKernelGGL(nn_coop_kernel_persist<false>, dim3(total_blocks), dim3(kBlock), 0, stream, P);
        return hipGetLastError();
    }
    // batches (problems with their own clouds) and sweeps over shared clouds: round 3's wave-synchronous kernel
    // (grid_wave.hip says why); VISMA_ICP_COOP_KERNEL=cert / wave forces one of the two (A/B timing)
    static const int forced = [] {
        const char *e = std::getenv("VISMA_ICP_COOP_KERNEL");
        return !e ? 0 : (e[0] == 'c' ? 1 : (e[0] == 'w' ? 2 : 0));
    }();
    if (forced == 2 || (forced == 0 && (descs || nprob > 1)))
        return launch_nn_wave(total_blocks, bpp, nprob, descs, ns, s12, start, g, nrm, nrm64, T64, off, r2f, point_to_plane, one,
                              idx_out, d2_out, partials, cand_count, st, out_stride, src64, sorted64, fold, d64_out, wst_io,
                              warm & (3 | 8), stream);       // (8: certificates off -- the wave kernel's no-partner one too)
    if (fold.solve) return hipErrorInvalidValue;            // (as above: only launch_nn_wave's kernels honour it)
    Xform64 Tp{};
    if (Tprev) { Tp = *Tprev; warm |= 4; } else warm &= ~4;
    if (point_to_plane) {
        if (one) VISMA_COOP_LAUNCH(nn_coop_kernel_one<true>); else VISMA_COOP_LAUNCH(nn_coop_kernel_many<true>);
    } else {
        if (one) VISMA_COOP_LAUNCH(nn_coop_kernel_one<false>); else VISMA_COOP_LAUNCH(nn_coop_kernel_many<false>);
    }
    return hipGetLastError();
}
#undef VISMA_COOP_LAUNCH

int coop_persist_capacity(int point_to_plane)
{
    // (per device: contexts of one process may sit on different GPUs or partitions)
    static int cap[2][64];
    static bool known[2][64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 0; }
    const int p = point_to_plane ? 1 : 0;
    if (!known[p][dev]) {
        int per_cu = 0, cus = 0;
        hipError_t e = point_to_plane
                           ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, nn_coop_kernel_persist<true>, kBlock, 0)
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, nn_coop_kernel_persist<false>, kBlock, 0);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) { (void)hipGetLastError(); per_cu = 0; cus = 0; }
        cap[p][dev] = per_cu * cus;
        known[p][dev] = true;
    }
    return cap[p][dev];
}

}  // namespace visma
