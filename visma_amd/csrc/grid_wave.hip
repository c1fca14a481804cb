// grid_wave.hip -- round 3's warm-started search kernel, kept for BATCHES (problems with their own clouds, sweeps over
// shared clouds): every wave searches its own 64 queries, no certificate, no workgroup barriers.  Measured against the
// certificate kernel of grid_coop.hip (round 4) on one box: config 3 986 k vs 888 k iterations/s, config 5 2.18 M vs
// 1.93 M, a 24-yaw sweep 0.92 vs 1.00 ms -- hundreds of small problems, 30 iterations each from far-off starts, with more
// waves than the chip holds: the waves of a workgroup in lock step through six barriers hide less latency than sixteen
// independent waves per CU.  Single registrations run grid_coop.hip.  The per-query state is grid_coop.hip's (Pt64: the
// winner's f64 point, index | LB << 32); this kernel reads the point and leaves LB = 0 (nothing certified from it).
//
// (round 3's description follows)  The exact radius-cell search, WARM-STARTED and with the candidates of a WAVE flattened
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
//    (prevq_io: its fp32 coordinates, 16 bytes per query, streamed in with the source point) gives
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
#include "device_common.h"
// (round 3's measurement hooks: no-ops here)
#define COOP_PROBE_BEGIN() do { } while (0)
#define COOP_MARK(k) do { } while (0)
#define COOP_PHASE(k, u, f) do { } while (0)
#define COOP_WAVE_DONE() do { } while (0)

namespace visma {

namespace {

struct P12 { float x, y, z; };                // fp32 rounding of a cell-sorted f64 target point
constexpr int kCoopCap = 512;                 // chunk descriptors per wave and list window (4 KiB)
#ifndef VISMA_WAVE_DEPTH
#define VISMA_WAVE_DEPTH 7
#endif
constexpr int kCoopDepth = VISMA_WAVE_DEPTH;  // chunks per lane octet in flight

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

// Everything the launch is given, as ONE by-value argument read from the kernarg segment at the top of EVERY pass through a
// pointer the compiler cannot see through (grid_coop.hip's persistent kernel does the same): with ordinary arguments the loop
// around the pass had every invariant of the body hoisted out of it and spilled.
struct SweepParams {
    int ns; const float *s12f; const unsigned *start; GridParams g; float r2f; int *idx_out; float *d2_out; double *partials;
    unsigned long long *cand_count; DevIcpState *st; int bpp; long long out_stride; int nprob; const Pt64 *src64;
    const Pt64 *sorted64; FoldArgs fold; Pt64 *prevq_io; SweepArgs sa;
};
template <class T>
__device__ __forceinline__ T ld_karg_w(const T __attribute__((address_space(4))) *p)
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

template <bool PLANE, bool ONE, bool SWEEP = false>
__device__ __forceinline__ bool wave_body(
    int ns, const float *__restrict__ s12f, const unsigned *__restrict__ start, GridParams g,
    const float4 *__restrict__ nrm, Xform64 T64, Offset64 off, float r2f, int *__restrict__ idx_out,
    float *__restrict__ d2_out, double *__restrict__ partials, unsigned long long *__restrict__ cand_count,
    const DevIcpState *__restrict__ st, int bpp, long long out_stride, const ProbDesc *__restrict__ descs,
    int nprob, const Pt64 *__restrict__ src64, const Pt64 *__restrict__ sorted64,
    const Pt64 *__restrict__ nrm64, const FoldArgs &fold, double *__restrict__ d64_out,
    Pt64 *__restrict__ prevq_io, int warm, int sweep_pass = 0)
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
        prevq_io += d.out_off;
    } else {
        prob = blockIdx.x / bpp;
        lb = blockIdx.x - prob * bpp;
        row0 = (long long)prob * bpp;
    }
    if (st) st += prob;
    {
        Xform32 T32_unused;
        if (!load_loop_state(st, T32_unused, T64, off, r2f)) return false;
    }
    const double r2d = (double)r2f;                         // (double)(float)(r*r): KDTreeFlann.cpp:184-185
    COOP_PROBE_BEGIN();
    idx_out += (long long)prob * out_stride;
    d2_out += (long long)prob * out_stride;
    prevq_io += (long long)prob * out_stride;
    double acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++) acc[a] = 0.0;
    unsigned ncand = 0, ncand_all = 0;

    // (inside the persistent sweep's loop: opaque, or everything derived from the thread's number is hoisted out and spilled)
    const int tid = thread_number<SWEEP, kBlock>(), lane = tid & 63, wave = tid >> 6;
    const int oct = lane >> 3, l8 = lane & 7;
    // the query -> lane map of nn_grid_reduce_kernel<G = 1> (XCD-aware chunking of the Morton order)
    int vb = lb;
    if ((bpp & 7) == 0) vb = (lb & 7) * (bpp >> 3) + (lb >> 3);
    const int total_groups = bpp * kBlock;
    const int per_group = (ns + total_groups - 1) / total_groups;
    const int gid = vb * kBlock + tid;
    const long long i_begin = (long long)gid * per_group;
    const long long i_end = i_begin + per_group < ns ? i_begin + per_group : ns;

    __shared__ float4 s_qp[kBlock];                         // (px, py, pz, W) of the query of each lane
    __shared__ uint2 s_item[kBlock / 64][kCoopCap + 64];    // chunk descriptors, completed by chunk results (+ 64 null
                                                            // descriptors behind the last one: the list is read unguarded)
    uint2 *items = s_item[wave];

    // one query (or none: the lanes past the end still work on the others' chunks)
    auto query = [&](long long i, bool active) {
        // ---- the query: the reference's transform of a source point (PointCloud.cpp:75-80), in f64
        Pt64 s8 = Pt64{0.0, 0.0, 0.0, 0ull};
        float4 qprev = make_float4(NAN, NAN, NAN, 0.f);     // the previous pass's winner (fp32 view), NaN = none
        if (active) {
            s8 = src64[i];
            if (warm & 1) {
                // (the state of grid_coop.hip: the previous winner's f64 point; all bits set = NaN = none)
                const Pt64 w8 = prevq_io[i];
                qprev = make_float4((float)w8.x, (float)w8.y, (float)w8.z, 0.f);
            }
        }
        // (se3_act: the restatement of SE3Type's action on a point, core/se3.h:103-106 -- same products, same order)
        double pd[3];
        {
            const double sv[3] = {s8.x, s8.y, s8.z};
            se3_act(T64.m, sv, pd);
        }
        const double pxd = pd[0], pyd = pd[1], pzd = pd[2];
        const float px = (float)pxd, py = (float)pyd, pz = (float)pzd;
        COOP_PHASE(0, 0u, px + py + pz + qprev.x + qprev.y + qprev.z);   // source + previous winner arrived
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
        auto load_row = [&](int k, bool want) {
            const bool ok = want && span > 0 && zok[k / 3] && yok[k % 3];
            u4a v = {0u, 0u, 0u, 0u};
            if (ok) v = *reinterpret_cast<const u4a *>(reinterpret_cast<const char *>(start) + (unsigned)(row_c + (k / 3 - 1) * pitch_z + (k % 3 - 1) * pitch_y) * 4u);
            return v;
        };
        // rounding band (exact_band): E bounds |d64 - sqrt(d2_32)|; L = squared fp32 distance at or beyond
        // which a candidate cannot be accepted in f64; W >= band(m) - m for every m < L
        const float r_f = sqrtf(r2f);
        const float rup = r_f * (1.0f + 2.4e-7f);
        const float E = 2.4e-7f * (fabsf(px) + fabsf(py) + fabsf(pz) + rup) + 4.8e-7f * rup;
        const float tlim = rup + 2.0f * E;
        const float L = tlim * tlim * (1.0f + 6e-7f);
        const float W = (4.0f * E * tlim + 4.0f * E * E) * (1.0f + 1e-6f) + L * 5e-7f;
        s_qp[tid] = make_float4(px, py, pz, W);
        // slab distances (in cells) of the neighbouring rows / cells, margins as in nn_grid_reduce_kernel
        const float fx = (px - g.mn[0]) * g.inv_h - (float)cx;
        const float fy = (py - g.mn[1]) * g.inv_hs - (float)cy;
        const float fz = (pz - g.mn[2]) * g.inv_hs - (float)cz;
        const float mgn = 1e-3f + 4.0f * E * g.inv_hs;
        const float lo_x = fmaxf(fx - mgn, 0.f), hi_x = fmaxf(1.0f - fx - mgn, 0.f);
        const float lo_y = fmaxf(fy - mgn, 0.f), hi_y = fmaxf(1.0f - fy - mgn, 0.f);
        const float lo_z = fmaxf(fz - mgn, 0.f), hi_z = fmaxf(1.0f - fz - mgn, 0.f);
        const float h2 = g.hs * g.hs * (1.0f - 1e-5f);
        float exq[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int xi = x0 + j;
            const float ex = xi == cx ? 0.f : (xi < cx ? lo_x + (float)(cx - xi - 1) : hi_x + (float)(xi - cx - 1));
            exq[j] = ex * ex * h2;
        }
        const float ey2[3] = {lo_y * lo_y, 0.f, hi_y * hi_y}, ez2[3] = {lo_z * lo_z, 0.f, hi_z * hi_z};
        float rowB[9];                                       // squared slab bound of row k = (dy, dz)
#pragma unroll
        for (int k = 0; k < 9; k++) rowB[k] = (ey2[k % 3] + ez2[k / 3]) * h2;
        auto row_bound_of = [&](int k) { return rowB[k]; };
        // what nothing nearer than can be missed by: the previous winner's distance now, or the radius
        float bound0 = L;
        {
            const float dprev = sqdist_f32(qprev, px, py, pz);
            if (dprev < L) bound0 = dprev;                   // (NaN = no previous winner: the radius)
        }
        // ---- only the rows whose slab can hold a point at or within bound0 are looked up at all (a converged
        // pass needs 1.9 of the 9 per query), all of them in ONE round of gathers: nothing depends on a load
        // from here to the candidates
        u4a cs[9];
#pragma unroll
        for (int k = 0; k < 9; k++) cs[k] = load_row(k, active && !(row_bound_of(k) > bound0));
        // ---- the slots of each row that can hold a candidate at or within bound0: cells whose slab bound
        // (y, z and x slab distances, squared) does not exceed it.  A skipped cell holds nothing that could
        // win or tie (margins: fp32 binning + rounding band, as in the lane-serial kernel).
        // (B + exq[j] <= bound0 tested as B <= bound0 - exq[j]: the slab bounds carry a relative margin of ~1e-3, a
        //  rounding of the subtraction cannot drop a cell that matters; a cell that does not exist: -inf)
        const float lim0 = span > 0 ? bound0 - exq[0] : -INFINITY;
        const float lim1 = span > 1 ? bound0 - exq[1] : -INFINITY;
        const float lim2 = span > 2 ? bound0 - exq[2] : -INFINITY;
        unsigned xb[9], xe[9];
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const float B = row_bound_of(k);
            const bool n0 = B <= lim0, n1 = B <= lim1, n2 = B <= lim2;
            const u4a v = cs[k];                             // (a row not looked up reads as zeros: empty)
            const unsigned b = n0 ? v.x : (n1 ? v.y : v.z);
            unsigned e = n2 ? v.w : (n1 ? v.z : (n0 ? v.y : b));
            if (!(n0 || n1 || n2)) e = b;
            xb[k] = b;
            xe[k] = e;
        }
        COOP_PHASE(1, xb[0] ^ xe[8] ^ xb[4] ^ xe[2] ^ xb[6], 0.f);            // row bounds arrived, rows pruned
        if (cand_count) {
            // profiling only (one uniform branch): candidates listed, cell-table rows looked up
#pragma unroll
            for (int k = 0; k < 9; k++) {
                ncand += xe[k] - xb[k];
                if (active && !(row_bound_of(k) > bound0) && span > 0 && zok[k / 3] && yok[k % 3]) ncand_all++;
            }
        }
        // the two best chunks (minimum, first slot, flag byte) and the third chunk minimum
        float gh0 = L, gh1 = L, gh2 = L;
        unsigned gb0 = 0xFFFFFFFFu, gb1 = 0xFFFFFFFFu, gm0 = 0u, gm1 = 0u;
        auto chunk_insert = [&](float m, unsigned b, unsigned flags) {
            const bool c1 = m < gh0, c2 = m < gh1;
            gh2 = __builtin_amdgcn_fmed3f(gh1, gh2, m);
            gh1 = __builtin_amdgcn_fmed3f(gh0, gh1, m);
            gh0 = fminf(gh0, m);
            gb1 = c2 ? b : gb1; gm1 = c2 ? flags : gm1;
            gb1 = c1 ? gb0 : gb1; gm1 = c1 ? gm0 : gm1;
            gb0 = c1 ? b : gb0; gm0 = c1 ? flags : gm0;
        };
        // ---- the rows, chunked and flattened over the wave.  A query's chunks take CONSECUTIVE list entries: the
        // owner reads its results back as one short run (walking the rows again, one LDS round trip per chunk, took
        // 2.5 us of every wave's 17).
        {
            unsigned nq = 0;
#pragma unroll
            for (int k = 0; k < 9; k++) nq += (xe[k] - xb[k] + 7u) >> 3;
            const unsigned incl = wave_scan_incl(nq, lane);
            const unsigned M = (unsigned)__shfl((int)incl, 63, 64);
            const unsigned off_q = incl - nq;
            auto window = [&](const unsigned w0) {
                {
                    // descriptor: (the chunk's first slot, owner's query in LDS | candidates << 16); a query's chunks
                    // take CONSECUTIVE list entries, row after row
                    const unsigned own = (unsigned)tid << 4; // where this lane's query lies in s_qp
                    unsigned j = off_q - w0;                 // (a run that begins before the window wraps: never < cap)
#pragma unroll
                    for (int k = 0; k < 9; k++) {
                        unsigned b = xb[k];
                        while (b < xe[k]) {
                            if (j < (unsigned)kCoopCap) items[j] = make_uint2(b, own | (min(xe[k] - b, 8u) << 16));
                            j++;
                            b += 8u;
                        }
                    }
                }
                const unsigned Mw = min(M - w0, (unsigned)kCoopCap);
                items[Mw + lane] = make_uint2(0u, 0u);       // null descriptors (count 0) for the last, partial trip
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                COOP_MARK(2);                                // chunk list written
                for (unsigned t = 0; t < Mw; t += 8u * kCoopDepth) {
                    // kCoopDepth chunks per lane octet in flight: every load of the list is independent
                    P12 c4[kCoopDepth];
                    unsigned meta[kCoopDepth];
#pragma unroll
                    for (int u = 0; u < kCoopDepth; u++) {
                        const uint2 dsc = items[t + u * 8 + oct];
                        meta[u] = dsc.y;
                        // scalar base + 32-bit byte offset (the launcher keeps 12 * slots below 2^32; the array carries
                        // kSortedSlack entries of slack)
                        c4[u] = *reinterpret_cast<const P12 *>(reinterpret_cast<const char *>(s12) + (((dsc.x * 3u) << 2) + (unsigned)l8 * 12u));
                    }
#pragma unroll
                    for (int u = 0; u < kCoopDepth; u++) {
                        const unsigned cnt = meta[u] >> 16;
                        const float4 p = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_qp) + (meta[u] & 0xFFF0u));
                        float d = sqdist_f32(make_float4(c4[u].x, c4[u].y, c4[u].z, 0.f), p.x, p.y, p.z);
                        const bool mine = (unsigned)l8 < cnt;            // (lane 0 of the octet: the chunk exists)
                        d = mine ? d : INFINITY;
                        const float m = octet_min(d);
                        // (a lane past the chunk's end holds +inf: never within m + W of a finite minimum; a null
                        //  descriptor's result is not stored)
                        const unsigned long long bal = __builtin_amdgcn_ballot_w64(d <= m + p.w);
                        const unsigned flags = (unsigned)(bal >> (oct * 8)) & 0xFFu;
                        // the result takes the place of the descriptor's second word: the chunk minimum rounded DOWN to
                        // 16 mantissa bits | the flag byte (the first word, the chunk's position, stays)
                        if (l8 == 0 && mine) items[t + u * 8 + oct].y = (__float_as_uint(m) & 0xFFFFFF00u) | flags;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                COOP_MARK(3);                                // chunks worked off
                // the owner's run of results, four reads in flight (a lane past its run inserts +inf: no effect)
                for (unsigned c0 = 0; __builtin_amdgcn_ballot_w64(c0 < nq) != 0ull; c0 += 4u) {
                    uint2 r[4];
                    bool in[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const unsigned j = off_q - w0 + c0 + (unsigned)u;
                        in[u] = c0 + (unsigned)u < nq && j < (unsigned)kCoopCap;
                        r[u] = items[in[u] ? j : 0u];
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        chunk_insert(in[u] ? __uint_as_float(r[u].y & 0xFFFFFF00u) : INFINITY, r[u].x, r[u].y & 0xFFu);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            };
            // (one window unless the cloud is very dense)
            for (unsigned w0 = 0; w0 < M; w0 += kCoopCap) window(w0);
        }
        COOP_PHASE(4, gb0 + gb1 + gm0 + gm1, gh0 + gh1 + gh2);              // chunk results merged per query
        // ---- the f64 decision: flagged candidates of the kept chunks inside g + W
        double bd = r2d;                                     // best d2 so far (strictly below r2d once set)
        unsigned bidx = 0xFFFFFFFFu, bpos = 0xFFFFFFFFu;
        Pt64 bq = Pt64{0.0, 0.0, 0.0, 0ull};
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
        if (active && gb0 != 0xFFFFFFFFu) {
            const float thr = g_up + W;
            unsigned c[4] = {0u, 0u, 0u, 0u};
            int n = 0;
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
            unsigned qb[9], pre[10];
            pre[0] = 0u;
#pragma unroll
            for (int k = 0; k < 9; k++) {
                qb[k] = bcast_u(xb[k]);
                pre[k + 1] = pre[k] + (bcast_u(xe[k]) - qb[k]);
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
        COOP_PHASE(5, bidx, (float)(bd + bq.x));                            // f64 winner arrived and ranked
        if (active) {
            idx_out[i] = (bpos == 0xFFFFFFFFu) ? -1 : (int)bidx;
            d2_out[i] = (float)bd;
            // the winner as the candidate array holds it (fp32 rounding of the same f64 value): next pass's bound
            {
                // the state for the next pass: the winner's f64 point and index, LB = 0 (no certificate can follow from it)
                Pt64 o8;
                o8.x = o8.y = o8.z = __longlong_as_double(-1ll);
                o8.w = 0xFFFFFFFFull;
                if (bpos != 0xFFFFFFFFu) { o8.x = bq.x; o8.y = bq.y; o8.z = bq.z; o8.w = (unsigned long long)bidx; }
                prevq_io[i] = o8;
            }
            if (d64_out) d64_out[i] = bd;                    // (target-sharded ranks compare shards in f64)
            if (bpos != 0xFFFFFFFFu) {
                double nx = 0.0, ny = 0.0, nz = 0.0;
                if (PLANE) {
                    if (nrm64) { const Pt64 n8 = nrm64[(unsigned)bq.w]; nx = n8.x; ny = n8.y; nz = n8.z; }
                    else { const float4 n4 = nrm[(unsigned)bq.w]; nx = n4.x; ny = n4.y; nz = n4.z; }
                }
                accumulate_pq_d<PLANE>(acc, pxd, pyd, pzd, bq.x, bq.y, bq.z, nx, ny, nz, off);
            }
        }
    };
    if constexpr (ONE) {
        query(i_begin, i_begin < i_end);
    } else {
        for (int it = 0; it < per_group; it++) query(i_begin + it, i_begin + it < i_end);   // wave-uniform trip count
    }
    COOP_MARK(6);                                            // outputs + moments
    COOP_WAVE_DONE();
    block_reduce_store<NACC, kBlock / 64, SWEEP>(acc, partials, SWEEP || fold.tickets != nullptr);   // (rows read by another workgroup: past the L2)
    COOP_MARK(7);                                            // workgroup's partial row stored
    if (cand_count) {
        unsigned long long c = ncand, ca = ncand_all;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            c += __shfl_down(c, o, 64);
            ca += __shfl_down(ca, o, 64);
        }
        if (lane == 0 && ca) {
            unsigned long long *slot = cand_count + 2 * (blockIdx.x & 4095);
            atomicAdd(slot, c);
            atomicAdd(slot + 1, ca);
        }
    }
    bool published = false;
    if constexpr (SWEEP) {
        // (the persistent sweep launch: fold, closed-form update, compose, stop test and the hand-over of the next transform
        //  by the workgroup that completes the problem's fold)
        // (the fold's arguments are read from the kernarg segment HERE, not carried across the search)
        typedef const SweepParams __attribute__((address_space(4))) *KernargPtr;
        KernargPtr kp = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        FoldArgs f{};
        f.tickets = ld_karg_w(&kp->fold.tickets); f.partials2 = ld_karg_w(&kp->fold.partials2);
        f.ticket_stride = ld_karg_w(&kp->fold.ticket_stride); f.stats_out = ld_karg_w(&kp->fold.stats_out);
        f.stats_stride = ld_karg_w(&kp->fold.stats_stride);
        f.solve = ld_karg_w(&kp->st);
        f.sweep_relay = ld_karg_w(&kp->sa.relay);
        f.sweep_tag = ld_karg_w(&kp->sa.tag0) + (unsigned)sweep_pass + 1u;
        f.sweep_passes = ld_karg_w(&kp->sa.passes0) + sweep_pass;
        published = fused_fold<PLANE, kBlock, true, true>(f, ld_karg_w(&kp->partials), row0, lb, bpp, prob);
    } else {
        if (fold.tickets) published = fused_fold<PLANE, kBlock, false, kSolveInFold>(fold, partials, row0, lb, bpp, prob);
    }
    COOP_MARK(8);                                            // fold (most workgroups: just the ticket)
    return published;
}

#define VISMA_WAVE_PARAMS                                                                                         \
    int ns, const float *__restrict__ s12f, const unsigned *__restrict__ start, GridParams g,                    \
        const float4 *__restrict__ nrm, Xform64 T64, Offset64 off, float r2f, int *__restrict__ idx_out,         \
        float *__restrict__ d2_out, double *__restrict__ partials, unsigned long long *__restrict__ cand_count,  \
        const DevIcpState *__restrict__ st, int bpp, long long out_stride, const ProbDesc *__restrict__ descs,   \
        int nprob, const Pt64 *__restrict__ src64, const Pt64 *__restrict__ sorted64,                            \
        const Pt64 *__restrict__ nrm64, const FoldArgs fold, double *__restrict__ d64_out,                       \
        Pt64 *__restrict__ prevq_io, int warm
#define VISMA_WAVE_ARGS                                                                                          \
    ns, s12f, start, g, nrm, T64, off, r2f, idx_out, d2_out, partials, cand_count, st, bpp, out_stride, descs,   \
        nprob, src64, sorted64, nrm64, fold, d64_out, prevq_io, warm
// One query per lane: C4's 262,144 queries are 4096 waves, all resident at once only at 4 waves per SIMD
// (<= 128 VGPRs; the kernel needs 104).  Several queries per lane: the 23 / 29 f64 moments stay live across
// the queries, so the compiler gets the registers it asks for (2 waves per SIMD; such launches have more
// waves than the chip holds anyway).
#ifndef VISMA_WAVE_WAVES
#define VISMA_WAVE_WAVES 4
#endif
template <bool PLANE>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(VISMA_WAVE_WAVES, VISMA_WAVE_WAVES))) void nn_wave_kernel_one(VISMA_WAVE_PARAMS)
{
    wave_body<PLANE, true>(VISMA_WAVE_ARGS);
}
template <bool PLANE>
__global__ __launch_bounds__(kBlock) void nn_wave_kernel_many(VISMA_WAVE_PARAMS)
{
    wave_body<PLANE, false>(VISMA_WAVE_ARGS);
}
#undef VISMA_WAVE_PARAMS
#undef VISMA_WAVE_ARGS

// ---- The PERSISTENT SWEEP launch (round 6): the warm passes of nprob registrations over SHARED clouds -- the 24 yaw starts of
// feh::RegisterModelToScene (src/annotation.cpp:35-61), a device-resident loop of one problem -- inside ONE launch.  Until now
// every pass was a search launch plus a one-thread solve launch (5 k -> 20 k, 24 starts: 27 us per pass for 20 us of kernels).
// Here a workgroup runs its problem's passes back to back: search and block reduction as in nn_wave_kernel_one, the ticket
// fold of its problem, and the workgroup that completes the fold advances the problem's state (advance_state: the code of
// solve_state_kernel) and hands the next transform to the problem's other workgroups through 25 self-validating words in
// device memory (kernels.h: FoldArgs::sweep_relay) -- nobody fences, every wait is bounded by the wall clock.  Problems
// advance independently; a problem that stops frees its workgroups.  Same results as one launch per pass, bit for bit (same
// search, same fold order, same solve).  Needs every workgroup resident (the launcher's caller checks the capacity); a wait
// that runs out sets *dead and everybody leaves: the states hold what was completed, the host carries on with launches.
template <bool PLANE>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void nn_wave_kernel_sweep(const SweepParams P)
{
    // the transform of the pass that is due: 32-bit halves of its 12 doubles + the command word, in LDS
    __shared__ unsigned s_w[32];
    const int prob = (int)blockIdx.x / P.bpp;
    {
        // pass 0: the state as the launches before this one left it
        const DevIcpState *sp = P.st + prob;
        if (!sp->active) return;
        if (threadIdx.x < 12) {
            const unsigned long long b = (unsigned long long)__double_as_longlong(sp->Tc[threadIdx.x]);
            s_w[2 * threadIdx.x] = (unsigned)b;
            s_w[2 * threadIdx.x + 1] = (unsigned)(b >> 32);
        }
    }
    __syncthreads();
    typedef const SweepParams __attribute__((address_space(4))) *KernargPtr;
    for (int pass = 0;; pass++) {
        KernargPtr kp = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));                         // (nothing read through it is loop-invariant to the compiler)
#define VISMA_KARG(F_) ld_karg_w(&kp->F_)
        Xform64 T64;
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)s_w[2 * k]);
            const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)s_w[2 * k + 1]);
            T64.m[k] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        }
        DevIcpState *st = VISMA_KARG(st);
        const int bpp = VISMA_KARG(bpp);
        const int pr = (int)blockIdx.x / bpp;
        Offset64 off;
        {
            const DevIcpState *sp = st + pr;                 // (constants of the loop: never written after its start)
#pragma unroll
            for (int a = 0; a < 3; a++) off.v[a] = sp->world_frame ? sp->centre[a] : 0.0;
        }
        const FoldArgs nofold{};                             // (wave_body reads the fold's arguments itself: SWEEP)
        (void)wave_body<PLANE, true, true>(VISMA_KARG(ns), VISMA_KARG(s12f), VISMA_KARG(start), VISMA_KARG(g), nullptr, T64, off,
                                           VISMA_KARG(r2f), VISMA_KARG(idx_out), VISMA_KARG(d2_out), VISMA_KARG(partials),
                                           VISMA_KARG(cand_count), nullptr, bpp, VISMA_KARG(out_stride), nullptr, VISMA_KARG(nprob),
                                           VISMA_KARG(src64), VISMA_KARG(sorted64), nullptr, nofold, nullptr, VISMA_KARG(prevq_io), 1,
                                           pass);
        kp = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        const SweepArgs sa = VISMA_KARG(sa);
#undef VISMA_KARG
        if (pass + 1 >= sa.max_passes) break;
        // ---- the problem's next transform: the first wave polls the problem's 25 words for the tag of the pass that is due
        __syncthreads();                                     // (everybody has read s_w)
        if (threadIdx.x < 64) {
            const int lane = (int)threadIdx.x;
            const unsigned tag = sa.tag0 + (unsigned)pass + 1u;
            unsigned long long w = 0ull;
            const long long t0 = (long long)wall_clock64();
            const int pr2 = (int)blockIdx.x / ld_karg_w(&kp->bpp);
            for (;;) {
                if (lane < kPersistWords) w = __hip_atomic_load(sa.relay + 32ll * pr2 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ok = lane >= kPersistWords || (unsigned)(w >> 32) == tag;
                if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) break;
                const bool dead = __hip_atomic_load(sa.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull;
                if (dead || (long long)wall_clock64() - t0 > sa.wait_ticks) {
                    // (somebody's workgroups are not running, or left: no fold of this launch can complete any more)
                    if (lane == 0) __hip_atomic_store(sa.dead, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    w = (unsigned long long)kPersistAbort;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (lane < 32) s_w[lane] = (unsigned)w;
        }
        __syncthreads();
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)s_w[kPersistWords - 1]) != kPersistGo) break;
    }
}

int nn_wave_sweep_capacity()
{
    static int cap[64];
    static bool known[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 0; }
    if (!known[dev]) {
        int per_cu = 0, cus = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, nn_wave_kernel_sweep<false>, kBlock, 0);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) { (void)hipGetLastError(); per_cu = 0; cus = 0; }
        cap[dev] = per_cu * cus;
        known[dev] = true;
    }
    return cap[dev];
}

hipError_t launch_nn_wave_sweep(int bpp, int nprob, int ns, const float *s12, const unsigned *start, const GridParams &g,
                                float r2f, int32_t *idx_out, float *d2_out, double *partials, unsigned long long *cand_count,
                                DevIcpState *st, long long out_stride, const Pt64 *src64, const Pt64 *sorted64,
                                const FoldArgs &fold, Pt64 *prevq_io, const SweepArgs &sa, hipStream_t stream)
{
    if (!src64 || !sorted64 || !s12 || !prevq_io || !st || !fold.tickets || !sa.relay || !sa.dead || sa.max_passes < 1 ||
        bpp < 1 || nprob < 1 || (long long)bpp * kBlock < ns || bpp * nprob > nn_wave_sweep_capacity())
        return hipErrorInvalidValue;
    SweepParams P{};
    P.ns = ns; P.s12f = s12; P.start = start; P.g = g; P.r2f = r2f; P.idx_out = idx_out; P.d2_out = d2_out; P.partials = partials;
    P.cand_count = cand_count; P.st = st; P.bpp = bpp; P.out_stride = out_stride; P.nprob = nprob; P.src64 = src64;
    P.sorted64 = sorted64; P.fold = fold; P.prevq_io = prevq_io; P.sa = sa;
    hipLaunchKernelGGL(nn_wave_kernel_sweep<false>, dim3(bpp * nprob), dim3(kBlock), 0, stream, P);
    return hipGetLastError();
}

#define VISMA_WAVE_LAUNCH(KERNEL_)                                                                               \
    hipLaunchKernelGGL(KERNEL_, dim3(total_blocks), dim3(kBlock), 0, stream, ns, s12, start, g, nrm, T64, off,   \
                       r2f, idx_out, d2_out, partials, cand_count, st, bpp, out_stride, descs, nprob, src64,     \
                       sorted64, nrm64, fold, d64_out, prevq_io, warm)

// The warm-started, flattened exact search.  Shared clouds: `nprob` problems of `bpp` workgroups each
// (descs == NULL); own clouds: descs[nprob], total_blocks workgroups.  `one`: at most one query per lane.
// warm & 2: a workgroup -> problem map (int per workgroup) follows descs[nprob].
// prevq_io (one float4 per query, laid out like idx_out): read when `warm & 1` (the winners of the previous pass
// over the SAME source order and target, as fp32 points of the candidate array; NaN = none), always written.
hipError_t launch_nn_wave(int total_blocks, int bpp, int nprob, const ProbDesc *descs, int ns, const float *s12,
                          const unsigned *start, const GridParams &g, const float4 *nrm, const Pt64 *nrm64,
                          const Xform64 &T64, const Offset64 &off, float r2f, int point_to_plane, int one,
                          int32_t *idx_out, float *d2_out, double *partials, unsigned long long *cand_count,
                          const DevIcpState *st, long long out_stride, const Pt64 *src64, const Pt64 *sorted64,
                          const FoldArgs &fold, double *d64_out, Pt64 *prevq_io, int warm, hipStream_t stream)
{
    if (!src64 || !sorted64 || !s12 || !prevq_io) return hipErrorInvalidValue;
    if (point_to_plane) {
        if (one) VISMA_WAVE_LAUNCH(nn_wave_kernel_one<true>); else VISMA_WAVE_LAUNCH(nn_wave_kernel_many<true>);
    } else {
        if (one) VISMA_WAVE_LAUNCH(nn_wave_kernel_one<false>); else VISMA_WAVE_LAUNCH(nn_wave_kernel_many<false>);
    }
    return hipGetLastError();
}
#undef VISMA_WAVE_LAUNCH

}  // namespace visma
