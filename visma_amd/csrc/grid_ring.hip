// grid_ring.hip -- the search for radii that are LARGE against the target's point spacing (round 6).
//
// Every other grid kernel of this library searches the 27 cells around a query in a grid whose cells are as large as the
// radius (KDTreeFlann.cpp:164-189: SearchHybrid(query, radius, 1)).  That is the right shape while a radius-sized cell holds
// a handful of points (C4: ~10).  A caller whose radius spans tens of point spacings -- SURVEY 8d's literal motion needs
// r = 0.15 m on a target sampled every 2 mm: 2,800 points per occupied cell, 15,000 in a query's ball -- would scan tens
// of thousands of candidates per query for a nearest neighbour that is centimetres away.
//
// Here the cell edge is decoupled from the radius (GridParams::ring > 0: cells of a few point spacings, chosen by
// HipEngine::build_grid from the occupancy of the radius-sized grid) and a query visits the rows (y, z) of the cell table
// around its own row NEAREST FIRST -- ring by ring; the order is the same for every query, so it is a table of offsets
// sorted by their lower bound (ring_visiting_order) --, each row only over the x-extent that can still hold a candidate nearer
// than the best so far, and stops when the next row's bound lies beyond the best.  The best so far starts at the radius -- or, for
// every pass after the first, at the distance to the previous pass's winner under the new transform (the state of the
// cooperative kernels, same buffer), so a converging registration visits the few cells around its winner.  The cost of a
// query follows the number of target points nearer than its nearest neighbour's distance, not the radius.
//
// Exactness: candidates are the caller's f64 points (32 B each), the source is transformed in f64, d2 is the reference's
// f64 sum of squares in x, y, z order (flann dist.h:159-176), acceptance is d2 < (double)(float)(r*r)
// (KDTreeFlann.cpp:184-185), ties go to the lowest original index -- the arithmetic of nn_grid_reduce_kernel<F64>.  Rows
// and cells are pruned with the fp32 view and that kernel's margins (1e-3 cell on every slab distance, 1e-5 relative on
// the squared bound, the bound rounded up), strictly: a pruned candidate is strictly farther than the best, so it could
// neither win nor tie.  With the packed fp32 copy of the target (12 B per candidate; the exact search's default) the walk
// ranks in fp32 and keeps, per lane, the two best and the value of the third; the candidates inside the rounding band of
// the octet's best (nn_grid_reduce_kernel<HYB>'s band) are then ranked in f64, and a lane that saw a third candidate inside
// the band sends its query through the f64 walk -- the result is the f64 walk's either way.
// G lanes work on one query (eight; four or two once a host loop's registration has converged to within a cell or so:
// HipEngine::note_ring_stats): each looks up R rows of the visiting order per step and scans their (short) x-ranges
// itself -- sixteen independent streams per query; a long range (first rows of a cold pass, rows inside a surface) is
// scanned by the group, G candidates per step and two in flight; the partial minima meet in a butterfly on (d2, index).
// One query per lane group wherever the launch may have that many workgroups: the sums are formed after the search.
// Statistics, outputs, fold: as nn_grid_reduce_kernel (same accumulators, block_reduce_store, fused_fold).
#include "device_common.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace visma {

namespace {

// (lanes per query G and rows of the visiting order per lane and step R are template parameters: launch_nn_ring)
#ifndef VISMA_RING_XP
#define VISMA_RING_XP 0     /* timing experiments only (wrong results): 1 no sums / block reduction, 2 no walk, 4 no fold */
#endif
#ifndef VISMA_RING_ROWS_LDS
#define VISMA_RING_ROWS_LDS 0
#endif
#ifndef VISMA_RING_LOCAL
#define VISMA_RING_LOCAL 4
#endif
#ifndef VISMA_RING_V
#define VISMA_RING_V 2
#endif
#ifndef VISMA_RING_U
#define VISMA_RING_U 2
#endif
constexpr int kRingLocal = VISMA_RING_LOCAL;    // a range up to this long is scanned by the lane that looked it up (4: measured,
                                                // 240 -> 201 us per iteration against 16; 0 = every range by the octet: 204) ...
constexpr int kRingV = VISMA_RING_V;                       // ... that many fp32 candidates of it in flight
#if VISMA_RING_ROWS_LDS
constexpr int kRingRowsLds = 512;               // rows of the visiting order kept in LDS (4 KB; measured neutral: off)
#endif
constexpr int kRingU = VISMA_RING_U;                       // 32-byte candidates in flight per lane (ranges scanned by the octet)
constexpr unsigned kRingNone = 0xFFFFFFFFu;     // no winner (the largest index: loses every tie)
constexpr unsigned kRingState = 0xFFFFFFFEu;    // the winner is the point the state holds (no slot known)

struct P12 { float x, y, z; };                  // fp32 rounding of a cell-sorted f64 target point (launch_pack12)

// the two best (d2, slot) a lane has seen and the value of the third: every candidate not recorded is at least h2 away
struct Top2 {
    float h0, h1, h2;
    unsigned p0, p1;
    __device__ __forceinline__ void init(float lim)
    {
        h0 = h1 = h2 = lim;
        p0 = p1 = kRingNone;
    }
    __device__ __forceinline__ void insert(float d, unsigned pos)
    {
        const bool c0 = d < h0, c1 = d < h1, c2 = d < h2;
        h2 = c1 ? h1 : (c2 ? d : h2);
        p1 = c0 ? p0 : (c1 ? pos : p1);
        h1 = c0 ? h0 : (c1 ? d : h1);
        p0 = c0 ? pos : p0;
        h0 = c0 ? d : h0;
    }
};

// ONE: every octet has at most one query (the launcher's geometry): the sums are formed after the search, nothing of
// them is live during it.
#ifdef VISMA_RING_WAVES   /* occupancy experiments: waves per SIMD the register allocation must allow */
#define VISMA_RING_OCCUPANCY __attribute__((amdgpu_waves_per_eu(VISMA_RING_WAVES, VISMA_RING_WAVES)))
#else
#define VISMA_RING_OCCUPANCY
#endif
template <bool PLANE, bool ONE, int kRingG, int kRingR>
__global__ __launch_bounds__(kBlock) VISMA_RING_OCCUPANCY void nn_ring_kernel(
    int ns, const Pt64 *__restrict__ src64, const Pt64 *__restrict__ sorted64, const P12 *__restrict__ s12,
    const unsigned *__restrict__ start, const GridParams g, const RingRow *__restrict__ tab, int ring_rows,
    const float4 *__restrict__ nrm, const Pt64 *__restrict__ nrm64, Xform64 T64, Offset64 off, float r2f,
    int *__restrict__ idx_out, float *__restrict__ d2_out, double *__restrict__ d64_out, Pt64 *__restrict__ state_io, int warm,
    double *__restrict__ partials, unsigned long long *__restrict__ cand_count, const DevIcpState *__restrict__ st, int bpp,
    long long out_stride, const FoldArgs fold)
{
    constexpr int NACC = Acc<PLANE>::N;
    const int prob = (int)blockIdx.x / bpp;
    const int lb = (int)blockIdx.x - prob * bpp;
    const long long row0 = (long long)prob * bpp;
    if (st) st += prob;
    {
        Xform32 T32_unused;
        if (!load_loop_state(st, T32_unused, T64, off, r2f)) return;
    }
    const double r2d = (double)r2f;                         // (double)(float)(r*r): KDTreeFlann.cpp:184-185
    idx_out += (long long)prob * out_stride;
    d2_out += (long long)prob * out_stride;
    if (state_io) state_io += (long long)prob * out_stride;
    unsigned long long ncand = 0ull, nrows_seen = 0ull;

    static_assert(kRingG == 1 || kRingG == 2 || kRingG == 4 || kRingG == 8, "lanes per query");
    constexpr unsigned long long kGroupMask = (1ull << kRingG) - 1ull;
    const int tid = (int)threadIdx.x, lane = tid & 63, l8 = lane & (kRingG - 1), obase = lane & ~(kRingG - 1);
    // the query -> octet map: nn_grid_reduce_kernel's (one contiguous eighth of the Morton-ordered queries per XCD)
    int vb = lb;
    if ((bpp & 7) == 0) vb = (lb & 7) * (bpp >> 3) + (lb >> 3);
    constexpr int kOctets = kBlock / kRingG;
    const long long total_groups = (long long)bpp * kOctets;
    const long long per_group = ((long long)ns + total_groups - 1) / total_groups;
    const long long gid = (long long)vb * kOctets + tid / kRingG;
    const long long i_begin = gid * per_group;
    const long long i_end = i_begin + per_group < (long long)ns ? i_begin + per_group : (long long)ns;

    const int K = g.ring;
    const float h2 = g.h * g.h * (1.0f - 1e-5f);
    const float inv_h2 = 1.0f / h2;
    const float mgn = 1e-3f;                               // fp32 binning of query and candidates (kGridMaxDim)
    const float reach = (float)(K + 1);                    // cells: farther outside the table than this = no partner

#if VISMA_RING_ROWS_LDS
    // the head of the visiting order (what a converging registration reads) in LDS
    __shared__ RingRow s_rows[kRingRowsLds];
    for (int n = tid; n < kRingRowsLds && n < ring_rows; n += kBlock) s_rows[n] = tab[n];
    __syncthreads();
#endif
#if VISMA_RING_ROWS_LDS
    auto row_of = [&](int n) { return n < kRingRowsLds ? s_rows[n] : tab[n]; };
#else
    auto row_of = [&](int n) { return tab[n]; };        // (consecutive lanes, consecutive entries: the caches hold its head)
#endif

    // ---- one query: its transformed point (pd), its winner (w8; false = none); outputs written by the octet's first lane
    auto search = [&](long long i, double (&pd)[3], Pt64 &w8) -> bool {
        // the reference's transform of a source point (PointCloud.cpp:75-80), in f64
        const Pt64 s8 = src64[i];
        const double pxd = T64.m[0] * s8.x + T64.m[1] * s8.y + T64.m[2] * s8.z + T64.m[3] * 1.0;
        const double pyd = T64.m[4] * s8.x + T64.m[5] * s8.y + T64.m[6] * s8.z + T64.m[7] * 1.0;
        const double pzd = T64.m[8] * s8.x + T64.m[9] * s8.y + T64.m[10] * s8.z + T64.m[11] * 1.0;
        pd[0] = pxd; pd[1] = pyd; pd[2] = pzd;
        const float px = (float)pxd, py = (float)pyd, pz = (float)pzd;
        double bd = r2d;                                   // best d2 so far (strictly below r2d once set) ...
        unsigned bidx = kRingNone, bpos = kRingNone;       // ... its original index, its slot
        auto rank = [&](const Pt64 &c8, unsigned pos) {
            // flann L2 (dist.h:159-176): result += diff * diff over x, y, z
            const double dx = c8.x - pxd, dy = c8.y - pyd, dz = c8.z - pzd;
            double d = dx * dx;
            d += dy * dy;
            d += dz * dz;
            const unsigned id = (unsigned)c8.w;
            // strictly nearer, or as near with a lower index (a re-read of the winner is neither)
            const bool lt = d < bd || (d == bd && id < bidx && bidx != kRingNone);
            bd = lt ? d : bd;
            bidx = lt ? id : bidx;
            bpos = lt ? pos : bpos;
        };
        // ---- warm start: the previous pass's winner is a point of this target -- a candidate like any other, known
        // before anything is looked up (every lane of the octet holds it: the bound is octet-uniform from the start)
        Pt64 prev = Pt64{0.0, 0.0, 0.0, ~0ull};
        if (warm && state_io) prev = state_io[i];
        if ((unsigned)prev.w != kRingNone) rank(prev, kRingState);
        const double bd0 = bd;
        const unsigned bidx0 = bidx, bpos0 = bpos;

        // fp32 ranking (the packed 12-byte copy): rounding band as in nn_grid_reduce_kernel<HYB> -- p32 = fl(p64),
        // q32 = fl(q64), E more than twice the bound of |d64 - sqrt(d2_32)| for candidates within the limit
        const float r_f = sqrtf(r2f);
        const float rup = r_f * (1.0f + 2.4e-7f);
        const float E = 2.4e-7f * (fabsf(px) + fabsf(py) + fabsf(pz) + rup) + 4.8e-7f * rup;
        Top2 top;
        top.init(INFINITY);
        // candidates at or beyond the limit never matter: the radius, or the previous winner's distance
        const float sp = fminf(rup, sqrtf((float)bd) * (1.0f + 2.4e-7f)) + 2.0f * E;
        const float lim = sp * sp * (1.0f + 6e-7f);

        const float ux = (px - g.mn[0]) * g.inv_h, uy = (py - g.mn[1]) * g.inv_h, uz = (pz - g.mn[2]) * g.inv_h;
        // (written so that a NaN query is outside)
        const bool inside = ux >= -reach && ux <= (float)g.dim[0] + reach && uy >= -reach && uy <= (float)g.dim[1] + reach &&
                            uz >= -reach && uz <= (float)g.dim[2] + reach;
        const float flx = floorf(ux), fly = floorf(uy), flz = floorf(uz);
        const int cx = (int)flx, cy = (int)fly, cz = (int)flz;
        const float fx = ux - flx, fy = uy - fly, fz = uz - flz;          // position inside the cell, [0, 1)

        // ---- the walk over the rows, ring by ring, sixteen rows of the sequence per step.  F32: candidates are the packed
        // fp32 points, kept as the two best per lane + the value of the third; else: the f64 points, ranked at once.
        auto walk = [&](auto f32_tag) {
            constexpr bool F32 = decltype(f32_tag)::value;
            // fp32 binning of query and candidates (kGridMaxDim) [+ the rounding band]
            const float mg = F32 ? mgn + 4.0f * E * g.inv_h : mgn;
            const float lo_x = fmaxf(fx - mg, 0.f), hi_x = fmaxf(1.0f - fx - mg, 0.f);
            const float lo_y = fmaxf(fy - mg, 0.f), hi_y = fmaxf(1.0f - fy - mg, 0.f);
            const float lo_z = fmaxf(fz - mg, 0.f), hi_z = fmaxf(1.0f - fz - mg, 0.f);
            // what a candidate must not exceed to matter, as a squared fp32 distance rounded UP, the octet's smallest
            auto octet_best = [&]() {
                float f;
                if constexpr (F32) {
                    const float s1 = sqrtf(top.h0) + 2.0f * E;           // (nothing seen yet: inf)
                    f = fminf(s1 * s1 * (1.0f + 4e-7f), lim);
                } else {
                    f = (float)bd;
                    f += f * 1e-6f;
                }
#pragma unroll
                for (int m = kRingG >> 1; m > 0; m >>= 1) f = fminf(f, __shfl_xor(f, m, 64));
                return f;
            };
            float gbest = octet_best();
            const int total = ring_rows;
#pragma unroll 1
            for (int n0 = 0; n0 < total; n0 += kRingG * kRingR) {
                // the rows come nearest first: every row from here on is at least sqrt(base) cells away
                if (row_of(n0).base * h2 > gbest) break;
                // ---- kRingR rows of the sequence per lane: slab bound, x-extent, range of the sorted target
                unsigned rb[kRingR], re[kRingR];
#pragma unroll
                for (int u = 0; u < kRingR; u++) {
                    const int n = n0 + u * kRingG + l8;
                    const RingRow rr = row_of(n < total ? n : 0);
                    const int dy = rr.dy, dz = rr.dz;
                    const float ey = dy == 0 ? 0.f : (dy < 0 ? lo_y + (float)(-dy - 1) : hi_y + (float)(dy - 1));
                    const float ez = dz == 0 ? 0.f : (dz < 0 ? lo_z + (float)(-dz - 1) : hi_z + (float)(dz - 1));
                    const float bound = (ey * ey + ez * ez) * h2;
                    const int y = cy + dy, z = cz + dz;
                    const bool ok = n < total && y >= 0 && y < g.dim[1] && z >= 0 && z < g.dim[2] && !(bound > gbest);
                    rb[u] = re[u] = 0u;
                    if (ok) {
                        // cells of the row that can hold a point within the bound: x-cell cx + d (d >= 1) is at least
                        // hi_x + d - 1 cells away, cx - d at least lo_x + d - 1 (one cell more never hurts)
                        const float t = sqrtf(fmaxf(gbest - bound, 0.f) * inv_h2) * (1.0f + 1e-5f) + 1e-4f;
                        const int dxp = min(max((int)floorf(t - hi_x + 1.0f), 0), K + 1);
                        const int dxm = min(max((int)floorf(t - lo_x + 1.0f), 0), K + 1);
                        const int x0 = max(cx - dxm, 0), x1 = min(cx + dxp, g.dim[0] - 1);
                        if (x0 <= x1) {
                            const unsigned row = (unsigned)((z * g.dim[1] + y) * g.dim[0]);
                            rb[u] = start[row + (unsigned)x0];
                            re[u] = start[row + (unsigned)x1 + 1u];
                            nrows_seen += 1ull;
                        }
                    }
                }
                // ---- short ranges (nearly all: a row of a few cells holds a few points): each lane scans its own,
                // its rows side by side -- sixteen independent streams per octet, nothing crosses lanes
                {
                    unsigned b[kRingR], e[kRingR];
                    bool more = false;
#pragma unroll
                    for (int u = 0; u < kRingR; u++) {
                        const bool local = re[u] - rb[u] <= (unsigned)kRingLocal;
                        b[u] = rb[u];
                        e[u] = local ? re[u] : rb[u];
                        ncand += (unsigned long long)(e[u] - b[u]);
                        more = more || b[u] < e[u];
                    }
#pragma unroll 1
                    while (more) {
                        if constexpr (F32) {
                            // (kRingV candidates of each row in flight: a range of kRingLocal is worked off in a few trips)
                            P12 q[kRingR][kRingV];
#pragma unroll
                            for (int u = 0; u < kRingR; u++)
#pragma unroll
                                for (int v = 0; v < kRingV; v++) q[u][v] = s12[b[u] + (unsigned)v < e[u] ? b[u] + (unsigned)v : rb[0]];
                            more = false;
#pragma unroll
                            for (int u = 0; u < kRingR; u++) {
#pragma unroll
                                for (int v = 0; v < kRingV; v++) {
                                    const float d = sqdist_f32(make_float4(q[u][v].x, q[u][v].y, q[u][v].z, 0.f), px, py, pz);
                                    const bool real = b[u] + (unsigned)v < e[u];
                                    top.insert((real && d < lim) ? d : INFINITY, b[u] + (unsigned)v);
                                }
                                b[u] = min(b[u] + (unsigned)kRingV, e[u]);
                                more = more || b[u] < e[u];
                            }
                        } else {
                            Pt64 q[kRingR];
#pragma unroll
                            for (int u = 0; u < kRingR; u++) q[u] = sorted64[b[u] < e[u] ? b[u] : rb[0]];
                            more = false;
#pragma unroll
                            for (int u = 0; u < kRingR; u++) {
                                if (b[u] < e[u]) { rank(q[u], b[u]); b[u]++; }
                                more = more || b[u] < e[u];
                            }
                        }
                    }
                }
                // ---- long ranges (the first rows of a cold pass, rows that run inside a surface): the octet scans them
                // one after the other, eight candidates per step
#pragma unroll
                for (int u = 0; u < kRingR; u++) {
                    unsigned m8 = (unsigned)(__builtin_amdgcn_ballot_w64(re[u] - rb[u] > (unsigned)kRingLocal) >> obase) & (unsigned)kGroupMask;
#pragma unroll 1
                    while (m8) {
                        const int j = __builtin_ctz(m8);
                        m8 &= m8 - 1u;
                        const unsigned b = (unsigned)__shfl((int)rb[u], obase + j, 64);
                        const unsigned e = (unsigned)__shfl((int)re[u], obase + j, 64);
                        if (l8 == 0) ncand += (unsigned long long)(e - b);
#pragma unroll 1
                        for (unsigned base = b; base < e; base += kRingG * kRingU) {
                            if constexpr (F32) {
                                P12 q[kRingU];
                                unsigned jc[kRingU];
#pragma unroll
                                for (int v = 0; v < kRingU; v++) {
                                    jc[v] = base + (unsigned)(l8 + v * kRingG);
                                    q[v] = s12[jc[v] < e ? jc[v] : b];
                                }
#pragma unroll
                                for (int v = 0; v < kRingU; v++) {
                                    float d = sqdist_f32(make_float4(q[v].x, q[v].y, q[v].z, 0.f), px, py, pz);
                                    d = (jc[v] < e && d < lim) ? d : INFINITY;   // (a padding slot is not a candidate)
                                    top.insert(d, jc[v]);
                                }
                            } else {
                                Pt64 q[kRingU];
                                unsigned jc[kRingU];
#pragma unroll
                                for (int v = 0; v < kRingU; v++) {
                                    // a slot past the end re-reads the range's first point: evaluating a candidate twice
                                    // cannot change the (d2, index) minimum
                                    const unsigned ju = base + (unsigned)(l8 + v * kRingG);
                                    jc[v] = ju < e ? ju : b;
                                    q[v] = sorted64[jc[v]];
                                }
#pragma unroll
                                for (int v = 0; v < kRingU; v++) rank(q[v], jc[v]);
                            }
                        }
                    }
                }
                gbest = octet_best();
            }
        };
        if (inside) {
            bool exact_walk = s12 == nullptr;
            if (!exact_walk) {
#if !(VISMA_RING_XP & 2)
                walk(std::true_type{});
#endif
                // ---- decisive?  every candidate inside the rounding band of the octet's best is ranked in f64; a lane
                // that saw a third one inside the band does not know its position: the query is searched again in f64
                float m = top.h0;
#pragma unroll
                for (int k = kRingG >> 1; k > 0; k >>= 1) m = fminf(m, __shfl_xor(m, k, 64));
                const float s1 = sqrtf(m) + 2.0f * E;
                const float s1sq = s1 * s1 * (1.0f + 4e-7f);
                const bool over = top.h2 < INFINITY && top.h2 <= s1sq;   // (inf: no third candidate)
                exact_walk = ((__builtin_amdgcn_ballot_w64(over) >> obase) & kGroupMask) != 0ull;
                if (!exact_walk) {
                    const bool in0 = top.h0 <= s1sq && top.h0 < INFINITY, in1 = top.h1 <= s1sq && top.h1 < INFINITY;
                    Pt64 c0 = Pt64{0.0, 0.0, 0.0, 0ull}, c1 = Pt64{0.0, 0.0, 0.0, 0ull};
                    if (in0) c0 = sorted64[top.p0];
                    if (in1) c1 = sorted64[top.p1];
                    if (in0) rank(c0, top.p0);
                    if (in1) rank(c1, top.p1);
                }
            }
            if (exact_walk) {
                bd = bd0; bidx = bidx0; bpos = bpos0;
                walk(std::false_type{});
            }
        }
        // ---- butterfly over the octet: smallest (d2, index) wins everywhere (none = the largest index: loses every tie)
#pragma unroll
        for (int m = kRingG >> 1; m > 0; m >>= 1) {
            const double od = __shfl_xor(bd, m, 64);
            const unsigned oi = (unsigned)__shfl_xor((int)bidx, m, 64);
            const unsigned op = (unsigned)__shfl_xor((int)bpos, m, 64);
            if (od < bd || (od == bd && oi < bidx)) { bd = od; bidx = oi; bpos = op; }
        }
        bpos = (unsigned)__shfl((int)bpos, obase, 64);       // (lanes that tie on (d2, index) may hold different slots of it)
        const bool found = bpos != kRingNone;
        w8 = prev;
        if (found && bpos != kRingState) w8 = sorted64[bpos];
        if (l8 == 0) {
            if (state_io) {
                Pt64 o8;
                o8.x = o8.y = o8.z = __longlong_as_double(-1ll);
                o8.w = ~0ull;
                if (found) { o8 = w8; o8.w = w8.w & 0xFFFFFFFFull; }
                state_io[i] = o8;
            }
            idx_out[i] = found ? (int)bidx : -1;
            d2_out[i] = (float)bd;
            if (d64_out) d64_out[i] = bd;                    // (target-sharded ranks compare shards in f64)
        }
        return found;
    };
    auto add_pair = [&](double *acc, const double (&pd)[3], const Pt64 &w8) {
        double nx = 0.0, ny = 0.0, nz = 0.0;
        if (PLANE) {
            if (nrm64) { const Pt64 n8 = nrm64[(unsigned)w8.w]; nx = n8.x; ny = n8.y; nz = n8.z; }
            else { const float4 n4 = nrm[(unsigned)w8.w]; nx = n4.x; ny = n4.y; nz = n4.z; }
        }
        accumulate_pq_d<PLANE>(acc, pd[0], pd[1], pd[2], w8.x, w8.y, w8.z, nx, ny, nz, off);
    };

    double acc[NACC];
    if constexpr (ONE) {
        double pd[3] = {0.0, 0.0, 0.0};
        Pt64 w8 = Pt64{0.0, 0.0, 0.0, ~0ull};
        bool found = false;
        if (i_begin < i_end) found = search(i_begin, pd, w8);
#pragma unroll
        for (int a = 0; a < NACC; a++) acc[a] = 0.0;
#if !(VISMA_RING_XP & 1)
        if (found && l8 == 0) add_pair(acc, pd, w8);
#endif
    } else {
#pragma unroll
        for (int a = 0; a < NACC; a++) acc[a] = 0.0;
        int it = 0;
        for (long long i = i_begin; i < i_end; i++, it++) {
            double pd[3];
            Pt64 w8;
            const bool found = search(i, pd, w8);
            // lane (it mod 8) of the octet adds this query's correspondence to its sums
            if (found && (it & (kRingG - 1)) == l8) add_pair(acc, pd, w8);
        }
    }
#if !(VISMA_RING_XP & 1)
    block_reduce_store<NACC>(acc, partials, fold.tickets != nullptr);
#endif
    if (cand_count) {
        // profiling only: candidates examined / rows of the cell table looked up (one slot pair per workgroup mod 4096)
        unsigned long long c = ncand, ca = nrows_seen;
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
#if !(VISMA_RING_XP & 4)
    if (fold.tickets) fused_fold<PLANE, kBlock, false, kSolveInFold>(fold, partials, row0, lb, bpp, prob);
#endif
}

// cells of a table that hold at least one point
__global__ __launch_bounds__(256) void count_occupied_kernel(const unsigned *__restrict__ count, long long n,
                                                             unsigned long long *__restrict__ out)
{
    unsigned long long c = 0ull;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) c += count[i] != 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

}  // namespace

// `lanes` (1, 2, 4, 8) lanes per query, `nblocks` workgroups per problem, `nprob` problems over shared clouds (st: their loop
// states or NULL); g.ring > 0.
// state_io: per query the winner's f64 point and original index (all bits set = none), laid out like idx_out: read when
// `warm` (every entry must be none or a point of THIS target), always written.
hipError_t launch_nn_ring(int lanes, int nblocks, int nprob, int ns, const Pt64 *src64, const Pt64 *sorted64, const float *s12, const unsigned *start,
                          const GridParams &g, const RingTable &tab, const float4 *nrm, const Pt64 *nrm64, const Xform64 &T64,
                          const Offset64 &off, float r2f, int point_to_plane, int32_t *idx_out, float *d2_out, double *d64_out,
                          Pt64 *state_io, int warm, double *partials, unsigned long long *cand_count, const DevIcpState *st,
                          long long out_stride, const FoldArgs &fold, hipStream_t stream)
{
    if (!src64 || !sorted64 || !start || g.ring < 1 || g.sub != 1 || !tab.rows || tab.nrows != (2 * g.ring + 1) * (2 * g.ring + 1) || nblocks < 1 || nprob < 1 ||
        (lanes != 1 && lanes != 2 && lanes != 4 && lanes != 8))
        return hipErrorInvalidValue;
    if (point_to_plane && !nrm && !nrm64) return hipErrorInvalidValue;
    const bool one = (long long)nblocks * (kBlock / lanes) >= (long long)ns;       // at most one query per lane group
#define VISMA_RING_LAUNCH(PLANE_, ONE_, G_, R_)                                                                                  \
    hipLaunchKernelGGL((nn_ring_kernel<PLANE_, ONE_, G_, R_>), dim3(nblocks * nprob), dim3(kBlock), 0, stream, ns, src64,        \
                       sorted64, reinterpret_cast<const P12 *>(s12), start, g, tab.rows, tab.nrows, nrm, nrm64, T64, off, r2f, idx_out, d2_out,       \
                       d64_out, state_io, warm, partials, cand_count, st, nblocks, out_stride, fold)
#define VISMA_RING_CASE(G_, R_)                                                                                                  \
    if (lanes == G_) {                                                                                                           \
        if (point_to_plane) { if (one) VISMA_RING_LAUNCH(true, true, G_, R_); else VISMA_RING_LAUNCH(true, false, G_, R_); }     \
        else { if (one) VISMA_RING_LAUNCH(false, true, G_, R_); else VISMA_RING_LAUNCH(false, false, G_, R_); }                  \
    }
#ifndef VISMA_RING_R8
#define VISMA_RING_R8 2     /* rows of the visiting order per lane and step with eight lanes per query (A/B: 1, 3, 4) */
#endif
    VISMA_RING_CASE(8, VISMA_RING_R8) VISMA_RING_CASE(4, 4) VISMA_RING_CASE(2, 4) VISMA_RING_CASE(1, 8)
#undef VISMA_RING_CASE
#undef VISMA_RING_LAUNCH
    return hipGetLastError();
}

// The visiting order of the rows around a query for `rings` rings: every offset (dy, dz) of the square, sorted by the squared
// distance (in cells) that a point of the row is at least away from ANY point of the query's own row of cells -- the same
// for every query, so the order is a table.  Ties: by |dy| + |dz|, then dy, then dz (any fixed order would do).
std::vector<RingRow> ring_visiting_order(int rings)
{
    std::vector<RingRow> rows;
    if (rings < 1 || rings > kRingMaxRings) return rows;
    const int side = 2 * rings + 1;
    rows.reserve((size_t)side * side);
    for (int dz = -rings; dz <= rings; dz++)
        for (int dy = -rings; dy <= rings; dy++) {
            const int ay = std::max(std::abs(dy) - 1, 0), az = std::max(std::abs(dz) - 1, 0);
            rows.push_back(RingRow{(short)dy, (short)dz, (float)(ay * ay + az * az)});
        }
    std::sort(rows.begin(), rows.end(), [](const RingRow &a, const RingRow &b) {
        if (a.base != b.base) return a.base < b.base;
        const int ma = std::abs((int)a.dy) + std::abs((int)a.dz), mb = std::abs((int)b.dy) + std::abs((int)b.dz);
        if (ma != mb) return ma < mb;
        if (a.dy != b.dy) return a.dy < b.dy;
        return a.dz < b.dz;
    });
    return rows;
}

hipError_t build_ring_table(int rings, void **d_tab, int *nrows)
{
    if (rings < 1 || rings > kRingMaxRings || !d_tab || !nrows) return hipErrorInvalidValue;
    const std::vector<RingRow> rows = ring_visiting_order(rings);
    void *d = nullptr;
    hipError_t e = hipMalloc(&d, sizeof(RingRow) * rows.size());
    if (e != hipSuccess) return e;
    e = hipMemcpy(d, rows.data(), sizeof(RingRow) * rows.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return e; }
    *d_tab = d;
    *nrows = (int)rows.size();
    return hipSuccess;
}

// *out (device, zeroed by the caller) += the number of entries of count[0 .. n) that are not zero
hipError_t launch_count_occupied(const unsigned *count, int64_t n, unsigned long long *out, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(count_occupied_kernel, dim3(blocks), dim3(256), 0, stream, count, (long long)n, out);
    return hipGetLastError();
}

}  // namespace visma
