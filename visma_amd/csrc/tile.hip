// tile.hip -- EXPERIMENTAL (off by default, VISMA_ICP_TILE=1): the LDS-streamed radius-cell
// search.  Same contract and same answers as nn_grid_reduce_kernel's exact flavour (grid.hip),
// different memory strategy; measured 2.4-3x SLOWER at the densities of BASELINE.json
// (profiles/r02_probe_tile6.txt) and kept for radii >> point spacing and as
// the record of the experiment.  ONE kernel per ICP iteration; a workgroup owns a
// Morton-contiguous block of queries (G lanes per query) and
//   (A) transforms them (the reference's f64 transform, PointCloud.cpp:75-80), computes their
//       cell coordinates and enters the (y,z) rows they touch into a hashed row table in LDS,
//   (B) fetches the run bounds of every live row once, prefix-sums the run lengths and
//       compacts them into one LDS point array; a footprint above the LDS budget (CAP points,
//       MAXR rows) is split into passes over subsets of the rows,
//   (C) streams the runs into LDS with coalesced loads, 16 lanes per row, as 12-byte SoA
//       points (each target point is read once per workgroup, not once per query),
//   (D) searches every query's 3x3 rows FROM LDS in fp32 (Top3 + 4th value, slab pruning with
//       a live-row mask),
//   (E) re-ranks the rounding band in the reference's f64 arithmetic (flann dist.h:159-176,
//       KDTreeFlann.cpp:184-185), exactly as grid.hip does,
//   (F) forms the moments of the winner from the f64 coordinates, reduces them through LDS and
//       folds the per-workgroup rows in the same launch (device_common.h: fused_fold).
// Replaces KDTreeFlann::SearchHybrid + GetRegistrationResultAndCorrespondences
// (O3D/Core/Registration/Registration.cpp:41-96) + the estimator's accumulation
// (src/constrained_ICP.cpp:25-37, O3D/Core/Utility/Eigen.cpp:137-182).
// Template configurations: <NTH threads, G lanes/query, CAP points, MAXR rows>; STAMPS adds
// per-phase s_memtime stamps (visma_icp_timing.tile_phase_cycles).
#include "device_common.h"

#include <limits.h>

#include <type_traits>

namespace visma {

namespace {

constexpr unsigned kNone = 0xFFFFFFFFu;
constexpr int kGroupRows = 32;          // partial rows folded by one level-1 last arriver

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ float med3_f32(float a, float b, float c)
{
    float r;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// visiting order of the 9 (dy, dz) rows: centre, the 4 edge neighbours, the 4 corners.
// k = (dz + 1) * 3 + (dy + 1)
__device__ __forceinline__ int row_of_visit(int kk)
{
    // {4, 1, 3, 5, 7, 0, 2, 6, 8} packed 4 bits each
    return (int)((0x862075314ull >> (4 * kk)) & 15ull);
}

// lane <-> lane exchanges inside groups of 2 / 4 / 8 lanes as DPP moves (no LDS crossbar trip)
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
constexpr int kDppXor1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;   // row_half_mirror: lane i <-> 7 - i of each 8

// minimum over the G lanes of a query (G = 1, 2, 4, 8), every lane gets it
template <int G>
__device__ __forceinline__ float group_min(float v)
{
    if (G >= 8) v = fminf(v, dpp_f32<kDppHalfMirror>(v));
    if (G >= 4) v = fminf(v, dpp_f32<kDppXor2>(v));
    if (G >= 2) v = fminf(v, dpp_f32<kDppXor1>(v));
    return v;
}

// (best, runner-up, position of the best) over the G lanes of a query
template <int CTRL>
__device__ __forceinline__ void merge_step(float &b1, float &b2, unsigned &pos)
{
    const float o1 = dpp_f32<CTRL>(b1), o2 = dpp_f32<CTRL>(b2);
    const unsigned op = dpp_u32<CTRL>(pos);
    const float n2 = fminf(fmaxf(b1, o1), fminf(b2, o2));
    // equal d2 at two different candidates: n2 == b1, decided in f64 later; either position will do,
    // but every lane must pick the same one
    const bool take = o1 < b1 || (o1 == b1 && op < pos);
    pos = take ? op : pos;
    b1 = fminf(b1, o1);
    b2 = n2;
}
template <int G>
__device__ __forceinline__ void group_merge(float &b1, float &b2, unsigned &pos)
{
    if (G >= 8) merge_step<kDppHalfMirror>(b1, b2, pos);
    if (G >= 4) merge_step<kDppXor2>(b1, b2, pos);
    if (G >= 2) merge_step<kDppXor1>(b1, b2, pos);
}

}  // namespace

// LDS carve (dynamic, every offset a multiple of 16):
//   tx,ty,tz float[CAP] x3   the streamed footprint, structure of arrays (12 B per point)
//   rowlo  u32[MAXR]         first target slot of a footprint row
//   rowoff u32[MAXR]         (first: one-past-last target slot) LDS offset of the row
//   clist  u32[MAXR]         footprint rows that hold points, compacted
//   hkey   u32[MAXR]         row table keys ((y << 16) | z, open addressing)
//   rbeg   u32[9][QPB]       per query, visiting order: first candidate of the run (LDS offset in tile
//                            mode, target slot otherwise)
//   rlen   u32[9][QPB]       ... its length
//   rgl    u32[9][QPB]       ... its first target slot
//   scr    int[128]          bounding box / scan / flags scratch
// G consecutive lanes work on one query (QPB = NTH / G queries per workgroup): a tile serves
// NT/NS target points per query whatever the geometry, so one query per lane would leave room
// for only one or two workgroups per CU; G lanes per query shrink the tile G-fold and split
// the bound loads and the candidates of a query between them.
template <bool PLANE, int NTH, int G, int CAP, int MAXR, bool PRUNE, bool STAMPS>
__global__ __launch_bounds__(NTH, 3) void nn_tile_reduce_kernel(const TileArgs a)
{
    constexpr int NACC = Acc<PLANE>::N;
    constexpr int NW = NTH / 64;
    constexpr int QPB = NTH / G;
    constexpr int U = G >= 8 ? 2 : 4;                      // candidates per lane per trip of the search loop
    constexpr int RPL = (9 + G - 1) / G;                   // rows per lane
    static_assert(MAXR % NTH == 0 && (MAXR & (MAXR - 1)) == 0, "MAXR: a power of two, a multiple of the workgroup size");
    constexpr int kLogR = MAXR == 256 ? 8 : (MAXR == 512 ? 9 : (MAXR == 1024 ? 10 : (MAXR == 2048 ? 11 : 7)));
    static_assert((1 << kLogR) == MAXR, "MAXR out of range");
    static_assert(CAP <= 65535 && CAP % 4 == 0, "CAP out of range");
    static_assert((size_t)CAP * 12 >= (size_t)QPB * NACC * 8, "moment scratch does not fit the tile region");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *tx = reinterpret_cast<float *>(smem);
    float *ty = tx + CAP;
    float *tz = ty + CAP;
    unsigned *rowlo = reinterpret_cast<unsigned *>(tz + CAP);
    unsigned *rowoff = rowlo + MAXR;
    unsigned *clist = rowoff + MAXR;
    unsigned *hkey = clist + MAXR;
    unsigned *rbeg = hkey + MAXR;
    unsigned *rlen = rbeg + 9 * QPB;
    unsigned *rgl = rlen + 9 * QPB;
    int *scr = reinterpret_cast<int *>(rgl + 9 * QPB);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = tid / G, sub = tid % G;
    long long tstamp[10];
    int nstamp = 0;
#define VISMA_STAMP() do { if (STAMPS && a.stats && nstamp < 10) tstamp[nstamp] = clock64(); ++nstamp; } while (0)
#define VISMA_STAMP_FIRST() do { if (passes_run == 0) VISMA_STAMP(); } while (0)
    VISMA_STAMP();                                            // 0: start

    // ---- which problem, which chunk of its queries -------------------------------------
    int prob, lb, bpp = a.bpp, ns = a.ns;
    const Pt64 *src64 = a.src64;
    const float4 *sorted = a.sorted;
    const Pt64 *sorted64 = a.sorted64;
    const unsigned *start = a.start;
    GridParams g = a.g;
    int *idx_out = a.idx_out;
    float *d2_out = a.d2_out;
    long long row0;                                        // first partial row of this problem
    if (a.descs) {
        int lo = 0, hi = a.nprob - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (a.descs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        prob = lo;
        const ProbDesc d = a.descs[prob];
        lb = (int)blockIdx.x - d.first_block;
        bpp = d.nblocks;
        ns = d.ns;
        src64 += d.src_off;
        sorted += d.sorted_off;
        sorted64 += d.sorted_off;
        start += d.start_off;
        g = d.g;
        idx_out += d.out_off;
        d2_out += d.out_off;
        row0 = d.first_block;
    } else {
        prob = blockIdx.x / bpp;
        lb = blockIdx.x - prob * bpp;
        idx_out += (long long)prob * a.out_stride;
        d2_out += (long long)prob * a.out_stride;
        row0 = (long long)prob * bpp;
    }
    Xform64 T64 = a.T64;
    Offset64 off = a.off;
    float r2f = a.r2f;
    {
        Xform32 T32unused;
        if (!load_loop_state(a.st ? a.st + prob : nullptr, T32unused, T64, off, r2f)) return;
    }
    const double r2d = (double)r2f;                        // (double)(float)(r*r): KDTreeFlann.cpp:184-185

    // XCD-aware chunking: workgroup b runs on XCD b % 8; give each XCD one contiguous share
    // of the Morton-ordered queries so that its private L2 holds one region of the target.
    int vb;
    {
        const int x = lb & 7, qq = bpp >> 3, r = bpp & 7;
        vb = x * qq + min(x, r) + (lb >> 3);
    }
    const int i = vb * QPB + q;
    const bool valid = i < ns;

    // ---- (A) transform, cell -------------------------------------------------------------
    Pt64 s8;
    s8.x = s8.y = s8.z = 0.0;
    s8.w = 0ull;
    if (valid) s8 = src64[i];
    const double pxd = T64.m[0] * s8.x + T64.m[1] * s8.y + T64.m[2] * s8.z + T64.m[3] * 1.0;
    const double pyd = T64.m[4] * s8.x + T64.m[5] * s8.y + T64.m[6] * s8.z + T64.m[7] * 1.0;
    const double pzd = T64.m[8] * s8.x + T64.m[9] * s8.y + T64.m[10] * s8.z + T64.m[11] * 1.0;
    const float px = (float)pxd, py = (float)pyd, pz = (float)pzd;
    const int cx = cell_coord(px, g.mn[0], g.inv_h, g.dim[0]);
    const int cy = cell_coord(py, g.mn[1], g.inv_hs, g.dim[1]);
    const int cz = cell_coord(pz, g.mn[2], g.inv_hs, g.dim[2]);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dim[0] - 1);
    const int ylo = max(cy - 1, 0), yhi = min(cy + 1, g.dim[1] - 1);
    const int zlo = max(cz - 1, 0), zhi = min(cz + 1, g.dim[2] - 1);
    (void)ylo; (void)yhi; (void)zlo; (void)zhi;
    VISMA_STAMP();                                            // 1: source loaded, transformed

    // ---- (B) run bounds: lane `sub` of a query fetches the rows it visits kk = sub, sub+G, ... ----
    unsigned rb[RPL], rl[RPL];
    int ry[RPL], rz[RPL];
#pragma unroll
    for (int m = 0; m < RPL; m++) {
        const int kk = sub + m * G;
        const int k = row_of_visit(kk < 9 ? kk : 0);
        const int dy = k % 3 - 1, dz = k / 3 - 1;
        const int z = cz + dz, y = cy + dy;
        const bool ok = kk < 9 && valid && (x0 <= x1) && z >= 0 && z < g.dim[2] && y >= 0 && y < g.dim[1];
        const int row = ((ok ? z : 0) * g.dim[1] + (ok ? y : 0)) * g.dim[0];
        // ONE 16-byte load (4-byte aligned): the starts of cells x0 .. x0+3 (the table has slack at its end)
        typedef unsigned u4a __attribute__((ext_vector_type(4), aligned(4)));
        u4a v = {0u, 0u, 0u, 0u};
        if (ok) v = *reinterpret_cast<const u4a *>(start + row + x0);
        const int span = x1 + 1 - x0;
        const unsigned b = v.x, e = ok ? (span >= 3 ? v.w : (span == 2 ? v.z : v.y)) : 0u;
        rb[m] = b;
        rl[m] = e > b ? e - b : 0u;
        ry[m] = y;
        rz[m] = z;
    }

    // Rounding band.  p32 = fl(p64), q32 = fl(q64): the difference vector is off by at most
    // u (|p|+|q|) per component, u = 2^-24, and the fp32 evaluation of d2 adds 3u relative, so
    // |d64 - sqrt(d2_32)| <= u (2 |p| + 2.5 r).  E is more than twice that.
    const float r_f = sqrtf(r2f);
    const float r_up = r_f * (1.0f + 2.4e-7f), r_dn = r_f * (1.0f - 2.4e-7f);
    const float E = 2.4e-7f * (fabsf(px) + fabsf(py) + fabsf(pz) + r_up) + 4.8e-7f * r_up;
    float lim;                                               // candidates at or beyond it never matter
    {
        const float t = r_up + 2.0f * E;
        lim = t * t * (1.0f + 6e-7f);
    }
    // A row (dy,dz) is SKIPPED when its slab cannot hold a candidate that matters: every point
    // of it is at least |(dist to the slab in y, in z)| away.  Margins: 1e-3 cell (fp32 binning
    // of query and candidates, kGridMaxDim) plus the rounding band, 1e-5 relative on the square.
    const float fy = (py - g.mn[1]) * g.inv_hs - (float)cy;
    const float fz = (pz - g.mn[2]) * g.inv_hs - (float)cz;
    const float mgn = 1e-3f + 4.0f * E * g.inv_hs;
    const float lo_y = fmaxf(fy - mgn, 0.f), hi_y = fmaxf(1.0f - fy - mgn, 0.f);
    const float lo_z = fmaxf(fz - mgn, 0.f), hi_z = fmaxf(1.0f - fz - mgn, 0.f);
    const float h2 = g.hs * g.hs * (1.0f - 1e-5f);

    // results of the query (valid on every lane of its group after the search)
    float b1 = lim, b2 = lim;
    unsigned gpos1 = kNone;                                  // target slot of the fp32 winner
    Pt64 q8;                                                 // the winner, f64
    q8.x = q8.y = q8.z = 0.0;
    q8.w = 0ull;
    double bd = r2d;
    bool hit = false, amb = false;
    unsigned ncand = 0, ncand_all = 0;
    unsigned total_pts = 0, total_rows = 0;
    int passes_run = 0, global_passes = 0;

    // The chunk is searched in `npass` parts (1 unless its footprint does not fit the tile):
    // part p = queries [p, p+1) * QPB / npass.
    int npass = a.force_fallback ? 0 : 1;
    bool global_mode = a.force_fallback != 0;
    for (int pass = 0; pass < (npass ? npass : 1);) {
        const int qlo = npass ? pass * (QPB / npass) : 0, qhi = npass ? qlo + QPB / npass : QPB;
        const bool active = q >= qlo && q < qhi;
        bool tile = !global_mode;
        unsigned total = 0, nrows = 0;
        int slot[RPL];                                       // where the row of visit slot kk lives in the row table
#pragma unroll
        for (int m = 0; m < RPL; m++) slot[m] = 0;
        if (tile) {
            // ---- footprint rows of this part: a hashed table keyed by (y, z) -- only rows that hold
            //      candidates get an entry, whatever the shape of the chunk ----------------------
#pragma unroll
            for (int s = 0; s < MAXR / NTH; s++) {
                hkey[tid + s * NTH] = kNone;
                rowlo[tid + s * NTH] = kNone;
                rowoff[tid + s * NTH] = 0u;                  // "rowhi" until the scan
            }
            if (tid == 0) scr[40] = 0;
            __syncthreads();
            if (active) {
#pragma unroll
                for (int m = 0; m < RPL; m++)
                    if (rl[m]) {
                        const unsigned key = ((unsigned)ry[m] << 16) | (unsigned)rz[m];
                        unsigned h = (key * 2654435761u) >> (32 - kLogR);
                        int probe = 0;
                        for (; probe < MAXR; probe++) {
                            const unsigned old = atomicCAS(&hkey[h], kNone, key);
                            if (old == kNone || old == key) break;
                            h = (h + 1) & (MAXR - 1);
                        }
                        if (probe == MAXR) { scr[40] = 1; continue; }     // table full: this part is split
                        slot[m] = (int)h;
                        atomicMin(&rowlo[h], rb[m]);
                        atomicMax(&rowoff[h], rb[m] + rl[m]);
                    }
            }
            __syncthreads();
            if (scr[40]) tile = false;
            const int nslots = MAXR;
            // ---- exclusive scan of (row length, row holds points) -> LDS offsets, compact list --
            if (tile) {
                constexpr int SPT = MAXR / NTH;
                unsigned len[SPT], sum = 0;                  // sum = points | rows << 16
#pragma unroll
                for (int s = 0; s < SPT; s++) {
                    const int sl = tid * SPT + s;
                    const unsigned l = rowlo[sl], h = rowoff[sl];
                    len[s] = (sl < nslots && h > l) ? h - l : 0u;
                    if (len[s] > 0xFFFFu) len[s] = 0xFFFFu;  // (too large anyway; keeps the packed sum valid)
                    sum += len[s] ? (len[s] | 0x10000u) : 0u;
                }
                // points of a part can exceed 16 bits only when it does not fit: detect by a second, clamped sum
                unsigned inc = sum;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned t = __shfl_up(inc, o, 64);
                    if (lane >= o) inc += t;
                }
                unsigned long long wide = sum & 0xFFFFu;     // exact point count of the workgroup
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) wide += __shfl_xor(wide, o, 64);
                if (lane == 63) scr[16 + wave] = (int)inc;
                if (lane == 0) scr[24 + wave] = (int)(wide > 0x7FFFFFFFull ? 0x7FFFFFFF : wide);
                __syncthreads();
                unsigned wbase = 0, packed_total = 0;
                unsigned long long wtot = 0;
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    const unsigned t = (unsigned)scr[16 + w];
                    if (w < wave) wbase += t;
                    packed_total += t;
                    wtot += (unsigned)scr[24 + w];
                }
                if (wtot > (unsigned long long)CAP) {
                    tile = false;                            // block-uniform
                } else {
                    total = packed_total & 0xFFFFu;
                    nrows = packed_total >> 16;
                    unsigned run = wbase + inc - sum;
#pragma unroll
                    for (int s = 0; s < SPT; s++) {
                        const int sl = tid * SPT + s;
                        rowoff[sl] = run & 0xFFFFu;
                        if (len[s]) clist[run >> 16] = (unsigned)sl;
                        run += len[s] ? (len[s] | 0x10000u) : 0u;
                    }
                }
                __syncthreads();
            }
        }
        if (!tile && !global_mode) {
            // does not fit: halve the part; a single query that does not fit is searched from global memory
            if (QPB / npass > 1 && npass < 8) {
                npass *= 2;
                pass *= 2;
                continue;
            }
        }
        VISMA_STAMP_FIRST();                                  // 2 (first part): rows, unions, scan

        // ---- (C) stream the footprint into LDS: 16 lanes per row, four rows x 48 points in flight ----
        if (tile && total > 0) {
            constexpr int NG16 = NTH / 16;
            const unsigned l16 = tid & 15, g16 = tid >> 4;
            for (unsigned c0 = g16; c0 < nrows; c0 += 4 * NG16) {
                unsigned lo[4], of[4], ln[4];
                float4 v[4][3];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const unsigned c = c0 + u * NG16;
                    const unsigned sl = clist[c < nrows ? c : 0];
                    lo[u] = rowlo[sl];
                    of[u] = rowoff[sl];
                    ln[u] = 0;                               // (empty rows share offsets: length from the next row that holds points)
                    if (c < nrows) ln[u] = ((c + 1 < nrows) ? rowoff[clist[c + 1]] : total) - of[u];
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int h = 0; h < 3; h++)
                        if (l16 + 16 * h < ln[u]) v[u][h] = sorted[lo[u] + l16 + 16 * h];
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int h = 0; h < 3; h++)
                        if (l16 + 16 * h < ln[u]) {
                            const unsigned o = of[u] + l16 + 16 * h;
                            tx[o] = v[u][h].x; ty[o] = v[u][h].y; tz[o] = v[u][h].z;
                        }
#pragma unroll 1
                for (int u = 0; u < 4; u++)
                    for (unsigned j = l16 + 48; j < ln[u]; j += 16) {
                        const float4 w4 = sorted[lo[u] + j];
                        tx[of[u] + j] = w4.x; ty[of[u] + j] = w4.y; tz[of[u] + j] = w4.z;
                    }
            }
        }

        // ---- row lists of the active queries (visiting order) ------------------------------
        if (active) {
#pragma unroll
            for (int m = 0; m < RPL; m++) {
                const int kk = sub + m * G;
                if (kk < 9) {
                    ncand_all += rl[m];
                    unsigned o = rb[m];
                    if (tile) o = rl[m] ? rowoff[slot[m]] + (rb[m] - rowlo[slot[m]]) : 0u;
                    rbeg[kk * QPB + q] = o;
                    rlen[kk * QPB + q] = rl[m];
                    rgl[kk * QPB + q] = rb[m];
                }
            }
        }
        __syncthreads();                                     // tile + lists complete
        VISMA_STAMP_FIRST();                                  // 3 (first part): tile streamed

        // ---- (D) fp32 search of the active queries: best and runner-up --------------------------
        if (active) {
            float c1 = lim, c2 = lim;
            unsigned cpos = kNone;
            auto search = [&](auto tile_tag) {
                constexpr bool TILE = decltype(tile_tag)::value;
                // one run: every lane of the query takes candidates sub, sub + G, ...
                auto run_row = [&](const unsigned s, const unsigned l, const unsigned gs) {
                    if (sub == 0) ncand += l;
                    for (unsigned j0 = 0; j0 < l; j0 += G * U) {
                        float qx[U], qy[U], qz[U];
                        unsigned jc[U];
#pragma unroll
                        for (int u = 0; u < U; u++) {
                            const unsigned j = j0 + sub + u * G;
                            jc[u] = j < l ? j : l - 1;
                            if constexpr (TILE) {
                                qx[u] = tx[s + jc[u]]; qy[u] = ty[s + jc[u]]; qz[u] = tz[s + jc[u]];
                            } else {
                                const float4 w4 = sorted[s + jc[u]];
                                qx[u] = w4.x; qy[u] = w4.y; qz[u] = w4.z;
                            }
                        }
#pragma unroll
                        for (int u = 0; u < U; u++) {
                            const float ddx = qx[u] - px, ddy = qy[u] - py, ddz = qz[u] - pz;
                            float d = __builtin_fmaf(ddz, ddz, __builtin_fmaf(ddy, ddy, ddx * ddx));
                            d = (j0 + sub + u * G < l) ? d : INFINITY;   // a padding slot is not a second candidate
                            const bool lt = d < c1;
                            c2 = med3_f32(c1, c2, d);
                            cpos = lt ? gs + jc[u] : cpos;
                            c1 = lt ? d : c1;
                        }
                    }
                };
                // the centre row is never pruned
                run_row(rbeg[q], rlen[q], rgl[q]);
                // which of the other eight rows can still hold a candidate that matters: decided for all
                // of them at once (straight-line code, the list reads overlap), then only those are walked
                float gbest = PRUNE ? group_min<G>(c1) : INFINITY;
                unsigned live = 0;
#pragma unroll
                for (int kk = 1; kk < 9; kk++) {
                    constexpr int order[9] = {4, 1, 3, 5, 7, 0, 2, 6, 8};
                    const int k = order[kk];
                    const int dy = k % 3 - 1, dz = k / 3 - 1;
                    const float ey = dy == 0 ? 0.f : (dy < 0 ? lo_y : hi_y);
                    const float ez = dz == 0 ? 0.f : (dz < 0 ? lo_z : hi_z);
                    const float bound = (ey * ey + ez * ez) * h2;
                    if (rlen[kk * QPB + q] != 0u && !(bound > gbest)) live |= 1u << kk;
                }
                while (live) {                               // uniform over the G lanes of a query
                    const int kk = __ffs((int)live) - 1;
                    live &= live - 1u;
                    if (PRUNE) {                             // the best may have improved since the mask was formed
                        const int k = row_of_visit(kk);
                        const int dy = k % 3 - 1, dz = k / 3 - 1;
                        const float ey = dy == 0 ? 0.f : (dy < 0 ? lo_y : hi_y);
                        const float ez = dz == 0 ? 0.f : (dz < 0 ? lo_z : hi_z);
                        if ((ey * ey + ez * ez) * h2 > group_min<G>(c1)) continue;
                    }
                    run_row(rbeg[kk * QPB + q], rlen[kk * QPB + q], rgl[kk * QPB + q]);
                }
            };
            if (tile) search(std::true_type{}); else search(std::false_type{});
            group_merge<G>(c1, c2, cpos);
            b1 = c1; b2 = c2; gpos1 = cpos;

            // ---- (E) decisive?  otherwise re-rank in f64 (lane 0 of the query) -------------------
            if (sub == 0 && gpos1 != kNone) {
                const float s1 = sqrtf(b1), s2 = sqrtf(b2);
                amb = (s1 + 2.0f * E >= s2) || (s1 + E >= r_dn);
            }
            if (amb) {
                // exact: every candidate inside the band, ranked by the reference's f64 sum of squares,
                // lowest original index on exact ties, accepted iff d2 < (double)(float)(r*r)
                const float sl = fminf(sqrtf(b1), r_up) + 2.0f * E;
                const float L = sl * sl * (1.0f + 6e-7f);
                unsigned bidx = kNone;
#pragma unroll 1
                for (int kk = 0; kk < 9; kk++) {
                    const unsigned s = rbeg[kk * QPB + q], l = rlen[kk * QPB + q], gs = rgl[kk * QPB + q];
#pragma unroll 1
                    for (unsigned j = 0; j < l; j++) {
                        float cx_, cy_, cz_;
                        if (tile) { cx_ = tx[s + j]; cy_ = ty[s + j]; cz_ = tz[s + j]; }
                        else { const float4 w4 = sorted[s + j]; cx_ = w4.x; cy_ = w4.y; cz_ = w4.z; }
                        const float ddx = cx_ - px, ddy = cy_ - py, ddz = cz_ - pz;
                        if (!(__builtin_fmaf(ddz, ddz, __builtin_fmaf(ddy, ddy, ddx * ddx)) <= L)) continue;
                        const Pt64 c8 = sorted64[gs + j];
                        // flann L2 (dist.h:159-176): result += diff * diff over x, y, z
                        const double dx = c8.x - pxd, dy = c8.y - pyd, dz = c8.z - pzd;
                        double d = dx * dx;
                        d += dy * dy;
                        d += dz * dz;
                        const unsigned id = (unsigned)c8.w;
                        const bool lt = d < bd || (d == bd && id < bidx && bidx != kNone);
                        if (lt) { bd = d; bidx = id; q8 = c8; }
                    }
                }
                hit = bidx != kNone;
            }
        }
        if (passes_run == 0) { total_pts = total; total_rows = nrows; }
        ++passes_run;
        if (!tile) ++global_passes;
        ++pass;
        if (pass < (npass ? npass : 1)) __syncthreads();     // the next part reuses the tile and the lists
    }
    nstamp = 4;
    VISMA_STAMP();                                            // 4: search (+ re-rank) done

    // the decisive queries fetch their winner in f64
    if (sub == 0 && !amb && gpos1 != kNone) {
        q8 = sorted64[gpos1];
        const double dx = q8.x - pxd, dy = q8.y - pyd, dz = q8.z - pzd;
        double d = dx * dx;
        d += dy * dy;
        d += dz * dz;
        bd = d;
        hit = true;
    }
    if (valid && sub == 0) {
        idx_out[i] = hit ? (int)(unsigned)q8.w : -1;
        d2_out[i] = (float)bd;
    }
    VISMA_STAMP();                                            // 5: winner fetched

    // ---- (F) moments of the winner; workgroup sum through LDS in a fixed order -------------
    __syncthreads();                                         // everyone is done with the tile and the lists
    double *mom = reinterpret_cast<double *>(smem);          // [QPB][NACC]   (tile region)
    double *part = reinterpret_cast<double *>(rowlo);        // [NTH/32][33]  (rows / lists region)
    double *tot = part + (NTH / 32) * 33;                    // [32]
    static_assert((size_t)MAXR * 16 + (size_t)27 * QPB * 4 >= (size_t)((NTH / 32) * 33 + 32) * 8, "fold scratch does not fit");
    if (sub == 0) {
        double acc[NACC];
#pragma unroll
        for (int k = 0; k < NACC; k++) acc[k] = 0.0;
        if (hit) {
            double nx = 0.0, ny = 0.0, nz = 0.0;
            if (PLANE) {
                if (a.nrm64) { const Pt64 n8 = a.nrm64[(unsigned)q8.w]; nx = n8.x; ny = n8.y; nz = n8.z; }
                else { const float4 n4 = a.nrm[(unsigned)q8.w]; nx = n4.x; ny = n4.y; nz = n4.z; }
            }
            // p = T64 * s is at hand (pxd, pyd, pzd): the identity transform reproduces accumulate_pair_d's p + off
        Xform64 I64;
#pragma unroll
        for (int k = 0; k < 12; k++) I64.m[k] = (k % 5 == 0) ? 1.0 : 0.0;
        accumulate_pair_d<PLANE>(acc, pxd, pyd, pzd, q8.x, q8.y, q8.z, nx, ny, nz, I64, off);
        }
#pragma unroll
        for (int k = 0; k < NACC; k++) mom[q * NACC + k] = acc[k];
    }
    __syncthreads();
    const int sa = tid & 31, sg = tid >> 5;
    constexpr int NG = NTH / 32;
    {
        double v = 0.0;
        if (sa < NACC)
            for (int r = sg; r < QPB; r += NG) v += mom[r * NACC + sa];
        part[sg * 33 + sa] = v;
    }
    __syncthreads();
    double *rows = a.partials + (row0 + lb) * kReduceAcc;
    if (tid < NACC) {
        double v = 0.0;
#pragma unroll
        for (int gg = 0; gg < NG; gg++) v += part[gg * 33 + tid];
        if (a.tickets) store_agent_f64(rows + tid, v);
        else rows[tid] = v;
    }
    VISMA_STAMP();                                            // 6: row stored
    if (a.stats && tid == 0) {
        // profiling: tiles / global-memory parts / streamed points / f64 re-ranks / footprint rows / histogram
        unsigned long long *t = a.stats + 24 * (blockIdx.x & 511);
        atomicAdd(t + 0, 1ull);
        atomicAdd(t + 1, (unsigned long long)global_passes);
        atomicAdd(t + 2, (unsigned long long)total_pts);
        atomicAdd(t + 4, (unsigned long long)total_rows);
        atomicAdd(t + 7, (unsigned long long)passes_run);
        if (STAMPS) {
#pragma unroll
            for (int k = 0; k < 6; k++) atomicAdd(t + 8 + k, (unsigned long long)(tstamp[k + 1] - tstamp[k]));
        }
    }
    if (a.stats) {
        unsigned long long c = ncand, ca = ncand_all, am = amb ? 1ull : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            c += __shfl_down(c, o, 64);
            ca += __shfl_down(ca, o, 64);
            am += __shfl_down(am, o, 64);
        }
        if (lane == 0) {
            unsigned long long *t = a.stats + 24 * (blockIdx.x & 511);
            atomicAdd(t + 3, am);
            atomicAdd(t + 5, c);
            atomicAdd(t + 6, ca);
        }
    }
#undef VISMA_STAMP
#undef VISMA_STAMP_FIRST
    if (!a.tickets) return;

    // ---- fused fold: the last arriver of each group of rows folds it, the last group folder
    //      folds the group rows; every sum in a fixed order -------------------------------
    const int ngroups = (bpp + kGroupRows - 1) / kGroupRows;
    unsigned *tk = a.tickets + (long long)prob * a.ticket_stride;   // [0]: level 2, [1 + g]: level 1
    const int grp = lb / kGroupRows;
    const int gsize = min(kGroupRows, bpp - grp * kGroupRows);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every storing wave drains (write-through stores)
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(tk + 1 + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        scr[32] = (t == (unsigned)(gsize - 1)) ? 1 : 0;
        if (scr[32]) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(tk + 1 + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
    }
    __syncthreads();
    if (!scr[32]) return;
    {
        const double *grows = a.partials + (row0 + (long long)grp * kGroupRows) * kReduceAcc;
        double v = 0.0;
        if (sa < NACC)
            for (int r = sg; r < gsize; r += NG) v += load_agent_f64(grows + (long long)r * kReduceAcc + sa);
        part[sg * 33 + sa] = v;
    }
    __syncthreads();
    double *rows2 = a.partials2 + ((long long)prob * a.ticket_stride + grp) * kReduceAcc;
    if (tid < 32) {
        double t = 0.0;
#pragma unroll
        for (int gg = 0; gg < NG; gg++) t += part[gg * 33 + tid];
        if (tid < NACC) store_agent_f64(rows2 + tid, t);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        scr[33] = (t == (unsigned)(ngroups - 1)) ? 1 : 0;
        if (scr[33]) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!scr[33]) return;
    {
        const double *g2 = a.partials2 + (long long)prob * a.ticket_stride * kReduceAcc;
        double v = 0.0;
        if (sa < NACC)
            for (int r = sg; r < ngroups; r += NG) v += load_agent_f64(g2 + (long long)r * kReduceAcc + sa);
        part[sg * 33 + sa] = v;
    }
    __syncthreads();
    if (tid < 32) {
        double t = 0.0;
#pragma unroll
        for (int gg = 0; gg < NG; gg++) t += part[gg * 33 + tid];
        tot[tid] = t;
    }
    __syncthreads();
    double *stats = a.stats_out + (long long)prob * a.stats_stride;
    if (tid == 0) expand_moments<PLANE>(tot, stats);
    if (a.host_out) publish_tagged_stats(stats, a.host_out, a.seq);
}

// ---- launch ------------------------------------------------------------------------------
template <bool PLANE, int NTH, int G, int CAP, int MAXR, bool PRUNE, bool STAMPS = false>
static hipError_t launch_tile_t(const TileArgs &a, int total_blocks, hipStream_t stream)
{
    constexpr size_t lds = (size_t)CAP * 12 + (size_t)MAXR * 16 + (size_t)27 * (NTH / G) * 4 + 128 * 4;
    static bool once = false;
    if (!once) {
        hipError_t e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&nn_tile_reduce_kernel<PLANE, NTH, G, CAP, MAXR, PRUNE, STAMPS>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        once = true;
    }
    hipLaunchKernelGGL((nn_tile_reduce_kernel<PLANE, NTH, G, CAP, MAXR, PRUNE, STAMPS>), dim3(total_blocks), dim3(NTH), lds,
                       stream, a);
    return hipGetLastError();
}

// queries per workgroup of a tile configuration
int tile_threads(int config)
{
    switch (config) {
    case 1: return 64;     // 256 threads, 4 lanes per query
    case 2: return 256;    // 256 threads, 1 lane per query
    case 3: return 64;     // 128 threads, 2 lanes per query
    case 4: return 128;    // 128 threads, 1 lane per query
    case 10: return 128;   // config 0 with the per-phase cycle stamps (profiling builds of the probe)
    default: return 128;   // 256 threads, 2 lanes per query
    }
}

#define VISMA_TILE_CASES(PL)                                                                       \
    switch (config) {                                                                              \
    case 1: return launch_tile_t<PL, 256, 4, 2176, 256, true>(a, total_blocks, stream);            \
    case 2: return launch_tile_t<PL, 256, 1, 6144, 1024, true>(a, total_blocks, stream);           \
    case 3: return launch_tile_t<PL, 128, 2, 2176, 256, true>(a, total_blocks, stream);            \
    case 4: return launch_tile_t<PL, 128, 1, 3328, 512, true>(a, total_blocks, stream);            \
    case 10: return launch_tile_t<PL, 256, 2, 3328, 512, true, true>(a, total_blocks, stream);     \
    default: return launch_tile_t<PL, 256, 2, 3328, 512, true>(a, total_blocks, stream);           \
    }

hipError_t launch_nn_tile_reduce(const TileArgs &a, int point_to_plane, int config, int total_blocks,
                                 hipStream_t stream)
{
    if (point_to_plane) { VISMA_TILE_CASES(true) }
    VISMA_TILE_CASES(false)
}
#undef VISMA_TILE_CASES

}  // namespace visma
