// grid.hip -- radius-cell uniform grid: exact radius-limited nearest neighbour
// in O(NS * 27 cells) instead of O(NS * NT)  (SURVEY.md 8f row 1).
//
// Same answers as the brute-force kernel, bit for bit: the candidate set of a
// query is every target point in the 3x3x3 block of cells around it, the cell
// edge h = 1.001 * max_dist (>= max_dist with room for fp32 rounding of the
// cell coordinate), so every point with fp32 d2 < (float)(max_dist^2) is a
// candidate; candidates are ranked by (d2, original index), i.e. lowest index
// on exact ties, with the same fp32 arithmetic (device_common.h).
//
// Replaces the reference's KD-tree (KDTreeFlann::SetRawData / SearchHybrid,
// O3D/Core/Geometry/KDTreeFlann.cpp:164-208): build = bounding box ->
// per-point cell id + histogram -> exclusive scan -> scatter (counting sort,
// target points stored contiguously per cell with their original index in .w);
// query = fused transform + 9 contiguous runs (x-adjacent cells are adjacent in
// memory) + Jacobian/residual accumulation + wave-shuffle reduction, ONE kernel
// per ICP iteration.
#include "device_common.h"

#include <stdlib.h>
#include <algorithm>

#include <math.h>
#include <string.h>

namespace visma {

// ---- order-preserving float <-> uint map (for atomic min/max) --------------
__device__ __forceinline__ unsigned f2ord(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void bbox_init_kernel(unsigned *box)
{
    if (threadIdx.x < 3) box[threadIdx.x] = 0xFFFFFFFFu;      // min x,y,z
    else if (threadIdx.x < 6) box[threadIdx.x] = 0u;          // max x,y,z
}

__global__ __launch_bounds__(256) void bbox_kernel(const float4 *__restrict__ pts, int n,
                                                   unsigned *__restrict__ box)
{
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float4 q = pts[i];
        mn[0] = fminf(mn[0], q.x); mx[0] = fmaxf(mx[0], q.x);
        mn[1] = fminf(mn[1], q.y); mx[1] = fmaxf(mx[1], q.y);
        mn[2] = fminf(mn[2], q.z); mx[2] = fmaxf(mx[2], q.z);
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_down(mn[a], off, 64));
            mx[a] = fmaxf(mx[a], __shfl_down(mx[a], off, 64));
        }
    }
    // one set of six atomics per WORKGROUP (they all hit the same six words: 8192 waves issuing their own
    // took 0.57 ms for 4 M points, twenty times the read of the points)
    __shared__ float wmn[4][3], wmx[4][3];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) { wmn[w][a] = mn[a]; wmx[w][a] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        const float lo = fminf(fminf(wmn[0][a], wmn[1][a]), fminf(wmn[2][a], wmn[3][a]));
        const float hi = fmaxf(fmaxf(wmx[0][a], wmx[1][a]), fmaxf(wmx[2][a], wmx[3][a]));
        atomicMin(&box[a], f2ord(lo));
        atomicMax(&box[3 + a], f2ord(hi));
    }
}

// cell of every point and its RANK inside the cell (the value the counting atomic returns): the scatter pass then
// needs no second round of atomics and no cleared cursor table (4.2 M points, 52 M cells: 330 + 27 us of the build)
__global__ __launch_bounds__(256) void cell_count_kernel(const float4 *__restrict__ pts, int n,
                                                         GridParams g,
                                                         uint2 *__restrict__ cell_rank,
                                                         unsigned *__restrict__ count)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 q = pts[i];
    int cx = cell_coord(q.x, g.mn[0], g.inv_h, g.dim[0]);
    int cy = cell_coord(q.y, g.mn[1], g.inv_hs, g.dim[1]);
    int cz = cell_coord(q.z, g.mn[2], g.inv_hs, g.dim[2]);
    cx = min(max(cx, 0), g.dim[0] - 1);
    cy = min(max(cy, 0), g.dim[1] - 1);
    cz = min(max(cz, 0), g.dim[2] - 1);
    const unsigned c = (unsigned)((cz * g.dim[1] + cy) * g.dim[0] + cx);
    cell_rank[i] = make_uint2(c, atomicAdd(&count[c], 1u));
}

// ---- exclusive scan of `count` (n entries) into `start` (n+1 entries) --------
constexpr int kScanPerBlock = 2048;   // 256 threads x 8

__global__ __launch_bounds__(256) void scan_block_sum_kernel(const unsigned *__restrict__ count,
                                                             long long n,
                                                             unsigned *__restrict__ bsum)
{
    __shared__ unsigned w[4];
    const long long base = (long long)blockIdx.x * kScanPerBlock + threadIdx.x * 8;
    unsigned s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (base + k < n) s += count[base + k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

// one workgroup: in-place exclusive scan of bsum[0..nb)
__global__ __launch_bounds__(1024) void scan_top_kernel(unsigned *__restrict__ bsum, int nb)
{
    __shared__ unsigned part[1024];
    const int per = (nb + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(lo + per, nb);
    unsigned s = 0;
    for (int i = lo; i < hi; i++) s += bsum[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
        unsigned v = (threadIdx.x >= off) ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned run = part[threadIdx.x] - s;        // exclusive prefix of this thread's range
    for (int i = lo; i < hi; i++) {
        const unsigned v = bsum[i];
        bsum[i] = run;
        run += v;
    }
}

__global__ __launch_bounds__(256) void scan_apply_kernel(const unsigned *__restrict__ count,
                                                         long long n,
                                                         const unsigned *__restrict__ bsum,
                                                         unsigned *__restrict__ start)
{
    __shared__ unsigned tsum[256];
    const long long base = (long long)blockIdx.x * kScanPerBlock + threadIdx.x * 8;
    unsigned v[8], s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        v[k] = (base + k < n) ? count[base + k] : 0u;
        s += v[k];
    }
    tsum[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        unsigned t = (threadIdx.x >= off) ? tsum[threadIdx.x - off] : 0u;
        __syncthreads();
        tsum[threadIdx.x] += t;
        __syncthreads();
    }
    unsigned run = bsum[blockIdx.x] + tsum[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (base + k < n) start[base + k] = run;
        run += v[k];
    }
    // the last element of all: start[n] = total
    if (base <= n - 1 && n - 1 < base + 8) start[n] = run;
}

__global__ void scan_total_kernel(const unsigned *__restrict__ count, long long n, unsigned *__restrict__ start)
{
    start[n] = start[n - 1] + count[n - 1];
}

__global__ __launch_bounds__(256) void cell_scatter_kernel(const float4 *__restrict__ pts, int n,
                                                           const uint2 *__restrict__ cell_rank,
                                                           const unsigned *__restrict__ start,
                                                           float4 *__restrict__ sorted,
                                                           const Pt64 *__restrict__ pts64,
                                                           Pt64 *__restrict__ sorted64)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint2 cr = cell_rank[i];
    const unsigned pos = start[cr.x] + cr.y;
    const float4 q = pts[i];
    sorted[pos] = make_float4(q.x, q.y, q.z, __uint_as_float((unsigned)i));
    if (pts64) {                                   // the f64 copy goes to the same slot
        Pt64 q8 = pts64[i];
        q8.w = (unsigned long long)(unsigned)i;
        sorted64[pos] = q8;
    }
}

__global__ void pack12_kernel(const float4 *__restrict__ src, float *__restrict__ dst, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float4 p = src[i]; dst[3 * i] = p.x; dst[3 * i + 1] = p.y; dst[3 * i + 2] = p.z; }
}
hipError_t launch_pack12(const float4 *src, float *dst, int64_t n, hipStream_t stream)
{
    if (n > 0) hipLaunchKernelGGL(pack12_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, dst, (long long)n);
    return hipGetLastError();
}

hipError_t launch_grid_bbox(const float4 *tgt, int64_t nt, unsigned *box6, hipStream_t stream)
{
    hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, stream, box6);
    if (nt > 0) {
        int blocks = (int)((nt + 255) / 256);
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(bbox_kernel, dim3(blocks), dim3(256), 0, stream, tgt, (int)nt, box6);
    }
    return hipGetLastError();
}

void grid_decode_bbox(const unsigned box6[6], float mn[3], float mx[3])
{
    for (int a = 0; a < 6; a++) {
        const unsigned o = box6[a];
        const unsigned u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
        float f;
        memcpy(&f, &u, sizeof(f));
        (a < 3 ? mn[a] : mx[a - 3]) = f;
    }
}

GridParams grid_plan(const float mn[3], const float mx[3], double max_dist, int64_t max_cells, int max_sub)
{
    GridParams g;
    float h = (float)(max_dist * 1.001);
    // cells larger than the radius are as exact (the 27 cells still cover it): fewer, longer rows --
    // fewer dependent trips per query, more candidates, a smaller table.  VISMA_ICP_GRID_CELL = factor
    if (const char *e = getenv("VISMA_ICP_GRID_CELL")) {
        const double f = atof(e);
        if (f >= 1.0 && f <= 64.0) h = (float)(max_dist * 1.001 * f);
    }
    if (!(h > 0.f) || !isfinite(h)) h = 1.0f;
    double ext[3];
    for (int a = 0; a < 3; a++) {
        ext[a] = (double)mx[a] - (double)mn[a];
        if (!(ext[a] >= 0.0) || !isfinite(ext[a])) ext[a] = 0.0;
    }
    for (;;) {
        double n = 1.0;
        bool too_long = false;
        for (int a = 0; a < 3; a++) {
            double d = floor(ext[a] / h) + 1.0;
            if (d > 2.0e9) d = 2.0e9;
            // the cell coordinate is computed in fp32: its error grows with the coordinate,
            // and the 0.1 % slack between h and max_dist must stay above it
            if (d > (double)kGridMaxDim) too_long = true;
            g.dim[a] = (int)d;
            n *= d;
        }
        if (n <= (double)max_cells && !too_long) break;
        h *= 1.26f;
    }
    g.sub = 1;
    if (max_sub >= 2) {
        // rows at half pitch: thinner rows, far fewer candidates per query; 4x the table
        const double dy = floor(ext[1] / (0.5 * h)) + 1.0, dz = floor(ext[2] / (0.5 * h)) + 1.0;
        if ((double)g.dim[0] * dy * dz <= (double)kGridMaxCellsFine) {
            g.sub = 2;
            g.dim[1] = (int)dy;
            g.dim[2] = (int)dz;
        }
    }
    for (int a = 0; a < 3; a++) g.mn[a] = mn[a];
    g.h = h;
    g.inv_h = 1.0f / h;
    g.hs = g.sub == 2 ? 0.5f * h : h;
    g.inv_hs = 1.0f / g.hs;
    g.ncell = (int64_t)g.dim[0] * g.dim[1] * g.dim[2];
    g.ring = 0;
    return g;
}

// Cells of edge `cell` (smaller than the radius) for the ring search of grid_ring.hip: the table must fit like any other
// (the edge grows until it does); g.ring = the largest ring of rows a radius can reach, + 1.  A cell that ends up as large
// as the radius is the ordinary plan.
GridParams grid_plan_ring(const float mn[3], const float mx[3], double max_dist, double cell, int64_t max_cells)
{
    GridParams g = grid_plan(mn, mx, max_dist, max_cells);
    float h = (float)cell;
    if (!(h > 0.f) || !isfinite(h) || !(h < g.h)) return g;
    h = std::max(h, (float)(max_dist * 1.001 / (double)(kRingMaxRings - 2)));   // (no more rings than the table holds)
    double ext[3];
    for (int a = 0; a < 3; a++) {
        ext[a] = (double)mx[a] - (double)mn[a];
        if (!(ext[a] >= 0.0) || !isfinite(ext[a])) ext[a] = 0.0;
    }
    GridParams f = g;
    for (;;) {
        if (!(h < g.h)) return g;
        double n = 1.0;
        bool too_long = false;
        for (int a = 0; a < 3; a++) {
            const double d = floor(ext[a] / h) + 1.0;
            if (d > (double)kGridMaxDim) too_long = true;      // (fp32 binning error below 1e-3 cell)
            f.dim[a] = (int)std::min(d, 2.0e9);
            n *= d;
        }
        if (n <= (double)max_cells && !too_long) break;
        h *= 1.26f;
    }
    f.h = f.hs = h;
    f.inv_h = f.inv_hs = 1.0f / h;
    f.sub = 1;
    f.ncell = (int64_t)f.dim[0] * f.dim[1] * f.dim[2];
    const double rings = ceil(max_dist * 1.001 / (double)h) + 1.0;
    if (rings > (double)kRingMaxRings) return g;
    f.ring = (int)rings;                                      // (its visiting order: build_ring_table)
    return f;
}

// cell_of: 2 * nt words (cell, rank of the point in its cell); bsum: grid_scan_blocks(g.ncell) + 1 words of scratch.
// NOTE the order of points inside a cell is the order the counting atomics were served in: it varies from build to
// build, and nothing downstream depends on it (ties go to the lowest ORIGINAL index, kept in .w).
hipError_t launch_grid_build(const float4 *tgt, int64_t nt, const GridParams &g,
                             unsigned *cell_of, unsigned *count, unsigned *bsum,
                             unsigned *start, float4 *sorted, hipStream_t stream,
                             const Pt64 *tgt64, Pt64 *sorted64)
{
    hipError_t e = hipMemsetAsync(count, 0, sizeof(unsigned) * (size_t)g.ncell, stream);
    if (e != hipSuccess) return e;
    const int pblocks = (int)((nt + 255) / 256);
    uint2 *cell_rank = reinterpret_cast<uint2 *>(cell_of);
    if (nt > 0)
        hipLaunchKernelGGL(cell_count_kernel, dim3(pblocks), dim3(256), 0, stream, tgt, (int)nt, g,
                           cell_rank, count);
    // one pass over the table (decoupled look-back, hipCUB): 52 M cells at C4 -- the three-kernel scan read the table
    // twice and wrote it once, 229 us
    const size_t tmp_bytes = exclusive_scan_u32_tmp_bytes((long long)g.ncell);
    e = launch_exclusive_scan_u32_lib(count, (long long)g.ncell, start, bsum, tmp_bytes, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(scan_total_kernel, dim3(1), dim3(1), 0, stream, count, (long long)g.ncell, start);
    if (nt > 0)
        hipLaunchKernelGGL(cell_scatter_kernel, dim3(pblocks), dim3(256), 0, stream, tgt, (int)nt,
                           cell_rank, start, sorted, tgt64, sorted64);
    return hipGetLastError();
}

// generic exclusive scan of n u32 values (bsum: n/2048 + 2 entries of scratch); also
// writes out[n] = total, so `out` needs n + 1 entries
hipError_t launch_exclusive_scan_u32(const unsigned *in, long long n, unsigned *bsum, unsigned *out,
                                     hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    const int nb = (int)((n + kScanPerBlock - 1) / kScanPerBlock);
    hipLaunchKernelGGL(scan_block_sum_kernel, dim3(nb), dim3(256), 0, stream, in, n, bsum);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(1024), 0, stream, bsum, nb);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(256), 0, stream, in, n, bsum, out);
    return hipGetLastError();
}

// words of scan scratch a grid build over `ncell` cells needs (+ 1): the three-kernel scan's block sums or the
// library scan's temporary storage, whichever is larger
int grid_scan_blocks(int64_t ncell)
{
    const int64_t own = (ncell + kScanPerBlock - 1) / kScanPerBlock;
    const int64_t lib = (int64_t)((exclusive_scan_u32_tmp_bytes((long long)ncell) + 3) / 4) + 64;
    return (int)std::max(own, lib);
}

// ------------------------------------------------------------------------
// Query: fused transform + grid NN + Jacobian/residual + reduction
// ------------------------------------------------------------------------
// G consecutive lanes cooperate on one query: lane g takes candidates
// j = b+g, b+g+G, ... of each contiguous run (G*16-byte coalesced segments), the
// G partial minima are merged with G-lane butterfly shuffles on the
// (d2, original index) key, and lane 0 of the group accumulates the
// correspondence's Jacobian/residual rows.
// ONE: every lane group has at most G queries (one per lane), so each lane keeps
// ONE winner and builds its Jacobian/residual moments after the search -- the 29
// f64 accumulators are then not live during the search (58 VGPRs less, one more
// wave per SIMD) and no lane repeats another lane's accumulation.
//
// F64: the DOUBLE-PRECISION search (visma_icp_set_search_precision).  fp32 distances on fp32
// coordinates decide near-ties differently from the reference's f64 KD-tree about once in
// 10^5 queries; with K matched pairs one flipped pair moves the update by ~(pair spacing)/K,
// i.e. above the 1e-5 parity tolerance for clouds of a few thousand points (measured: 5 of 300
// random small registrations, tools/fuzz_icp_vs_oracle.py).  Here candidates are the caller's
// f64 points (32 B each), the source is transformed in f64, d2 is the reference's f64
// sum-of-squares (x, y, z order) and acceptance is d2 < (double)(float)(r*r)
// (KDTreeFlann.cpp:184-185): the correspondences ARE the reference's.  Cells and row pruning
// still use the fp32 view of the same points (margins cover the difference).
//
// HYB: the EXACT search at fp32 cost (the default, visma_icp_set_search_precision 1).  Candidates
// are ranked with fp32 arithmetic on the fp32 view of the target, but the three best squared
// distances are kept with their positions, plus the value of the fourth.  A query is decisive
// when its runner-up lies outside the rounding band of the best (2E, E bounding
// |d64 - sqrt(d2_32)|); otherwise the two or three candidates inside the band are fetched in
// f64 and ranked with the reference's arithmetic (flann dist.h:159-176), lowest original index
// on exact ties; acceptance is always decided in f64: d2 < (double)(float)(r*r)
// (KDTreeFlann.cpp:184-185).  Four candidates inside one band (about once in 1e9 queries;
// clouds with duplicated points) re-scan the neighbourhood in f64.  Source transform and
// statistics are the F64 ones, so the results equal the F64 kernel's bit for bit.
__device__ __forceinline__ float med3_f32(float a, float b, float c)
{
    float r;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// sorted insertion of (d, pos) into the three best of a query; h4 = value of the fourth
// The N best (d2, position) of the candidates seen so far and the value of the (N+1)-th, as
// straight-line VALU code: compares first, then the values (v_med3 / v_min), then the positions
// (v_cndmask).  Written as one asm block per insertion: the compiler turns the equivalent selects
// into nested exec-mask regions (it knows d < h1 implies d < h2), which costs more scalar
// instructions than the selects it avoids.  d is never NaN here.
#ifndef VISMA_GRID_NTOP
#define VISMA_GRID_NTOP 2
#endif
template <int N>
struct TopN {
    static_assert(N >= 1 && N <= 3, "");
    float h[4];                          // h[0..N-1] the best values, h[N] the next one
    unsigned p[3];
    __device__ __forceinline__ void init(float lim)
    {
#pragma unroll
        for (int k = 0; k < 4; k++) h[k] = lim;
#pragma unroll
        for (int k = 0; k < 3; k++) p[k] = 0xFFFFFFFFu;
    }
    __device__ __forceinline__ float next() const { return h[N]; }
    __device__ __forceinline__ void insert(float d, unsigned pos)
    {
        unsigned long long c1, c2, c3;
        if constexpr (N == 1) {
            asm("v_cmp_lt_f32_e64 %[c1], %[d], %[h0]\n\t"
                "v_med3_f32 %[h1], %[h0], %[h1], %[d]\n\t"
                "v_min_f32_e32 %[h0], %[h0], %[d]\n\t"
                "v_cndmask_b32_e64 %[p0], %[p0], %[pos], %[c1]"
                : [h0] "+v"(h[0]), [h1] "+v"(h[1]), [p0] "+v"(p[0]), [c1] "=&s"(c1)
                : [d] "v"(d), [pos] "v"(pos));
        } else if constexpr (N == 2) {
            asm("v_cmp_lt_f32_e64 %[c1], %[d], %[h0]\n\t"
                "v_cmp_lt_f32_e64 %[c2], %[d], %[h1]\n\t"
                "v_med3_f32 %[h2], %[h1], %[h2], %[d]\n\t"
                "v_med3_f32 %[h1], %[h0], %[h1], %[d]\n\t"
                "v_min_f32_e32 %[h0], %[h0], %[d]\n\t"
                "v_cndmask_b32_e64 %[p1], %[p1], %[pos], %[c2]\n\t"
                "v_cndmask_b32_e64 %[p1], %[p1], %[p0], %[c1]\n\t"
                "v_cndmask_b32_e64 %[p0], %[p0], %[pos], %[c1]"
                : [h0] "+v"(h[0]), [h1] "+v"(h[1]), [h2] "+v"(h[2]), [p0] "+v"(p[0]), [p1] "+v"(p[1]),
                  [c1] "=&s"(c1), [c2] "=&s"(c2)
                : [d] "v"(d), [pos] "v"(pos));
        } else {
            asm("v_cmp_lt_f32_e64 %[c1], %[d], %[h0]\n\t"
                "v_cmp_lt_f32_e64 %[c2], %[d], %[h1]\n\t"
                "v_cmp_lt_f32_e64 %[c3], %[d], %[h2]\n\t"
                "v_med3_f32 %[h3], %[h2], %[h3], %[d]\n\t"
                "v_med3_f32 %[h2], %[h1], %[h2], %[d]\n\t"
                "v_med3_f32 %[h1], %[h0], %[h1], %[d]\n\t"
                "v_min_f32_e32 %[h0], %[h0], %[d]\n\t"
                "v_cndmask_b32_e64 %[p2], %[p2], %[pos], %[c3]\n\t"
                "v_cndmask_b32_e64 %[p2], %[p2], %[p1], %[c2]\n\t"
                "v_cndmask_b32_e64 %[p1], %[p1], %[pos], %[c2]\n\t"
                "v_cndmask_b32_e64 %[p1], %[p1], %[p0], %[c1]\n\t"
                "v_cndmask_b32_e64 %[p0], %[p0], %[pos], %[c1]"
                : [h0] "+v"(h[0]), [h1] "+v"(h[1]), [h2] "+v"(h[2]), [h3] "+v"(h[3]), [p0] "+v"(p[0]),
                  [p1] "+v"(p[1]), [p2] "+v"(p[2]), [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3)
                : [d] "v"(d), [pos] "v"(pos));
        }
    }
};
typedef TopN<VISMA_GRID_NTOP> TopK;

struct P12 { float x, y, z; };            // a candidate of the exact search: fp32 rounding of the f64 coordinates

template <bool PLANE, int G, int U, bool ONE, bool F64 = false, bool HYB = false>
__global__ __launch_bounds__(kBlock) void nn_grid_reduce_kernel(
    const float4 *__restrict__ src, int ns, const float4 *__restrict__ sorted,
    const unsigned *__restrict__ start, GridParams g, const float4 *__restrict__ nrm, Xform32 T32,
    Xform64 T64, Offset64 off, float r2f, int *__restrict__ idx_out, float *__restrict__ d2_out,
    double *__restrict__ partials, unsigned long long *__restrict__ cand_count,
    const DevIcpState *__restrict__ st, int bpp, long long out_stride,
    const ProbDesc *__restrict__ descs, int nprob, const Pt64 *__restrict__ src64 = nullptr,
    const Pt64 *__restrict__ sorted64 = nullptr, double r2d = 0.0, const Pt64 *__restrict__ nrm64 = nullptr,
    const FoldArgs fold = FoldArgs{}, double *__restrict__ d64_out = nullptr,
    Pt64 *__restrict__ prevq_out = nullptr)
{
    static_assert(!(F64 && HYB), "F64 and HYB are different searches");
    // The exact search only ranks with the fp32 copy (the winner's index comes from the f64 copy), so it
    // reads a PACKED cell-sorted copy, 12 bytes per candidate: `sorted` then points at P12 triples
    // (launch_pack12), a quarter fewer bytes per gathered run (C4: 53.3 -> 50.3 us per launch).
    const P12 *s12 = reinterpret_cast<const P12 *>(sorted);
    constexpr bool S64 = F64 || HYB;                       // f64 source, transform and statistics
    constexpr int NACC = Acc<PLANE>::N;
    int prob, lb;
    long long row0;
    if (descs) {
        // batch of problems with their own clouds: find the problem this workgroup
        // belongs to (largest p with first_block <= blockIdx.x; wave-uniform)
        int lo = 0, hi = nprob - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (descs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        prob = lo;
        const ProbDesc d = descs[prob];
        lb = (int)blockIdx.x - d.first_block;
        bpp = d.nblocks;
        src += d.src_off;
        ns = d.ns;
        if constexpr (HYB) s12 += d.sorted_off; else sorted += d.sorted_off;
        if constexpr (S64) { src64 += d.src_off; sorted64 += d.sorted_off; }
        if constexpr (PLANE) {                             // normals are indexed like the (unsorted) target
            if (nrm) nrm += d.sorted_off;
            if (nrm64) nrm64 += d.sorted_off;
        }
        row0 = d.first_block;
        start += d.start_off;
        g = d.g;
        out_stride = 0;
        idx_out += d.out_off;
        d2_out += d.out_off;
        if (prevq_out) prevq_out += d.out_off;
    } else {
        // `bpp` workgroups per problem: problem b = blockIdx.x / bpp shares the
        // clouds and the grid with the others but has its own transform / state
        // (the yaw sweep of src/annotation.cpp:35-61 is 24 such problems).
        prob = blockIdx.x / bpp;
        lb = blockIdx.x - prob * bpp;
        row0 = (long long)prob * bpp;
    }
    if (st) st += prob;
    if (!load_loop_state(st, T32, T64, off, r2f)) return;
    if constexpr (S64) r2d = (double)r2f;                  // (double)(float)(r*r), also when the radius comes from the state
    idx_out += (long long)prob * out_stride;
    d2_out += (long long)prob * out_stride;
    if (prevq_out) prevq_out += (long long)prob * out_stride;   // the winners: the warm start of the next pass (grid_coop.hip)
    double acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++) acc[a] = 0.0;
    unsigned ncand = 0, ncand_all = 0;

    const int sub = threadIdx.x % G;                       // lane within the query group
    const int groups_per_block = kBlock / G;
    // XCD-aware chunking: workgroup b runs on XCD b % 8.  Give each XCD ONE
    // contiguous eighth of the (Morton-ordered) queries, so its private L2 holds
    // one spatial region of the target instead of a slice of everything.
    int vb = lb;
    if ((bpp & 7) == 0) vb = (lb & 7) * (bpp >> 3) + (lb >> 3);
    // consecutive virtual blocks take consecutive query chunks; a block strides by
    // one over its own contiguous share
    const int total_groups = bpp * groups_per_block;
    const int per_group = (ns + total_groups - 1) / total_groups;
    const int gid = vb * groups_per_block + threadIdx.x / G;
    const int i_begin = gid * per_group;
    const int i_end = min(i_begin + per_group, ns);
    __shared__ unsigned row_b[8][kBlock], row_e[8][kBlock];
    __shared__ float row_bound[8][kBlock];
    float4 keep_s = make_float4(0.f, 0.f, 0.f, 0.f);      // the query this lane accumulates
    int keep_i = 0;                                        // (F64: its index instead)
    unsigned keep_pos = 0xFFFFFFFFu;                       // ... and its winner's slot in `sorted`
    // exact search, one query per lane: the transformed query and the winner's f64 point stay in
    // registers from the f64 decision to the statistics (no second trip to memory for them)
    constexpr bool KEEPQ = HYB && ONE && G == 1;
    double keep_p[3] = {0.0, 0.0, 0.0};
    Pt64 keep_q = Pt64{0.0, 0.0, 0.0, 0ull};
    auto flush = [&]() {
        if (keep_pos != 0xFFFFFFFFu) {
            float4 n4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (KEEPQ) {
                double nx = 0.0, ny = 0.0, nz = 0.0;
                if (PLANE) {
                    if (nrm64) { const Pt64 n8 = nrm64[(unsigned)keep_q.w]; nx = n8.x; ny = n8.y; nz = n8.z; }
                    else { n4 = nrm[(unsigned)keep_q.w]; nx = n4.x; ny = n4.y; nz = n4.z; }
                }
                accumulate_pq_d<PLANE>(acc, keep_p[0], keep_p[1], keep_p[2], keep_q.x, keep_q.y, keep_q.z, nx, ny, nz, off);
            } else if constexpr (S64) {
                const Pt64 s8 = src64[keep_i], q8 = sorted64[keep_pos];
                double nx = 0.0, ny = 0.0, nz = 0.0;
                if (PLANE) {
                    if (nrm64) { const Pt64 n8 = nrm64[(unsigned)q8.w]; nx = n8.x; ny = n8.y; nz = n8.z; }
                    else { n4 = nrm[(unsigned)q8.w]; nx = n4.x; ny = n4.y; nz = n4.z; }
                }
                accumulate_pair_d<PLANE>(acc, s8.x, s8.y, s8.z, q8.x, q8.y, q8.z, nx, ny, nz, T64, off);
            } else {
                const float4 q4 = sorted[keep_pos];
                if (PLANE) n4 = nrm[__float_as_uint(q4.w)];
                accumulate_pair<PLANE>(acc, keep_s, q4, n4, T64, off);
            }
            keep_pos = 0xFFFFFFFFu;
        }
    };
    int it = 0;
    for (int i = i_begin; i < i_end; i++, it++) {
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float px, py, pz;
        double pxd = 0.0, pyd = 0.0, pzd = 0.0;
        if constexpr (S64) {
            // the reference's transform of a source point (PointCloud.cpp:75-80), in f64
            const Pt64 s8 = src64[i];
            pxd = T64.m[0] * s8.x + T64.m[1] * s8.y + T64.m[2] * s8.z + T64.m[3] * 1.0;
            pyd = T64.m[4] * s8.x + T64.m[5] * s8.y + T64.m[6] * s8.z + T64.m[7] * 1.0;
            pzd = T64.m[8] * s8.x + T64.m[9] * s8.y + T64.m[10] * s8.z + T64.m[11] * 1.0;
            px = (float)pxd; py = (float)pyd; pz = (float)pzd;
        } else {
            s4 = src[i];
            xform_point_f32(T32, s4, px, py, pz);
        }
        const int cx = cell_coord(px, g.mn[0], g.inv_h, g.dim[0]);
        const int cy = cell_coord(py, g.mn[1], g.inv_hs, g.dim[1]);
        const int cz = cell_coord(pz, g.mn[2], g.inv_hs, g.dim[2]);
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dim[0] - 1);
        // (d2, original index) packed so that ONE 64-bit unsigned compare is the
        // lexicographic test; d2 >= 0, so its IEEE bits are order preserving.  The
        // start key (r2f, 0) makes the acceptance strict: d2 < r2f.
        unsigned long long bkey = (unsigned long long)__float_as_uint(r2f) << 32;
        unsigned bpos = 0xFFFFFFFFu;
        double bd = r2d;                                   // F64: best d2 so far (strictly below r2d once set)
        unsigned bidx = 0xFFFFFFFFu;                       // F64: ... and its original index
        // HYB: rounding band.  p32 = fl(p64), q32 = fl(q64): the difference vector is off by at most
        // u (|p|+|q|) per component, u = 2^-24, and the fp32 evaluation of d2 adds 3u relative, so
        // |d64 - sqrt(d2_32)| <= u (2 |p| + 2.5 r).  E is more than twice that.
        float hyb_E = 0.f, hyb_rup = 0.f;
        TopK top;
        if constexpr (HYB) {
            const float r_f = sqrtf(r2f);
            hyb_rup = r_f * (1.0f + 2.4e-7f);
            hyb_E = 2.4e-7f * (fabsf(px) + fabsf(py) + fabsf(pz) + hyb_rup) + 4.8e-7f * hyb_rup;
            const float t = hyb_rup + 2.0f * hyb_E;
            top.init(t * t * (1.0f + 6e-7f));              // candidates at or beyond it never matter
        }
        // what a row bound is compared with: the best squared distance so far, as fp32
        auto best_f32 = [&]() {
            if constexpr (F64) { const float f = (float)bd; return f + f * 1e-6f; }   // rounded UP a little
            else if constexpr (HYB) return top.h[0];
            else return __uint_as_float((unsigned)(bkey >> 32));
        };
        // Rows (y,z) are visited nearest first and a row is SKIPPED when its slab cannot
        // hold a better candidate: every point of row (dy,dz) is at least
        // |(dist to the slab in y, in z)| away.  Margins: 1e-3 cell on each slab distance
        // (fp32 binning of query and candidates, see kGridMaxDim) and 1e-5 relative on the
        // squared bound (fp32 d2 rounding), so a skipped candidate has d2 > best strictly
        // -- it could neither win nor tie.
        const float fy = (py - g.mn[1]) * g.inv_hs - (float)cy;     // position inside the cell, [0,1)
        const float fz = (pz - g.mn[2]) * g.inv_hs - (float)cz;
        const float mgn = HYB ? 1e-3f + 4.0f * hyb_E * g.inv_hs : 1e-3f;    // HYB: plus the rounding band
        const float lo_y = fmaxf(fy - mgn, 0.f), hi_y = fmaxf(1.0f - fy - mgn, 0.f);
        const float lo_z = fmaxf(fz - mgn, 0.f), hi_z = fmaxf(1.0f - fz - mgn, 0.f);
        const float h2 = g.hs * g.hs * (1.0f - 1e-5f);
        // squared lower bound of row offset (dy, dz), compile-time offsets
        auto row_bound_of = [&](int dy, int dz) {
            const float ey = dy == 0 ? 0.f : (dy < 0 ? lo_y + (float)(-dy - 1) : hi_y + (float)(dy - 1));
            const float ez = dz == 0 ? 0.f : (dz < 0 ? lo_z + (float)(-dz - 1) : hi_z + (float)(dz - 1));
            return (ey * ey + ez * ez) * h2;
        };
        // (begin, end) of the run of the 3 x-adjacent cells of row (dy, dz); they are
        // adjacent in memory.  (32-bit cell arithmetic: the cell count is below 2^31)
        // ONE 16-byte load (4-byte aligned) fetches the starts of cells x0 .. x0+3: half the L1
        // accesses of two scalar loads (the table carries a few entries of slack at its end).
        auto row_range = [&](int dy, int dz, bool want, unsigned &rb, unsigned &re) {
            typedef unsigned u4a __attribute__((ext_vector_type(4), aligned(4)));
            const int z = cz + dz, y = cy + dy;
            const bool ok = want && (x0 <= x1) && z >= 0 && z < g.dim[2] && y >= 0 && y < g.dim[1];
            const int row = ((ok ? z : 0) * g.dim[1] + (ok ? y : 0)) * g.dim[0];
            u4a v = {0u, 0u, 0u, 0u};
            if (ok) v = *reinterpret_cast<const u4a *>(start + row + x0);
            const int span = x1 + 1 - x0;                        // 1, 2 or 3 cells
            rb = v.x;
            re = span >= 3 ? v.w : (span == 2 ? v.z : v.y);
            if (!ok) re = 0u;
        };
        unsigned base = 0, e = 0;
        // One batch of U*G candidates of the current run.  A slot past the end of the run
        // re-reads the run's first point: evaluating a candidate twice cannot change the
        // (d2, index) minimum, and it saves the per-slot guard.
#ifdef VISMA_GRID_DEBUG_TRIPS
        int dbg_trips = 0;
#endif
        auto batch = [&]() {
#ifdef VISMA_GRID_DEBUG_TRIPS
            dbg_trips++;
#endif
            unsigned jc[U];
            if constexpr (F64) {
                Pt64 q[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const unsigned ju = base + sub + u * G;
                    jc[u] = ju < e ? ju : base;
                    q[u] = sorted64[jc[u]];
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    // flann L2 (dist.h:159-176): result += diff * diff over x, y, z
                    const double dx = q[u].x - pxd, dy = q[u].y - pyd, dz = q[u].z - pzd;
                    double d = dx * dx;
                    d += dy * dy;
                    d += dz * dz;
                    const unsigned id = (unsigned)q[u].w;
                    // strictly nearer, or as near with a lower index (a re-read of the winner is neither)
                    const bool lt = d < bd || (d == bd && id < bidx && bidx != 0xFFFFFFFFu);
                    bd = lt ? d : bd;
                    bidx = lt ? id : bidx;
                    bpos = lt ? jc[u] : bpos;
                }
            } else if constexpr (HYB) {
                // the whole batch from ONE address (immediate offsets): slots past the run's end read
                // whatever follows in the array (it has kSortedSlack entries of slack) and are masked
                float4 q[U];
                const unsigned b0 = base + sub;
                const unsigned left = b0 < e ? e - b0 : 0u;
                const P12 *qp = s12 + b0;                          // 12 bytes per candidate (see P12)
#pragma unroll
                for (int u = 0; u < U; u++) { const P12 t = qp[u * G]; q[u] = make_float4(t.x, t.y, t.z, 0.f); }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    float d = sqdist_f32(q[u], px, py, pz);
                    d = (unsigned)(u * G) < left ? d : INFINITY;      // a padding slot is not a candidate
                    top.insert(d, b0 + u * G);
                }
            } else {
                float4 q[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const unsigned ju = base + sub + u * G;
                    jc[u] = ju < e ? ju : base;
                    q[u] = sorted[jc[u]];
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const float d = sqdist_f32(q[u], px, py, pz);
                    const unsigned long long key =
                        ((unsigned long long)__float_as_uint(d) << 32) | __float_as_uint(q[u].w);
                    const bool lt = key < bkey;
                    bkey = lt ? key : bkey;
                    bpos = lt ? jc[u] : bpos;
                }
            }
            base += U * G;
        };
        // Every lane group walks ITS OWN row list (LDS, visiting order) with its own
        // cursor, one batch per trip of a single loop.  (A loop nest "for row: for batch"
        // makes the wave run max-over-queries batches for EVERY row -- 4-5x more load
        // instructions than a query needs.)
        // (round 5: the smallest bound among the rows SKIPPED -- with the runner-up and the outside of the 27 cells it is
        //  the lower bound LB this pass leaves to the certificates of grid_coop.hip)
        float lb_skip2 = INFINITY;
        bool near_tie = false;
        auto walk = [&]() {
            int kk = 0;
            for (;;) {
                while (base >= e && kk < 8) {                   // group-uniform: next live row
                    const unsigned nb = row_b[kk][threadIdx.x], ne = row_e[kk][threadIdx.x];
                    const float bound = row_bound[kk][threadIdx.x];
                    ++kk;
                    // the group's current best bounds what any lane still needs
                    float gbest = best_f32();
#pragma unroll
                    for (int m = G >> 1; m > 0; m >>= 1) gbest = fminf(gbest, __shfl_xor(gbest, m, 64));
                    if (bound > gbest) { lb_skip2 = fminf(lb_skip2, bound); continue; }
                    base = nb;
                    e = ne;
                    if (cand_count && sub == 0) ncand += e - base;
                }
                if (base >= e) break;
                batch();
            }
        };
        {
            // centre row + the 8 rows around it: all 18 (begin, end) loads in flight at once
            constexpr int order[9] = {4, 1, 3, 5, 7, 0, 2, 6, 8};
            unsigned rb[9], re[9];
#pragma unroll
            for (int k = 0; k < 9; k++) row_range(k % 3 - 1, k / 3 - 1, true, rb[k], re[k]);
#pragma unroll
            for (int kk = 1; kk < 9; kk++) {
                const int k = order[kk];
                row_b[kk - 1][threadIdx.x] = rb[k];
                row_e[kk - 1][threadIdx.x] = re[k];
                row_bound[kk - 1][threadIdx.x] = row_bound_of(k % 3 - 1, k / 3 - 1);
                if (cand_count && sub == 0) ncand_all += 1u;       // profiling: cell-table rows looked up
            }
            base = rb[4];
            e = re[4];                                           // centre row: never pruned
            if (cand_count && sub == 0) { ncand_all += 1u; ncand += e - base; }
            walk();
        }
        if (g.sub == 2) {
            // half-pitch rows: the 16 rows of the second ring, nearest first, in two passes
            // of 8.  They matter only while the best so far is farther than half a cell
            // (sparse places, queries without a neighbour): most lanes skip both passes.
            constexpr int r2y[16] = {0, -2, 2, 0, -1, 1, -2, 2, -2, 2, -1, 1, -2, 2, -2, 2};
            constexpr int r2z[16] = {-2, 0, 0, 2, -2, -2, -1, -1, 1, 1, 2, 2, -2, -2, 2, 2};
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
                float gbest = best_f32();
#pragma unroll
                for (int m = G >> 1; m > 0; m >>= 1) gbest = fminf(gbest, __shfl_xor(gbest, m, 64));
                bool any = false;
                unsigned rb[8], re[8];
                float bd[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    bd[k] = row_bound_of(r2y[pass * 8 + k], r2z[pass * 8 + k]);
                    const bool want = !(bd[k] > gbest);
                    any = any || want;
                    row_range(r2y[pass * 8 + k], r2z[pass * 8 + k], want, rb[k], re[k]);
                }
                if (!any) continue;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    row_b[k][threadIdx.x] = rb[k];
                    row_e[k][threadIdx.x] = re[k];
                    row_bound[k][threadIdx.x] = bd[k];
                    if (cand_count && sub == 0) ncand_all += 1u;
                }
                base = e = 0;
                walk();
            }
        }
        if constexpr (HYB) {
            constexpr int NT = VISMA_GRID_NTOP;
            if (G > 1) {
                // butterfly merge over the G lanes: the NT best of the union, and the next value
#pragma unroll
                for (int m = G >> 1; m > 0; m >>= 1) {
                    float oh[NT + 1];
                    unsigned op[NT];
#pragma unroll
                    for (int k = 0; k <= NT; k++) oh[k] = __shfl_xor(top.h[k], m, 64);
#pragma unroll
                    for (int k = 0; k < NT; k++) op[k] = (unsigned)__shfl_xor((int)top.p[k], m, 64);
                    // (the lanes of a pair must end up with the SAME set: insert in a fixed order)
                    const bool mine_first = (threadIdx.x & m) == 0;
                    TopK t2 = top;
                    float ih[NT + 1];
                    unsigned ip[NT];
#pragma unroll
                    for (int k = 0; k <= NT; k++) { ih[k] = mine_first ? oh[k] : top.h[k]; if (!mine_first) t2.h[k] = oh[k]; }
#pragma unroll
                    for (int k = 0; k < NT; k++) { ip[k] = mine_first ? op[k] : top.p[k]; if (!mine_first) t2.p[k] = op[k]; }
#pragma unroll
                    for (int k = 0; k < NT; k++) t2.insert(ih[k], ip[k]);
                    t2.h[NT] = fminf(t2.h[NT], ih[NT]);
                    top = t2;
                }
            }
            // decisive?  the candidates inside the rounding band of the best are ranked in f64
            if (sub == 0 && top.p[0] != 0xFFFFFFFFu) {
                // sqrt(h[k]) <= sqrt(h[0]) + 2E, tested on the squares (rounded UP: a candidate wrongly
                // taken for "in the band" only costs its f64 evaluation)
                const float s1 = sqrtf(top.h[0]) + 2.0f * hyb_E;
                const float s1sq = s1 * s1 * (1.0f + 4e-7f);
                bool in[NT + 1];
                in[0] = true;
#pragma unroll
                for (int k = 1; k <= NT; k++) in[k] = in[k - 1] && top.h[k] <= s1sq;
                near_tie = in[1];                            // (a second candidate inside the band: the runner-up is not bounded)
                auto rank = [&](unsigned pos) {
                    const Pt64 c8 = sorted64[pos];
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
                    if constexpr (KEEPQ) {
                        keep_q.x = lt ? c8.x : keep_q.x; keep_q.y = lt ? c8.y : keep_q.y;
                        keep_q.z = lt ? c8.z : keep_q.z; keep_q.w = lt ? c8.w : keep_q.w;
                    }
                };
                if (!in[NT]) {
                    rank(top.p[0]);
#pragma unroll
                    for (int k = 1; k < NT; k++)
                        if (in[k] && top.p[k] != 0xFFFFFFFFu) rank(top.p[k]);
                } else {
                    // more candidates inside one band than positions kept: every candidate of the 27 cells
                    // that the fp32 filter cannot exclude, in f64
                    const float sl = fminf(sqrtf(top.h[0]), hyb_rup) + 2.0f * hyb_E;
                    const float L = sl * sl * (1.0f + 6e-7f);
#pragma unroll 1
                    for (int k = 0; k < 9; k++) {
                        unsigned rb, re;
                        row_range(k % 3 - 1, k / 3 - 1, true, rb, re);
#pragma unroll 1
                        for (unsigned j = rb; j < re; j++)
                        {
                            const P12 t = s12[j];
                            if (sqdist_f32(make_float4(t.x, t.y, t.z, 0.f), px, py, pz) <= L) rank(j);
                        }
                    }
                }
            }
            if (G > 1) {                                       // every lane of the group learns the winner
                bpos = (unsigned)__shfl((int)bpos, (threadIdx.x & 63) & ~(G - 1), 64);
                bidx = (unsigned)__shfl((int)bidx, (threadIdx.x & 63) & ~(G - 1), 64);
                bd = __shfl(bd, (threadIdx.x & 63) & ~(G - 1), 64);
            }
        } else if (G > 1) {
            // butterfly merge over the G lanes: smallest (d2, index) wins everywhere
#pragma unroll
            for (int m = G >> 1; m > 0; m >>= 1) {
                if constexpr (F64) {
                    const double od = __shfl_xor(bd, m, 64);
                    const unsigned oi = (unsigned)__shfl_xor((int)bidx, m, 64);
                    const unsigned op = (unsigned)__shfl_xor((int)bpos, m, 64);
                    // 0xFFFFFFFF ("none") is the largest index, so it loses every tie at r2d
                    if (od < bd || (od == bd && oi < bidx)) { bd = od; bidx = oi; bpos = op; }
                } else {
                    const unsigned ohi = (unsigned)__shfl_xor((int)(unsigned)(bkey >> 32), m, 64);
                    const unsigned olo = (unsigned)__shfl_xor((int)(unsigned)bkey, m, 64);
                    const unsigned op = (unsigned)__shfl_xor((int)bpos, m, 64);
                    const unsigned long long ok = ((unsigned long long)ohi << 32) | olo;
                    if (ok < bkey) { bkey = ok; bpos = op; }
                }
            }
        }
        if (sub == 0) {
            if constexpr (HYB) {
                if (prevq_out) {
                    // the state of grid_coop.hip: the winner's f64 point, original index | LB << 32.  LB (round 5; 0 before: the
                    // first warm pass of every registration searched every query): what every target point but the winner is
                    // at least away -- the second-best candidate examined (every row visited was examined whole), the rows
                    // skipped, the outside of the 27 cells; margins and rounding band as grid_coop.hip forms its own.
                    float lbv = 0.f;
                    if (g.sub == 1 && VISMA_GRID_NTOP >= 2) {
                        const float fx = (px - g.mn[0]) * g.inv_h - (float)cx;
                        const float lo_x = fmaxf(fx - mgn, 0.f), hi_x = fmaxf(1.0f - fx - mgn, 0.f);
                        const float face = fminf(fminf(fminf(lo_x, hi_x), fminf(lo_y, hi_y)), fminf(lo_z, hi_z));
                        const float ux = fx + (float)cx, uy = fy + (float)cy, uz = fz + (float)cz;
                        const float far = fmaxf(fmaxf(fmaxf(-ux, ux - (float)g.dim[0]), fmaxf(-uy, uy - (float)g.dim[1])),
                                                fmaxf(-uz, uz - (float)g.dim[2])) - 2.0f * mgn;
                        const float outc = fmaxf(1.0f + face, far);
                        float lb2 = fminf(fminf(outc * outc * h2, lb_skip2), top.h[1]);
                        if (bpos == 0xFFFFFFFFu) lb2 = fminf(lb2, top.h[0]);     // nothing accepted: the best candidate bounds like the rest
                        const float lb = sqrtf(lb2) * (1.0f - 1e-6f) - 4.0f * hyb_E;
                        lbv = near_tie ? 0.f : fminf(fmaxf(lb, 0.f), 3.0e38f);
                    }
                    const unsigned long long lbw = (unsigned long long)__float_as_uint(lbv) << 32;
                    Pt64 w8;
                    w8.x = w8.y = w8.z = __longlong_as_double(-1ll);
                    w8.w = 0xFFFFFFFFull | lbw;
                    if (bpos != 0xFFFFFFFFu) { w8 = sorted64[bpos]; w8.w = (w8.w & 0xFFFFFFFFull) | lbw; }
                    prevq_out[i] = w8;
                }
            }
            if constexpr (S64) {
#ifdef VISMA_GRID_DEBUG_TRIPS   /* measurement build: trips of this query | of its wave | of its wave's matched queries */
                {
                    int wmax = dbg_trips, wmax_m = bpos == 0xFFFFFFFFu ? 0 : dbg_trips;
                    for (int m = 32; m > 0; m >>= 1) {
                        wmax = max(wmax, __shfl_xor(wmax, m, 64));
                        wmax_m = max(wmax_m, __shfl_xor(wmax_m, m, 64));
                    }
                    idx_out[i] = dbg_trips | ((bpos != 0xFFFFFFFFu) << 7) | (wmax << 8) | (wmax_m << 16);
                }
#else
                idx_out[i] = (bpos == 0xFFFFFFFFu) ? -1 : (int)bidx;
#endif
                d2_out[i] = (float)bd;
                if (d64_out) d64_out[i] = bd;                // (target-sharded ranks compare shards in f64)
            } else {
                idx_out[i] = (bpos == 0xFFFFFFFFu) ? -1 : (int)(unsigned)bkey;
                d2_out[i] = __uint_as_float((unsigned)(bkey >> 32));
            }
        }
        // lane (it mod G) of the group keeps this query's winner
        if ((it % G) == sub) { keep_s = s4; keep_i = i; keep_pos = bpos; }
        if constexpr (KEEPQ) { keep_p[0] = pxd; keep_p[1] = pyd; keep_p[2] = pzd; }
        if (!ONE && (it % G) == G - 1) flush();
    }
    flush();
    block_reduce_store<NACC>(acc, partials, fold.tickets != nullptr);
    if (cand_count) {
        // profiling only: candidates examined / candidates in the full 27-cell blocks.
        // One slot pair per workgroup (mod 4096) -- thousands of atomics on ONE address
        // cost ~40 us per launch and used to distort the very time being measured.
        unsigned long long c = ncand, ca = ncand_all;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            c += __shfl_down(c, o, 64);
            ca += __shfl_down(ca, o, 64);
        }
        if ((threadIdx.x & 63) == 0 && ca) {
            unsigned long long *slot = cand_count + 2 * (blockIdx.x & 4095);
            atomicAdd(slot, c);
            atomicAdd(slot + 1, ca);
        }
    }
    if (fold.tickets) fused_fold<PLANE, kBlock, false, kSolveInFold>(fold, partials, row0, lb, bpp, prob);
}

template <bool PLANE, int G, int U>
static void launch_grid_t(int nblocks, hipStream_t stream, const float4 *src, int ns,
                          const float4 *sorted, const unsigned *start, const GridParams &g,
                          const float4 *nrm, const Xform32 &T32, const Xform64 &T64,
                          const Offset64 &off, float r2f, int *idx_out, float *d2_out,
                          double *partials, unsigned long long *cand, const DevIcpState *st,
                          int nprob, long long out_stride, const Pt64 *src64, const Pt64 *sorted64, double r2d,
                          const Pt64 *nrm64, int exact, const FoldArgs &fold, double *d64_out, Pt64 *prevq_out)
{
    // one query per lane? (see ONE above)
    const long long total_groups = (long long)nblocks * (kBlock / G);
    const bool one = ((long long)ns + total_groups - 1) / total_groups <= G;
#define VISMA_GRID_LAUNCH(ONE_, F64_, HYB_)                                                                     \
    hipLaunchKernelGGL((nn_grid_reduce_kernel<PLANE, G, U, ONE_, F64_, HYB_>), dim3(nblocks * nprob),             \
                       dim3(kBlock), 0, stream, src, ns, sorted, start, g, nrm, T32, T64, off, r2f, idx_out,      \
                       d2_out, partials, cand, st, nblocks, out_stride, (const ProbDesc *)nullptr, nprob, src64,  \
                       sorted64, r2d, nrm64, fold, d64_out, prevq_out)
    if (src64 && exact) {
        if (one) VISMA_GRID_LAUNCH(true, false, true); else VISMA_GRID_LAUNCH(false, false, true);
    } else if (src64) {
        if (one) VISMA_GRID_LAUNCH(true, true, false); else VISMA_GRID_LAUNCH(false, true, false);
    } else {
        if (one) VISMA_GRID_LAUNCH(true, false, false); else VISMA_GRID_LAUNCH(false, false, false);
    }
#undef VISMA_GRID_LAUNCH
}

hipError_t launch_nn_grid_reduce(const float4 *src, int64_t ns, const float4 *sorted,
                                 const unsigned *start, const GridParams &g,
                                 const float4 *tgt_normals, const Xform32 &T32, const Xform64 &T64,
                                 const double frame_offset[3], float r2f, int point_to_plane,
                                 int32_t *idx_out, float *d2_out, double *partials,
                                 int max_partial_blocks, int *nblocks_out, int lanes_per_query,
                                 unsigned long long *cand_count, const DevIcpState *st,
                                 int nprob, int64_t out_stride, hipStream_t stream, const Pt64 *src64,
                                 const Pt64 *sorted64, double r2d, const Pt64 *nrm64, int exact,
                                 const FoldArgs *fold, double *d64_out, Pt64 *prevq_io, int warm, const Xform64 *Tprev,
                                 const PersistArgs *persist, Pt64 *ru_io, const RingTable *ring)
{
    if ((src64 == nullptr) != (sorted64 == nullptr)) return hipErrorInvalidValue;
    // (the persistent launch exists for the certificate kernel only)
    if (persist && !(lanes_per_query == kCoopLanes && src64 && exact && g.sub == 1 && prevq_io)) return hipErrorInvalidValue;
    Offset64 off;
    for (int a = 0; a < 3; a++) off.v[a] = frame_offset ? frame_offset[a] : 0.0;
    const FoldArgs fa = fold ? *fold : FoldArgs{};
    const int G = lanes_per_query % 100;
    int U = lanes_per_query / 100;
    if (U == 0) U = (G >= 8) ? 2 : 4;
    int64_t want = (ns * G + kBlock - 1) / kBlock;
    int nblocks = (int)(want > max_partial_blocks ? max_partial_blocks : want);
    if (nblocks < 1) nblocks = 1;
    if (g.ring > 0) {
        // cells smaller than the radius: the ring search (grid_ring.hip), G lanes per query
        if (!src64 || persist || !ring) return hipErrorInvalidValue;
        // (`sorted` is the packed 12-byte copy when the search is the exact one: HipEngine::search_sorted)
        hipError_t e = launch_nn_ring(G, nblocks, nprob, (int)ns, src64, sorted64, exact ? (const float *)sorted : nullptr, start, g,
                                      *ring, tgt_normals, nrm64, T64, off, r2f,
                                      point_to_plane, idx_out, d2_out, d64_out, prevq_io, warm & 1, partials, cand_count, st,
                                      (long long)out_stride, fa, stream);
        if (nblocks_out) *nblocks_out = nblocks;
        return e;
    }
    if (lanes_per_query == kCoopLanes) {
        if (src64 && exact && g.sub == 1 && prevq_io) {
            // the flattened exact search (grid_coop.hip): `sorted` is the packed 12-byte copy
            const bool one = (ns + (int64_t)nblocks * kBlock - 1) / ((int64_t)nblocks * kBlock) <= 1;
            hipError_t e = launch_nn_coop(nblocks * nprob, nblocks, nprob, nullptr, (int)ns, (const float *)sorted, start, g,
                                          tgt_normals, nrm64, T64, off, r2f, point_to_plane, one ? 1 : 0, idx_out, d2_out,
                                          partials, cand_count, st, (long long)out_stride, src64, sorted64, fa, d64_out,
                                          prevq_io, warm, stream, Tprev, persist, ru_io);
            if (nblocks_out) *nblocks_out = nblocks;
            return e;
        }
        U = 8;                                             // not the exact search: the lane-serial kernel
    }
#define VISMA_GRID_CASE(GG, UU)                                                                    \
    if (G == GG && U == UU) {                                                                      \
        if (point_to_plane)                                                                        \
            launch_grid_t<true, GG, UU>(nblocks, stream, src, (int)ns, sorted, start, g,           \
                                        tgt_normals, T32, T64, off, r2f, idx_out, d2_out,          \
                                        partials, cand_count, st, nprob, (long long)out_stride, src64, sorted64, r2d, nrm64, exact, fa, d64_out, prevq_io);   \
        else                                                                                       \
            launch_grid_t<false, GG, UU>(nblocks, stream, src, (int)ns, sorted, start, g,          \
                                         tgt_normals, T32, T64, off, r2f, idx_out, d2_out,         \
                                         partials, cand_count, st, nprob, (long long)out_stride, src64, sorted64, r2d, nrm64, exact, fa, d64_out, prevq_io);  \
        launched = true;                                                                           \
    }
    bool launched = false;
    // the (G, U) pairs the driver's policies use (HipEngine::grid_lanes: 801, 1201, 402, 802, 804, 408) and no others
    // since round 4 -- every pair is 12 kernel instantiations: PLANE x ONE x {fp32, f64, exact}
    VISMA_GRID_CASE(1, 8) VISMA_GRID_CASE(1, 12) VISMA_GRID_CASE(2, 4) VISMA_GRID_CASE(2, 8)
    VISMA_GRID_CASE(4, 8) VISMA_GRID_CASE(8, 4)
    if (!launched) return hipErrorInvalidValue;
#undef VISMA_GRID_CASE
    if (nblocks_out) *nblocks_out = nblocks;
    return hipGetLastError();
}

template <int G, int U, bool ONE, bool F64, bool HYB, bool PLANE = false>
static void launch_grid_batch_t(int total_blocks, hipStream_t stream, const float4 *src,
                                const float4 *sorted, const unsigned *start, const ProbDesc *descs,
                                int nprob, int *idx_out, float *d2_out, double *partials,
                                const DevIcpState *st, const Pt64 *src64, const Pt64 *sorted64, const FoldArgs &fold,
                                unsigned long long *cand, const float4 *nrm = nullptr, const Pt64 *nrm64 = nullptr,
                                Pt64 *prevq_out = nullptr)
{
    const Xform32 T32{};
    const Xform64 T64{};
    const Offset64 off{};
    const GridParams g{};
    hipLaunchKernelGGL((nn_grid_reduce_kernel<PLANE, G, U, ONE, F64, HYB>), dim3(total_blocks), dim3(kBlock), 0, stream,
                       src, 0, sorted, start, g, nrm, T32, T64, off, 0.f, idx_out,
                       d2_out, partials, cand, st, 1, 0ll, descs, nprob, src64, sorted64, 0.0,
                       nrm64, fold, (double *)nullptr, prevq_out);
}

// lanes_per_query = G + 100 * U; one_per_lane: every problem has at most G queries per lane group;
// src64 / sorted64 (both or neither): the f64 search (exact = 0) or the exact fp32+f64 search (exact = 1),
// arrays concatenated like src / sorted
hipError_t launch_nn_grid_reduce_batch(const float4 *src, const float4 *sorted, const unsigned *start,
                                       const ProbDesc *descs, int nprob, int total_blocks,
                                       int32_t *idx_out, float *d2_out, double *partials,
                                       int lanes_per_query, int one_per_lane, const DevIcpState *st,
                                       hipStream_t stream, const Pt64 *src64, const Pt64 *sorted64, int exact,
                                       const FoldArgs *fold, unsigned long long *cand_count,
                                       const float4 *nrm, const Pt64 *nrm64, Pt64 *prevq_io, int warm)
{
    if (!st || !descs || (src64 == nullptr) != (sorted64 == nullptr)) return hipErrorInvalidValue;
    const bool plane = nrm != nullptr || nrm64 != nullptr;
    if (plane && src64 && !exact) return hipErrorInvalidValue;      // point-to-plane batches: exact or fp32 search
    const FoldArgs fa = fold ? *fold : FoldArgs{};
    const int G = lanes_per_query % 100;
    int U = lanes_per_query / 100;
    if (lanes_per_query == kCoopLanes) {
        if (src64 && exact && prevq_io) {
            // (batches never use half-pitch rows: grid_plan(..., max_sub = 1))
            const Xform64 T64{};
            const Offset64 off{};
            const GridParams g{};
            return launch_nn_coop(total_blocks, 1, nprob, descs, 0, (const float *)sorted, start, g, nrm, nrm64, T64, off,
                                  0.f, plane ? 1 : 0, one_per_lane, idx_out, d2_out, partials, cand_count, st, 0ll, src64,
                                  sorted64, fa, nullptr, prevq_io, warm, stream);
        }
        U = 8;
    }
    bool launched = false;
#define VISMA_BATCH_ARGS total_blocks, stream, src, sorted, start, descs, nprob, idx_out, d2_out, partials, st
#define VISMA_BATCH_CASE(GG, UU)                                                                              \
    if (G == GG && U == UU && plane) {                                                                        \
        if (src64 && one_per_lane)                                                                            \
            launch_grid_batch_t<GG, UU, true, false, true, true>(VISMA_BATCH_ARGS, src64, sorted64, fa, cand_count, nrm, nrm64, prevq_io);   \
        else if (src64)                                                                                       \
            launch_grid_batch_t<GG, UU, false, false, true, true>(VISMA_BATCH_ARGS, src64, sorted64, fa, cand_count, nrm, nrm64, prevq_io);  \
        else if (one_per_lane)                                                                                \
            launch_grid_batch_t<GG, UU, true, false, false, true>(VISMA_BATCH_ARGS, nullptr, nullptr, fa, cand_count, nrm, nrm64, prevq_io); \
        else                                                                                                  \
            launch_grid_batch_t<GG, UU, false, false, false, true>(VISMA_BATCH_ARGS, nullptr, nullptr, fa, cand_count, nrm, nrm64, prevq_io); \
        launched = true;                                                                                      \
    } else if (G == GG && U == UU) {                                                                          \
        if (src64 && exact && one_per_lane)                                                                   \
            launch_grid_batch_t<GG, UU, true, false, true>(VISMA_BATCH_ARGS, src64, sorted64, fa, cand_count, nullptr, nullptr, prevq_io);            \
        else if (src64 && exact)                                                                              \
            launch_grid_batch_t<GG, UU, false, false, true>(VISMA_BATCH_ARGS, src64, sorted64, fa, cand_count, nullptr, nullptr, prevq_io);           \
        else if (src64 && one_per_lane)                                                                       \
            launch_grid_batch_t<GG, UU, true, true, false>(VISMA_BATCH_ARGS, src64, sorted64, fa, cand_count, nullptr, nullptr, prevq_io);            \
        else if (src64)                                                                                       \
            launch_grid_batch_t<GG, UU, false, true, false>(VISMA_BATCH_ARGS, src64, sorted64, fa, cand_count, nullptr, nullptr, prevq_io);           \
        else if (one_per_lane)                                                                                \
            launch_grid_batch_t<GG, UU, true, false, false>(VISMA_BATCH_ARGS, nullptr, nullptr, fa, cand_count, nullptr, nullptr, prevq_io);          \
        else                                                                                                  \
            launch_grid_batch_t<GG, UU, false, false, false>(VISMA_BATCH_ARGS, nullptr, nullptr, fa, cand_count, nullptr, nullptr, prevq_io);         \
        launched = true;                                                                                      \
    }
    // (the batch policy of HipEngine::run_batch: 801, 804, 402 -- the other pairs went in round 4)
    VISMA_BATCH_CASE(4, 8) VISMA_BATCH_CASE(1, 8) VISMA_BATCH_CASE(2, 4)
#undef VISMA_BATCH_CASE
#undef VISMA_BATCH_ARGS
    if (!launched) return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace visma
