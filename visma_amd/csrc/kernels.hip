// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the ICP path.
//
//  nn_brute_kernel   fused source transform + brute-force nearest neighbour
//                    (replaces PointCloud::Transform + the KDTreeFlann lookups of
//                    O3D/Core/Registration/Registration.cpp:41-96, 175-178)
//  reduce_kernel     per-correspondence SE(3) Jacobian/residual + wavefront
//                    shuffle reduction (replaces the gathers of
//                    src/constrained_ICP.cpp:30-35 and ComputeJTJandJTr,
//                    O3D/Core/Utility/Eigen.cpp:137-182)
//  finalize_kernel   fixed-order fold of the workgroup partials into the 38
//                    statistics (6x6 J^T J, 6x1 J^T r, cross-covariance, K, r^2)
//
// Arithmetic contract (checked bit-for-bit against oracle/icp_oracle.c vk_*):
//   p  = fmaf(r0,sx, fmaf(r1,sy, fmaf(r2,sz, t)))          fp32, per row
//   d2 = fmaf(dz,dz, fmaf(dy,dy, dx*dx)),  d = q - p        fp32
//   accept iff d2 < r2f; lowest target index wins exact ties.
// Statistics are accumulated in f64 from p = T64 * (double)s and q widened.
#include "device_common.h"

#include <algorithm>
#include "host_math.hpp"

#include <math.h>

namespace visma {

// ------------------------------------------------------------------------
// NN: fused transform + brute force
// ------------------------------------------------------------------------
// Each thread keeps SPT transformed source points in registers; the workgroup
// streams its share of the target through a double-buffered LDS tile (one
// coalesced 16-B global load per lane per 256 targets, broadcast ds_read_b128
// in the inner loop).  The inner loop only tracks the running minimum
// (v_min3_f32 over two targets); which 64-target sub-chunk produced it is
// recorded once per sub-chunk, and the exact index is recovered by the
// reduction kernel, which re-evaluates that one sub-chunk.
//
// HYB (the exact search, brute-force flavour): the source point is transformed in f64 and rounded
// (p32 = fl(p64)), the acceptance radius is widened by the rounding band, and besides the best
// squared distance and its sub-chunk the kernel keeps `second`: the smallest sub-chunk minimum
// among the OTHER sub-chunks (same min3 per pair of targets as before -- the minimum is taken per
// sub-chunk and folded into (best, second) once per 64 targets).  The reduction kernel re-ranks in
// f64 what the band leaves undecided.
template <int SPT, bool HYB = false>
__global__ __launch_bounds__(kBlock) void nn_brute_kernel(
    const float4 *__restrict__ src, int ns, const float4 *__restrict__ tgt,
    int chunks_total, int chunks_per_split, int src_tiles, int nsplits, Xform32 T,
    float r2f, unsigned long long *__restrict__ keys, long long ns_pad,
    const DevIcpState *__restrict__ st, const Pt64 *__restrict__ src64 = nullptr, Xform64 T64 = Xform64{},
    float *__restrict__ second_out = nullptr)
{
    __shared__ float4 lds[2][kTChunk];
    const int tid = threadIdx.x;
    {
        Offset64 o;
        if (!load_loop_state(st, T, T64, o, r2f)) return;
    }

    // XCD-aware (tile, split) decode: workgroup b runs on XCD b % 8, so give
    // each XCD its own residue class of target splits -- its L2 then streams
    // 1/8 of the target instead of all of it.
    int tile, split;
    {
        const int b = blockIdx.x;
        if ((nsplits & 7) == 0) {
            const int xcd = b & 7, s = b >> 3;
            tile = s % src_tiles;
            split = xcd + 8 * (s / src_tiles);
        } else {
            tile = b % src_tiles;
            split = b / src_tiles;
        }
    }

    float px[SPT], py[SPT], pz[SPT], best[SPT], second[SPT];
    unsigned win[SPT];
#pragma unroll
    for (int k = 0; k < SPT; k++) {
        const int i = tile * (kBlock * SPT) + k * kBlock + tid;
        if constexpr (HYB) {
            Pt64 s8;
            s8.x = s8.y = s8.z = 0.0;
            s8.w = 0ull;
            if (i < ns) s8 = src64[i];
            px[k] = (float)(T64.m[0] * s8.x + T64.m[1] * s8.y + T64.m[2] * s8.z + T64.m[3] * 1.0);
            py[k] = (float)(T64.m[4] * s8.x + T64.m[5] * s8.y + T64.m[6] * s8.z + T64.m[7] * 1.0);
            pz[k] = (float)(T64.m[8] * s8.x + T64.m[9] * s8.y + T64.m[10] * s8.z + T64.m[11] * 1.0);
            best[k] = exact_band_limit(px[k], py[k], pz[k], r2f);    // candidates at or beyond it never matter
        } else {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < ns) s = src[i];
            xform_point_f32(T, s, px[k], py[k], pz[k]);
            best[k] = r2f;      // strict d2 < r2f acceptance is built in
        }
        second[k] = INFINITY;
        win[k] = 0xFFFFFFFFu;   // "no correspondence"
    }

    const int c0 = split * chunks_per_split;
    int c1 = c0 + chunks_per_split;
    if (c1 > chunks_total) c1 = chunks_total;

    if (c0 < c1) {
        const float4 *g = tgt + (long long)c0 * kTChunk;
        float4 r0 = g[tid], r1 = g[tid + kBlock];
        lds[0][tid] = r0;
        lds[0][tid + kBlock] = r1;
        __syncthreads();
        int buf = 0;
        for (int c = c0; c < c1; ++c) {
            const bool has_next = (c + 1 < c1);
            if (has_next) {  // issue the next chunk's global loads early
                g += kTChunk;
                r0 = g[tid];
                r1 = g[tid + kBlock];
            }
#pragma unroll 1
            for (int sb = 0; sb < kTChunk / kSub; ++sb) {
                float before[SPT];
#pragma unroll
                for (int k = 0; k < SPT; k++) {
                    before[k] = best[k];
                    if constexpr (HYB) best[k] = INFINITY;     // the minimum of THIS sub-chunk
                }
                const float4 *t = &lds[buf][sb * kSub];
#pragma unroll 8
                for (int j = 0; j < kSub; j += 2) {
                    const float4 qa = t[j], qb = t[j + 1];
#pragma unroll
                    for (int k = 0; k < SPT; k++) {
                        const float da = sqdist_f32(qa, px[k], py[k], pz[k]);
                        const float db = sqdist_f32(qb, px[k], py[k], pz[k]);
                        best[k] = min3_f32(best[k], da, db);
                    }
                }
                const unsigned sid = (unsigned)(c * (kTChunk / kSub) + sb);
#pragma unroll
                for (int k = 0; k < SPT; k++) {
                    if constexpr (HYB) {
                        const float m = best[k];
                        const bool improved = m < before[k];
                        second[k] = fminf(second[k], improved ? before[k] : m);
                        best[k] = improved ? m : before[k];
                        win[k] = improved ? sid : win[k];
                    } else {
                        win[k] = (best[k] < before[k]) ? sid : win[k];
                    }
                }
            }
            if (has_next) {
                lds[buf ^ 1][tid] = r0;
                lds[buf ^ 1][tid + kBlock] = r1;
            }
            __syncthreads();
            buf ^= 1;
        }
    }

    unsigned long long *out = keys + (long long)split * ns_pad;
#pragma unroll
    for (int k = 0; k < SPT; k++) {
        const long long i = (long long)tile * (kBlock * SPT) + k * kBlock + tid;
        out[i] = ((unsigned long long)__float_as_uint(best[k]) << 32) | win[k];
        if constexpr (HYB) second_out[(long long)split * ns_pad + i] = second[k];
    }
}

NNLaunch nn_plan(int64_t ns, int64_t nt_pad)
{
    NNLaunch p;
    const int64_t chunks = nt_pad / kTChunk;
    // small clouds: fewer points per thread so that more workgroups exist
    p.spt = (ns >= 64 * 1024) ? kSptLarge : kSptSmall;
    const int64_t tile = (int64_t)kBlock * p.spt;
    p.src_tiles = (int)((ns + tile - 1) / tile);
    if (p.src_tiles < 1) p.src_tiles = 1;
    // aim for ~8 workgroups per CU (256 CUs) so the tail is short
    const int64_t want = 2048;
    int64_t splits = (want + p.src_tiles - 1) / p.src_tiles;
    if (splits > chunks) splits = chunks;
    if (splits >= 8) splits = (splits / 8) * 8;  // XCD-aware decode needs %8
    if (splits < 1) splits = 1;
    // every split must own at least one chunk
    int64_t per = (chunks + splits - 1) / splits;
    splits = (chunks + per - 1) / per;
    if (splits >= 8 && (splits & 7)) {
        // re-balance to a multiple of 8 when possible without empty splits
        int64_t s8 = (splits / 8) * 8;
        int64_t per8 = (chunks + s8 - 1) / s8;
        if ((chunks + per8 - 1) / per8 == s8) splits = s8;
    }
    p.tgt_splits = (int)(splits < 1 ? 1 : splits);
    return p;
}

hipError_t launch_nn_brute(const float4 *src, int64_t ns, const float4 *tgt,
                           int64_t nt_pad, const Xform32 &T, float r2f,
                           unsigned long long *keys, int64_t ns_pad,
                           const NNLaunch &plan, const DevIcpState *st, hipStream_t stream,
                           const Pt64 *src64, const Xform64 *T64, float *second)
{
    const int chunks = (int)(nt_pad / kTChunk);
    const int per = (chunks + plan.tgt_splits - 1) / plan.tgt_splits;
    const dim3 grid((unsigned)(plan.src_tiles * plan.tgt_splits));
    if (src64 && second) {
        const Xform64 t64 = T64 ? *T64 : Xform64{};
        if (plan.spt == kSptLarge)
            hipLaunchKernelGGL((nn_brute_kernel<kSptLarge, true>), grid, dim3(kBlock), 0, stream, src, (int)ns, tgt,
                               chunks, per, plan.src_tiles, plan.tgt_splits, T, r2f, keys, (long long)ns_pad, st, src64,
                               t64, second);
        else
            hipLaunchKernelGGL((nn_brute_kernel<kSptSmall, true>), grid, dim3(kBlock), 0, stream, src, (int)ns, tgt,
                               chunks, per, plan.src_tiles, plan.tgt_splits, T, r2f, keys, (long long)ns_pad, st, src64,
                               t64, second);
        return hipGetLastError();
    }
    if (plan.spt == kSptLarge)
        hipLaunchKernelGGL(nn_brute_kernel<kSptLarge>, grid, dim3(kBlock), 0, stream, src,
                           (int)ns, tgt, chunks, per, plan.src_tiles, plan.tgt_splits, T, r2f,
                           keys, (long long)ns_pad, st);
    else
        hipLaunchKernelGGL(nn_brute_kernel<kSptSmall>, grid, dim3(kBlock), 0, stream, src,
                           (int)ns, tgt, chunks, per, plan.src_tiles, plan.tgt_splits, T, r2f,
                           keys, (long long)ns_pad, st);
    return hipGetLastError();
}

// ------------------------------------------------------------------------
// Reduction: merge + refine + Jacobian/residual + wave-shuffle reduce
// ------------------------------------------------------------------------
// Exact flavour of the brute-force reduction (the search kernel ran with HYB): per query
//   * the smallest (d2, sub-chunk) over the splits, and `so` = the smallest fp32 d2 any OTHER
//     sub-chunk holds (the other splits' best, the winning split's `second`);
//   * every candidate of the winning sub-chunk inside the rounding band of the best is ranked in
//     f64 (flann dist.h:159-176; lowest index on exact ties; accepted iff d2 < (double)(float)(r*r));
//   * if `so` reaches into the band, other sub-chunks may hold the true winner: the WAVE re-scans the
//     whole target for that query (fp32 filter, f64 rank) -- about one query in a thousand.
// Statistics from the f64 coordinates: the results equal the exact grid search's.
template <bool PLANE>
__global__ __launch_bounds__(kBlock) void reduce_exact_kernel(
    const Pt64 *__restrict__ src64, int ns, const float4 *__restrict__ tgt, const Pt64 *__restrict__ tgt64, int nt,
    const float4 *__restrict__ nrm, const Pt64 *__restrict__ nrm64, const unsigned long long *__restrict__ keys,
    const float *__restrict__ second, int nsplits, long long ns_pad, Xform64 T64, Offset64 off, float r2f,
    int *__restrict__ idx_out, float *__restrict__ d2_out, double *__restrict__ partials,
    const DevIcpState *__restrict__ st, BrutePend pend)
{
    constexpr int NACC = Acc<PLANE>::N;
    {
        Xform32 t32;
        if (!load_loop_state(st, t32, T64, off, r2f)) return;
    }
    const double r2d = (double)r2f;
    double acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++) acc[a] = 0.0;
    const int lane = threadIdx.x & 63;
    const int nloop = ((ns + kBlock - 1) / kBlock + gridDim.x - 1) / gridDim.x;   // uniform trip count (wave-wide scans inside)
    for (int it = 0; it < nloop; it++) {
        const int i = (it * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
        const bool valid = i < ns;
        Pt64 s8;
        s8.x = s8.y = s8.z = 0.0;
        s8.w = 0ull;
        if (valid) s8 = src64[i];
        const double pxd = T64.m[0] * s8.x + T64.m[1] * s8.y + T64.m[2] * s8.z + T64.m[3] * 1.0;
        const double pyd = T64.m[4] * s8.x + T64.m[5] * s8.y + T64.m[6] * s8.z + T64.m[7] * 1.0;
        const double pzd = T64.m[8] * s8.x + T64.m[9] * s8.y + T64.m[10] * s8.z + T64.m[11] * 1.0;
        const float px = (float)pxd, py = (float)pyd, pz = (float)pzd;
        // (1) merge the splits
        unsigned long long key = ~0ull;
        int smin = 0;
        if (valid) {
            key = keys[i];
            for (int s = 1; s < nsplits; s++) {
                const unsigned long long k2 = keys[(long long)s * ns_pad + i];
                if (k2 < key) { key = k2; smin = s; }
            }
        }
        float so = INFINITY;
        if (valid)
            for (int s = 0; s < nsplits; s++) {
                const float v = (s == smin) ? second[(long long)s * ns_pad + i]
                                            : __uint_as_float((unsigned)(keys[(long long)s * ns_pad + i] >> 32));
                so = fminf(so, v);
            }
        const unsigned win = (unsigned)(key & 0xFFFFFFFFull);
        const float b1 = __uint_as_float((unsigned)(key >> 32));
        float r_up;
        const float E = exact_band(px, py, pz, r2f, &r_up);
        const float sl = fminf(sqrtf(b1), r_up) + 2.0f * E;
        const float L = sl * sl * (1.0f + 6e-7f);
        double bd = r2d;
        unsigned bidx = 0xFFFFFFFFu;
        auto rank = [&](unsigned j, double qx, double qy, double qz) {
            const double dx = qx - pxd, dy = qy - pyd, dz = qz - pzd;
            double d = dx * dx;
            d += dy * dy;
            d += dz * dz;
            const bool lt = d < bd || (d == bd && j < bidx && bidx != 0xFFFFFFFFu);
            bd = lt ? d : bd;
            bidx = lt ? j : bidx;
        };
        // (2) the winning sub-chunk, every candidate inside the band
        if (valid && win != 0xFFFFFFFFu) {
            const unsigned j0 = win * kSub;
#pragma unroll 4
            for (int j = 0; j < kSub; ++j) {
                const unsigned jj = j0 + j;
                if (jj < (unsigned)nt && sqdist_f32(tgt[jj], px, py, pz) <= L) {
                    const Pt64 c8 = tgt64[jj];
                    rank(jj, c8.x, c8.y, c8.z);
                }
            }
        }
        // (3) other sub-chunks reach into the band: the wave scans the whole target for that query
        bool undecided = valid && win != 0xFFFFFFFFu && so <= L;
        bool deferred = false;
        if (undecided && pend.count) {
            // hand the query to the rescan passes (every workgroup, one sweep of the target for all of them)
            const int slot = atomicAdd(pend.count, 1);
            if (slot < kBrutePendCap) {
                pend.q32[slot] = make_float4(px, py, pz, L);
                pend.q64[slot] = Pt64{pxd, pyd, pzd, (unsigned long long)i};
                pend.best[slot] = (unsigned long long)__double_as_longlong(r2d);
                pend.best_idx[slot] = 0xFFFFFFFFu;
                idx_out[i] = -2 - slot;                       // resolved by reduce_pending_kernel
                deferred = true;
            }
        }
        unsigned long long need = __ballot(undecided && !deferred);
        while (need) {
            const int srcl = __ffsll((long long)need) - 1;
            need &= need - 1ull;
            const float qpx = __shfl(px, srcl, 64), qpy = __shfl(py, srcl, 64), qpz = __shfl(pz, srcl, 64);
            const float qL = __shfl(L, srcl, 64);
            const double qxd = __shfl(pxd, srcl, 64), qyd = __shfl(pyd, srcl, 64), qzd = __shfl(pzd, srcl, 64);
            double wd = r2d;
            unsigned wi = 0xFFFFFFFFu;
            for (int j = lane; j < nt; j += 64) {
                if (sqdist_f32(tgt[j], qpx, qpy, qpz) <= qL) {
                    const Pt64 c8 = tgt64[j];
                    const double dx = c8.x - qxd, dy = c8.y - qyd, dz = c8.z - qzd;
                    double d = dx * dx;
                    d += dy * dy;
                    d += dz * dz;
                    const bool lt = d < wd || (d == wd && (unsigned)j < wi && wi != 0xFFFFFFFFu);
                    wd = lt ? d : wd;
                    wi = lt ? (unsigned)j : wi;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {                  // (d2, index) minimum over the wave
                const double od = __shfl_xor(wd, o, 64);
                const unsigned oi = (unsigned)__shfl_xor((int)wi, o, 64);
                if (od < wd || (od == wd && oi < wi)) { wd = od; wi = oi; }
            }
            if (lane == srcl) { bd = wd; bidx = wi; }          // (a scan of everything supersedes the sub-chunk result)
        }
        if (valid && !deferred) {
            idx_out[i] = bidx == 0xFFFFFFFFu ? -1 : (int)bidx;
            d2_out[i] = (float)bd;
            if (bidx != 0xFFFFFFFFu) {
                const Pt64 q8 = tgt64[bidx];
                double nx = 0.0, ny = 0.0, nz = 0.0;
                if (PLANE) {
                    if (nrm64) { const Pt64 n8 = nrm64[bidx]; nx = n8.x; ny = n8.y; nz = n8.z; }
                    else { const float4 n4 = nrm[bidx]; nx = n4.x; ny = n4.y; nz = n4.z; }
                }
                accumulate_pair_d<PLANE>(acc, s8.x, s8.y, s8.z, q8.x, q8.y, q8.z, nx, ny, nz, T64, off);
            }
        }
    }
    block_reduce_store<NACC>(acc, partials);
}

// The pending queries of reduce_exact_kernel against the whole target, all of them in one sweep: every
// thread takes targets j, j + stride, ... and tests each against every pending query (fp32 filter, then
// the reference's f64 distance).  PASS 0: the smallest f64 d2 per query (atomic minimum of its bits --
// order preserving for d2 >= 0; the start value is the radius bound, so the acceptance stays strict).
// PASS 1: the lowest index among the targets AT that distance.
template <int PASS>
__global__ __launch_bounds__(256) void brute_rescan_kernel(const float4 *__restrict__ tgt, const Pt64 *__restrict__ tgt64,
                                                           int nt, BrutePend pend, const DevIcpState *__restrict__ st)
{
    if (st && !st->active) return;
    const int F = min(*pend.count, kBrutePendCap);
    if (F <= 0) return;
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < nt; j += (long long)gridDim.x * 256) {
        const float4 t = tgt[j];
        for (int f = 0; f < F; f++) {
            const float4 q = pend.q32[f];                         // (uniform: a scalar load)
            if (sqdist_f32(t, q.x, q.y, q.z) <= q.w) {
                const Pt64 c8 = tgt64[j], q8 = pend.q64[f];
                // flann L2 (dist.h:159-176): result += diff * diff over x, y, z
                const double dx = c8.x - q8.x, dy = c8.y - q8.y, dz = c8.z - q8.z;
                double d = dx * dx;
                d += dy * dy;
                d += dz * dz;
                const unsigned long long bits = (unsigned long long)__double_as_longlong(d);
                if (PASS == 0) {
                    if (bits < pend.best[f]) atomicMin(&pend.best[f], bits);
                } else if (bits == pend.best[f]) {
                    atomicMin(&pend.best_idx[f], (unsigned)j);
                }
            }
        }
    }
}

// ... and their correspondences and moments, by the thread that owns the source slot (same mapping as
// reduce_exact_kernel, so the sums do not depend on the order in which the queries were listed)
template <bool PLANE>
__global__ __launch_bounds__(kBlock) void reduce_pending_kernel(
    const Pt64 *__restrict__ src64, int ns, const Pt64 *__restrict__ tgt64, const float4 *__restrict__ nrm,
    const Pt64 *__restrict__ nrm64, Xform64 T64, Offset64 off, float r2f, int *__restrict__ idx_out,
    float *__restrict__ d2_out, double *__restrict__ partials, const DevIcpState *__restrict__ st, BrutePend pend)
{
    constexpr int NACC = Acc<PLANE>::N;
    {
        Xform32 t32;
        if (!load_loop_state(st, t32, T64, off, r2f)) return;
    }
    double acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++) acc[a] = 0.0;
    const int nloop = ((ns + kBlock - 1) / kBlock + gridDim.x - 1) / gridDim.x;
    for (int it = 0; it < nloop; it++) {
        const int i = (it * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
        if (i >= ns) continue;
        const int code = idx_out[i];
        if (code > -2) continue;
        const int slot = -2 - code;
        const double bd = __longlong_as_double((long long)pend.best[slot]);
        // (nothing strictly inside the radius in f64: the start value is still there, whatever pass 1 found at it)
        const unsigned bidx = bd < (double)r2f ? pend.best_idx[slot] : 0xFFFFFFFFu;
        idx_out[i] = bidx == 0xFFFFFFFFu ? -1 : (int)bidx;
        d2_out[i] = (float)bd;
        if (bidx != 0xFFFFFFFFu) {
            const Pt64 s8 = src64[i], q8 = tgt64[bidx];
            double nx = 0.0, ny = 0.0, nz = 0.0;
            if (PLANE) {
                if (nrm64) { const Pt64 n8 = nrm64[bidx]; nx = n8.x; ny = n8.y; nz = n8.z; }
                else { const float4 n4 = nrm[bidx]; nx = n4.x; ny = n4.y; nz = n4.z; }
            }
            accumulate_pair_d<PLANE>(acc, s8.x, s8.y, s8.z, q8.x, q8.y, q8.z, nx, ny, nz, T64, off);
        }
    }
    block_reduce_store<NACC>(acc, partials);
}

template <bool PLANE>
__global__ __launch_bounds__(kBlock) void reduce_kernel(
    const float4 *__restrict__ src, int ns, const float4 *__restrict__ tgt,
    const float4 *__restrict__ nrm, const unsigned long long *__restrict__ keys,
    int nsplits, long long ns_pad, Xform32 T32, Xform64 T64, Offset64 off, float r2f,
    int *__restrict__ idx_out, float *__restrict__ d2_out, double *__restrict__ partials,
    const DevIcpState *__restrict__ st)
{
    constexpr int NACC = Acc<PLANE>::N;
    if (!load_loop_state(st, T32, T64, off, r2f)) return;
    double acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++) acc[a] = 0.0;

    for (int i = blockIdx.x * kBlock + threadIdx.x; i < ns; i += gridDim.x * kBlock) {
        // (1) merge the per-split candidates: smallest (d2, sub-chunk id)
        unsigned long long key = keys[i];
        for (int s = 1; s < nsplits; s++) {
            const unsigned long long k2 = keys[(long long)s * ns_pad + i];
            key = (k2 < key) ? k2 : key;
        }
        const unsigned win = (unsigned)(key & 0xFFFFFFFFull);
        const float best = __uint_as_float((unsigned)(key >> 32));
        const float4 s4 = src[i];
        int idx = -1;
        float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (win != 0xFFFFFFFFu) {
            // (2) recover the exact index: first j of the winning sub-chunk
            // whose distance (same fp32 arithmetic) equals the minimum
            float px, py, pz;
            xform_point_f32(T32, s4, px, py, pz);
            const float4 *t = tgt + (long long)win * kSub;
#pragma unroll 4
            for (int j = kSub - 1; j >= 0; --j) {
                const float4 q = t[j];
                if (sqdist_f32(q, px, py, pz) == best) {
                    idx = (int)(win * kSub) + j;
                    q4 = q;
                }
            }
        }
        idx_out[i] = idx;
        d2_out[i] = best;
        if (idx >= 0) {
            // (3) Jacobian / residual rows in f64
            float4 n4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (PLANE) n4 = nrm[idx];
            accumulate_pair<PLANE>(acc, s4, q4, n4, T64, off);
        }
    }
    // (4) wavefront shuffle reduction -> LDS across the 4 waves -> partials
    block_reduce_store<NACC>(acc, partials);
}

// ------------------------------------------------------------------------
// Target-sharded ranks: global winners by a min-reduce of keys
// ------------------------------------------------------------------------
// key = (fp32 d2 bits << 32) | GLOBAL target index; all ones = no neighbour.  The
// smallest key over the ranks is the (d2, lowest index) winner of the whole target.
__global__ __launch_bounds__(256) void shard_keys_kernel(const int *__restrict__ idx,
                                                         const float *__restrict__ d2, int ns,
                                                         unsigned offset,
                                                         unsigned long long *__restrict__ keys)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ns) return;
    const int j = idx[i];
    keys[i] = j < 0 ? ~0ull
                    : (((unsigned long long)__float_as_uint(d2[i]) << 32) | ((unsigned)j + offset));
}

// After the min-reduce: every rank writes the global (index, d2) of every source
// point, and accumulates the Jacobian/residual moments of the winners IT owns.
template <bool PLANE>
__global__ __launch_bounds__(kBlock) void shard_accumulate_kernel(
    const float4 *__restrict__ src, int ns, const unsigned long long *__restrict__ keys,
    const float4 *__restrict__ tgt, long long nt_local, unsigned offset,
    const float4 *__restrict__ nrm, Xform64 T64, Offset64 off, float r2f,
    int *__restrict__ idx_out, float *__restrict__ d2_out, double *__restrict__ partials)
{
    constexpr int NACC = Acc<PLANE>::N;
    double acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++) acc[a] = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < ns; i += gridDim.x * kBlock) {
        const unsigned long long key = keys[i];
        const bool hit = key != ~0ull;
        const unsigned gj = (unsigned)key;
        idx_out[i] = hit ? (int)gj : -1;
        d2_out[i] = hit ? __uint_as_float((unsigned)(key >> 32)) : r2f;
        const long long lj = (long long)gj - (long long)offset;
        if (hit && lj >= 0 && lj < nt_local) {
            float4 n4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (PLANE) n4 = nrm[lj];
            accumulate_pair<PLANE>(acc, src[i], tgt[lj], n4, T64, off);
        }
    }
    block_reduce_store<NACC>(acc, partials);
}

// ---- f64 variant: every shard ran the exact search -------------------------------------------
__global__ __launch_bounds__(256) void shard_keys64_kernel(const int *__restrict__ idx, const double *__restrict__ d64,
                                                           int ns, unsigned long long *__restrict__ keys)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ns) return;
    // d2 >= 0: the bits of a double order like the value
    keys[i] = idx[i] < 0 ? ~0ull : (unsigned long long)__double_as_longlong(d64[i]);
}

__global__ __launch_bounds__(256) void shard_claim64_kernel(const int *__restrict__ idx, const double *__restrict__ d64,
                                                            const unsigned long long *__restrict__ gkeys, int ns,
                                                            unsigned offset, unsigned long long *__restrict__ claim)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ns) return;
    const bool mine = idx[i] >= 0 && (unsigned long long)__double_as_longlong(d64[i]) == gkeys[i];
    claim[i] = mine ? (unsigned long long)((unsigned)idx[i] + offset) : ~0ull;
}

template <bool PLANE>
__global__ __launch_bounds__(kBlock) void shard_accumulate64_kernel(
    const Pt64 *__restrict__ src64, int ns, const unsigned long long *__restrict__ gkeys,
    const unsigned long long *__restrict__ claim, const Pt64 *__restrict__ tgt64, long long nt_local, unsigned offset,
    const float4 *__restrict__ nrm, const Pt64 *__restrict__ nrm64, Xform64 T64, Offset64 off, double r2d,
    int *__restrict__ idx_out, float *__restrict__ d2_out, double *__restrict__ partials,
    const DevIcpState *__restrict__ st)
{
    constexpr int NACC = Acc<PLANE>::N;
    if (st) {                                              // device loop: transform, frame and radius from the state
        Xform32 t32;
        float r2f = 0.f;
        if (!load_loop_state(st, t32, T64, off, r2f)) return;
        r2d = (double)r2f;
    }
    double acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++) acc[a] = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < ns; i += gridDim.x * kBlock) {
        const unsigned long long key = gkeys[i];
        const bool hit = key != ~0ull;
        const unsigned gj = (unsigned)claim[i];
        idx_out[i] = hit ? (int)gj : -1;
        d2_out[i] = hit ? (float)__longlong_as_double((long long)key) : (float)r2d;
        const long long lj = (long long)gj - (long long)offset;
        if (hit && lj >= 0 && lj < nt_local) {
            const Pt64 s8 = src64[i], q8 = tgt64[lj];
            double nx = 0.0, ny = 0.0, nz = 0.0;
            if (PLANE) {
                if (nrm64) { const Pt64 n8 = nrm64[lj]; nx = n8.x; ny = n8.y; nz = n8.z; }
                else { const float4 n4 = nrm[lj]; nx = n4.x; ny = n4.y; nz = n4.z; }
            }
            accumulate_pair_d<PLANE>(acc, s8.x, s8.y, s8.z, q8.x, q8.y, q8.z, nx, ny, nz, T64, off);
        }
    }
    block_reduce_store<NACC>(acc, partials);
}

hipError_t launch_shard_keys64(const int32_t *idx, const double *d64, int64_t ns, unsigned long long *keys,
                               hipStream_t stream)
{
    if (ns > 0)
        hipLaunchKernelGGL(shard_keys64_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, stream, idx, d64,
                           (int)ns, keys);
    return hipGetLastError();
}

hipError_t launch_shard_claim64(const int32_t *idx, const double *d64, const unsigned long long *gkeys, int64_t ns,
                                unsigned offset, unsigned long long *claim, hipStream_t stream)
{
    if (ns > 0)
        hipLaunchKernelGGL(shard_claim64_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, stream, idx, d64,
                           gkeys, (int)ns, offset, claim);
    return hipGetLastError();
}

hipError_t launch_shard_accumulate64(const Pt64 *src64, int64_t ns, const unsigned long long *gkeys,
                                     const unsigned long long *claim, const Pt64 *tgt64, int64_t nt_local,
                                     unsigned offset, const float4 *tgt_normals, const Pt64 *nrm64,
                                     const Xform64 &T64, const double frame_offset[3], double r2d,
                                     int point_to_plane, int32_t *idx_out, float *d2_out, double *partials,
                                     int max_partial_blocks, int *nblocks_out, hipStream_t stream,
                                     const DevIcpState *st)
{
    Offset64 off;
    for (int a = 0; a < 3; a++) off.v[a] = frame_offset ? frame_offset[a] : 0.0;
    int64_t want = (ns + kBlock - 1) / kBlock;
    int nblocks = (int)(want > max_partial_blocks ? max_partial_blocks : want);
    if (nblocks < 1) nblocks = 1;
    if (point_to_plane)
        hipLaunchKernelGGL(shard_accumulate64_kernel<true>, dim3(nblocks), dim3(kBlock), 0, stream, src64, (int)ns,
                           gkeys, claim, tgt64, (long long)nt_local, offset, tgt_normals, nrm64, T64, off, r2d, idx_out,
                           d2_out, partials, st);
    else
        hipLaunchKernelGGL(shard_accumulate64_kernel<false>, dim3(nblocks), dim3(kBlock), 0, stream, src64, (int)ns,
                           gkeys, claim, tgt64, (long long)nt_local, offset, tgt_normals, nrm64, T64, off, r2d, idx_out,
                           d2_out, partials, st);
    if (nblocks_out) *nblocks_out = nblocks;
    return hipGetLastError();
}

hipError_t launch_shard_keys(const int32_t *idx, const float *d2, int64_t ns, unsigned offset,
                             unsigned long long *keys, hipStream_t stream)
{
    if (ns > 0)
        hipLaunchKernelGGL(shard_keys_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, stream, idx, d2,
                           (int)ns, offset, keys);
    return hipGetLastError();
}

hipError_t launch_shard_accumulate(const float4 *src, int64_t ns, const unsigned long long *keys,
                                   const float4 *tgt, int64_t nt_local, unsigned offset,
                                   const float4 *tgt_normals, const Xform64 &T64,
                                   const double frame_offset[3], float r2f, int point_to_plane,
                                   int32_t *idx_out, float *d2_out, double *partials,
                                   int max_partial_blocks, int *nblocks_out, hipStream_t stream)
{
    Offset64 off;
    for (int a = 0; a < 3; a++) off.v[a] = frame_offset ? frame_offset[a] : 0.0;
    int64_t want = (ns + kBlock - 1) / kBlock;
    int nblocks = (int)(want > max_partial_blocks ? max_partial_blocks : want);
    if (nblocks < 1) nblocks = 1;
    if (point_to_plane)
        hipLaunchKernelGGL(shard_accumulate_kernel<true>, dim3(nblocks), dim3(kBlock), 0, stream, src, (int)ns,
                           keys, tgt, (long long)nt_local, offset, tgt_normals, T64, off, r2f, idx_out, d2_out,
                           partials);
    else
        hipLaunchKernelGGL(shard_accumulate_kernel<false>, dim3(nblocks), dim3(kBlock), 0, stream, src, (int)ns,
                           keys, tgt, (long long)nt_local, offset, tgt_normals, T64, off, r2f, idx_out, d2_out,
                           partials);
    if (nblocks_out) *nblocks_out = nblocks;
    return hipGetLastError();
}

template <bool PLANE>
__global__ __launch_bounds__(1024) void finalize_kernel(const double *__restrict__ partials,
                                                        int nblocks,
                                                        double *__restrict__ stats,
                                                        double *host_out, unsigned long long seq)
{
    fold_partials<PLANE>(partials, nblocks, stats);
    // optional: publish straight to mapped host memory (saves the publish launch)
    if (host_out) publish_tagged_stats(stats, host_out, seq);
}

// Copy the statistics into host-visible (mapped, coherent) memory and then raise
// a sequence number there: the host spin-waits on it instead of paying a
// DMA packet + stream-synchronise wake-up per ICP iteration.
__global__ void publish_stats_kernel(const double *__restrict__ stats, double *host_out,
                                     unsigned long long seq)
{
    publish_tagged_stats(stats, host_out, seq);
}

hipError_t launch_publish_stats(const double *stats, double *host_out, unsigned long long seq,
                                hipStream_t stream)
{
    hipLaunchKernelGGL(publish_stats_kernel, dim3(1), dim3(64), 0, stream, stats, host_out, seq);
    return hipGetLastError();
}

// ------------------------------------------------------------------------
// One-shot all-reduce of the 38 statistics over xGMI (no RCCL call, no host hop)
// ------------------------------------------------------------------------
// Every rank owns a mailbox of nranks x 38 granules {value, sequence tag} (16 bytes each, in
// uncached device memory, mapped into the peers through hipIpc).  Lane a of ONE workgroup
// stores its statistic as ONE 16-byte granule into slot [rank][a] of every peer's mailbox
// (remote stores over xGMI; a granule is written by one store, so its tag validates its value:
// no flag, no fence), then waits for the nranks granules of statistic a in its OWN mailbox and
// sums them in rank order -- every rank forms the same sum, bit for bit, so the ranks keep
// identical transforms without a broadcast.  Tags are the call count: nothing to reset.
// The mailbox has two halves used alternately (call count parity): a rank can only start
// call k+2 after every peer has SENT call k+1, which a peer does only after it has finished
// reading call k, so a granule is never overwritten before its reader has seen it.
__global__ __launch_bounds__(64) void ipc_allreduce_kernel(const double *stats_in, double *stats_out,
                                                           IpcPeers peers, int rank, int nranks,
                                                           unsigned long long *seq_dev, double *host_out,
                                                           unsigned long long host_seq, int *timeout_flag,
                                                           long long max_spins)
{
    const int a = threadIdx.x;
    bool late = false;
    // the number of this exchange lives in device memory (see FoldArgs::ipc_seq_dev): one wave, lane 0 advances it
    unsigned long long seq = 0ull;
    if (a == 0) {
        seq = __hip_atomic_load(seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
        __hip_atomic_store(seq_dev, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    seq = ((unsigned long long)(unsigned)__shfl((int)(unsigned)(seq >> 32), 0, 64) << 32) |
          (unsigned long long)(unsigned)__shfl((int)(unsigned)seq, 0, 64);
    const double sum = ipc_exchange(a, a < kNStats ? stats_in[a] : 0.0, peers, rank, nranks, seq, timeout_flag,
                                    max_spins, late);
    // (a sum with a missing term is never handed on: NaN stops the device loop's solve)
    if (a < kNStats) stats_out[a] = late ? __longlong_as_double(0x7ff8000000000000ll) : sum;
    // nothing is published after a timeout: the host sees no tags, then reads the flag
    if (host_out && !__any(late)) publish_tagged_stats(stats_out, host_out, host_seq);
}

hipError_t launch_ipc_allreduce(const double *stats_in, double *stats_out, const IpcPeers &peers, int rank,
                                int nranks, unsigned long long *seq_dev, double *host_out, unsigned long long host_seq,
                                int *timeout_flag, hipStream_t stream, long long max_spins)
{
    hipLaunchKernelGGL(ipc_allreduce_kernel, dim3(1), dim3(64), 0, stream, stats_in, stats_out, peers, rank, nranks,
                       seq_dev, host_out, host_seq, timeout_flag, max_spins);
    return hipGetLastError();
}

int reduce_max_blocks() { return 1024; }

hipError_t launch_finalize(const double *partials, int nblocks, int point_to_plane,
                           double *stats_out, hipStream_t stream, double *host_out,
                           unsigned long long seq)
{
    if (point_to_plane)
        hipLaunchKernelGGL(finalize_kernel<true>, dim3(1), dim3(1024), 0, stream, partials,
                           nblocks, stats_out, host_out, seq);
    else
        hipLaunchKernelGGL(finalize_kernel<false>, dim3(1), dim3(1024), 0, stream, partials,
                           nblocks, stats_out, host_out, seq);
    return hipGetLastError();
}

hipError_t launch_reduce(const float4 *src, int64_t ns, const float4 *tgt,
                         const float4 *tgt_normals, const unsigned long long *keys,
                         int nsplits, int64_t ns_pad, const Xform32 &T32,
                         const Xform64 &T64, const double frame_offset[3], float r2f,
                         int point_to_plane, int32_t *idx_out, float *d2_out, double *partials,
                         int max_partial_blocks, double *stats_out, const DevIcpState *st,
                         int *nblocks_out, hipStream_t stream, double *host_out,
                         unsigned long long seq, const BruteExact *ex)
{
    Offset64 off;
    for (int a = 0; a < 3; a++) off.v[a] = frame_offset ? frame_offset[a] : 0.0;
    int nblocks = (int)((ns + kBlock - 1) / kBlock);
    if (nblocks > max_partial_blocks) nblocks = max_partial_blocks;
    if (nblocks < 1) nblocks = 1;
    if (ex && ex->src64) {
        const BrutePend pend = ex->pend;
        if (pend.count) {
            hipError_t e0 = hipMemsetAsync(pend.count, 0, sizeof(int), stream);
            if (e0 != hipSuccess) return e0;
        }
        if (point_to_plane)
            hipLaunchKernelGGL(reduce_exact_kernel<true>, dim3(nblocks), dim3(kBlock), 0, stream, ex->src64, (int)ns, tgt,
                               ex->tgt64, (int)ex->nt, tgt_normals, ex->nrm64, keys, ex->second, nsplits, (long long)ns_pad,
                               T64, off, r2f, idx_out, d2_out, partials, st, pend);
        else
            hipLaunchKernelGGL(reduce_exact_kernel<false>, dim3(nblocks), dim3(kBlock), 0, stream, ex->src64, (int)ns, tgt,
                               ex->tgt64, (int)ex->nt, tgt_normals, ex->nrm64, keys, ex->second, nsplits, (long long)ns_pad,
                               T64, off, r2f, idx_out, d2_out, partials, st, pend);
        if (pend.count) {
            // the undecided queries: two sweeps of the target by all workgroups, then their moments into a
            // second set of partial rows (the fold sums 2 * nblocks rows)
            const int rblocks = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (ex->nt + 255) / 256));   // (an empty target: one idle workgroup)
            hipLaunchKernelGGL(brute_rescan_kernel<0>, dim3(rblocks), dim3(256), 0, stream, tgt, ex->tgt64, (int)ex->nt, pend, st);
            hipLaunchKernelGGL(brute_rescan_kernel<1>, dim3(rblocks), dim3(256), 0, stream, tgt, ex->tgt64, (int)ex->nt, pend, st);
            double *p2 = partials + (size_t)nblocks * kReduceAcc;
            if (point_to_plane)
                hipLaunchKernelGGL(reduce_pending_kernel<true>, dim3(nblocks), dim3(kBlock), 0, stream, ex->src64, (int)ns,
                                   ex->tgt64, tgt_normals, ex->nrm64, T64, off, r2f, idx_out, d2_out, p2, st, pend);
            else
                hipLaunchKernelGGL(reduce_pending_kernel<false>, dim3(nblocks), dim3(kBlock), 0, stream, ex->src64, (int)ns,
                                   ex->tgt64, tgt_normals, ex->nrm64, T64, off, r2f, idx_out, d2_out, p2, st, pend);
            nblocks *= 2;
        }
    } else if (point_to_plane)
        hipLaunchKernelGGL(reduce_kernel<true>, dim3(nblocks), dim3(kBlock), 0, stream, src,
                           (int)ns, tgt, tgt_normals, keys, nsplits, (long long)ns_pad, T32,
                           T64, off, r2f, idx_out, d2_out, partials, st);
    else
        hipLaunchKernelGGL(reduce_kernel<false>, dim3(nblocks), dim3(kBlock), 0, stream, src,
                           (int)ns, tgt, tgt_normals, keys, nsplits, (long long)ns_pad, T32,
                           T64, off, r2f, idx_out, d2_out, partials, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (nblocks_out) *nblocks_out = nblocks;
    if (!stats_out) return hipSuccess;   // the caller folds the partial rows itself
    return launch_finalize(partials, nblocks, point_to_plane, stats_out, stream, host_out, seq);
}

// ------------------------------------------------------------------------
// small utility kernels
// ------------------------------------------------------------------------
__global__ void fill_inf_kernel(float4 *dst, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = make_float4(INFINITY, INFINITY, INFINITY, 0.f);
}

hipError_t launch_fill_inf(float4 *dst, int64_t n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_inf_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       dst, (long long)n);
    return hipGetLastError();
}

__global__ void pack_float4_kernel(const float *src, int stride, float4 *dst, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float *s = src + i * stride;
        dst[i] = make_float4(s[0], s[1], s[2], 0.f);
    }
}

hipError_t launch_pack_float4(const float *src, int stride, float4 *dst, int64_t n,
                              hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_float4_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       stream, src, stride, dst, (long long)n);
    return hipGetLastError();
}

__global__ void so3_selftest_kernel(const double *w, double *R, double *w2, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double wi[3] = {w[3 * i], w[3 * i + 1], w[3 * i + 2]}, Ri[9], wo[3];
    rodrigues(wi, Ri);
    invrodrigues(Ri, wo);
    for (int a = 0; a < 9; a++) R[9 * i + a] = Ri[a];
    for (int a = 0; a < 3; a++) w2[3 * i + a] = wo[a];
}

// the derivatives and the projection as well (core/rodrigues.h:143-237)
__global__ void so3_selftest_jac_kernel(const double *w, int n, double *R, double *dR, double *w2, double *dw,
                                        double *proj)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double wi[3] = {w[3 * i], w[3 * i + 1], w[3 * i + 2]}, Ri[9], D1[27], wo[3], D2[27], P[9], A[9];
    rodrigues_jac(wi, Ri, D1);
    invrodrigues_jac(Ri, wo, D2);
    // a matrix near the rotation (sheared a little), projected back
    for (int a = 0; a < 9; a++) A[a] = Ri[a] * (1.0 + 0.01 * (a % 3)) + 0.003 * a;
    project_so3(A, P);
    for (int a = 0; a < 9; a++) { R[9 * i + a] = Ri[a]; proj[9 * i + a] = P[a]; }
    for (int a = 0; a < 27; a++) { dR[27 * i + a] = D1[a]; dw[27 * i + a] = D2[a]; }
    for (int a = 0; a < 3; a++) w2[3 * i + a] = wo[a];
}

hipError_t launch_so3_selftest_jac(const double *w, int n, double *R, double *dR, double *w2, double *dw, double *proj,
                                   hipStream_t stream)
{
    hipLaunchKernelGGL(so3_selftest_jac_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, w, n, R, dR, w2, dw, proj);
    return hipGetLastError();
}

// SE(3) on the device (so3.h: se3_compose / se3_act / se3_inv, the restatement of core/se3.h:96-110):
// gh = g * h, gv = g(v), gi = g^-1 for n elements (g, h: row-major 3x4)
__global__ void se3_selftest_kernel(const double *g, const double *h, const double *v, int n, double *gh, double *gv,
                                    double *gi)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a[12], b[12], o[12], p[3], q[3];
    for (int k = 0; k < 12; k++) { a[k] = g[12 * i + k]; b[k] = h[12 * i + k]; }
    for (int k = 0; k < 3; k++) p[k] = v[3 * i + k];
    se3_compose(a, b, o);
    for (int k = 0; k < 12; k++) gh[12 * i + k] = o[k];
    se3_act(a, p, q);
    for (int k = 0; k < 3; k++) gv[3 * i + k] = q[k];
    se3_inv(a, o);
    for (int k = 0; k < 12; k++) gi[12 * i + k] = o[k];
}

hipError_t launch_se3_selftest(const double *g, const double *h, const double *v, int n, double *gh, double *gv,
                               double *gi, hipStream_t stream)
{
    hipLaunchKernelGGL(se3_selftest_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, g, h, v, n, gh, gv, gi);
    return hipGetLastError();
}

hipError_t launch_so3_selftest(const double *w, double *R, double *w2, int n,
                               hipStream_t stream)
{
    hipLaunchKernelGGL(so3_selftest_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, w, R, w2,
                       n);
    return hipGetLastError();
}

}  // namespace visma
