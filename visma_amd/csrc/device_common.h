// device_common.h -- device helpers shared by the NN and reduction kernels.
//
// Arithmetic contract (checked bit-for-bit against oracle/icp_oracle.c vk_*):
//   p  = fmaf(r0,sx, fmaf(r1,sy, fmaf(r2,sz, t)))          fp32, per row
//   d2 = fmaf(dz,dz, fmaf(dy,dy, dx*dx)),  d = q - p        fp32
//   accept iff d2 < r2f; lowest target index wins exact ties.
// Statistics are accumulated in f64 from p = T64 * (double)s and q widened.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels.h"
#include "so3.h"
#include "icp_state.h"

namespace visma {

__device__ __forceinline__ void xform_point_f32(const Xform32 &T, const float4 s,
                                                float &px, float &py, float &pz)
{
    px = __builtin_fmaf(T.m[0], s.x, __builtin_fmaf(T.m[1], s.y, __builtin_fmaf(T.m[2], s.z, T.m[3])));
    py = __builtin_fmaf(T.m[4], s.x, __builtin_fmaf(T.m[5], s.y, __builtin_fmaf(T.m[6], s.z, T.m[7])));
    pz = __builtin_fmaf(T.m[8], s.x, __builtin_fmaf(T.m[9], s.y, __builtin_fmaf(T.m[10], s.z, T.m[11])));
}

// Transform / radius of this launch: from the device-resident loop state when
// one is given (asynchronous on-device loop), else from the kernel arguments.
// Returns false when the loop has already finished (the launch is a no-op).
__device__ __forceinline__ bool load_loop_state(const DevIcpState *st, Xform32 &T32, Xform64 &T64,
                                                Offset64 &off, float &r2f)
{
    if (!st) return true;
    if (!st->active) return false;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        T64.m[i] = st->Tc[i];
        T32.m[i] = (float)st->Tc[i];
    }
#pragma unroll
    for (int a = 0; a < 3; a++) off.v[a] = st->world_frame ? st->centre[a] : 0.0;
    r2f = st->r2f;
    return true;
}

// The cell coordinate of the radius-cell grid; build and every query kernel MUST use this
// same expression.
__device__ __forceinline__ int cell_coord(float v, float mn, float inv_h, int dim)
{
    float u = floorf((v - mn) * inv_h);
    u = fminf(fmaxf(u, -2.0f), (float)dim + 1.0f);   // also tames inf / huge values
    return (int)u;
}

__device__ __forceinline__ float sqdist_f32(const float4 q, float px, float py, float pz)
{
    const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

// The exact search ranks candidates in fp32 and re-ranks in f64 what fp32 cannot decide.  With
// p32 = fl(p64) and q32 = fl(q64) the difference vector is off by at most u (|p| + |q|) per
// component (u = 2^-24) and the fp32 evaluation of d2 adds 3u relative, so
// |d64 - sqrt(d2_32)| <= u (2 |p| + 2.5 r); the band half-width E is more than twice that.
__device__ __forceinline__ float exact_band(float px, float py, float pz, float r2f, float *r_up_out = nullptr)
{
    const float r_up = sqrtf(r2f) * (1.0f + 2.4e-7f);
    if (r_up_out) *r_up_out = r_up;
    return 2.4e-7f * (fabsf(px) + fabsf(py) + fabsf(pz) + r_up) + 4.8e-7f * r_up;
}
// squared fp32 distance at or beyond which a candidate cannot be accepted in f64
__device__ __forceinline__ float exact_band_limit(float px, float py, float pz, float r2f)
{
    float r_up;
    const float E = exact_band(px, py, pz, r2f, &r_up);
    const float t = r_up + 2.0f * E;
    return t * t * (1.0f + 6e-7f);
}

// one VALU op: min of three (v_min3_f32); inputs are never NaN-producing here
__device__ __forceinline__ float min3_f32(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// ---- agent-scope accesses for data handed between workgroups inside one launch ------------
__device__ __forceinline__ void store_agent_f64(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p),
                       (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_agent_f64(const double *p)
{
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p),
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __longlong_as_double((long long)v);
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Accumulator layouts
//  point-to-point (23): 0 K | 1 r2 | 2-4 sum p | 5-7 sum q |
//                       8-13 sum pp^T (xx xy xz yy yz zz) | 14-22 sum q p^T
//  point-to-plane (29): 0 K | 1 r2 | 2-22 upper J^T J | 23-28 J^T r
template <bool PLANE>
struct Acc {
    static constexpr int N = PLANE ? 29 : 23;
};

// Jacobian / residual rows of ONE correspondence, in f64 (design rules R1/R2):
// p = T64 * s (+ frame offset), q = target point (+ frame offset).
// (sx,sy,sz) = source point, (qx,qy,qz) = its target point, both in the centred frame
// (tx,ty,tz) = the TRANSFORMED source point T64 * s as the search computed it (same expression,
// same order: the statistics do not depend on which entry point formed p)
template <bool PLANE>
__device__ __forceinline__ void accumulate_pq_d(double *acc, const double tx, const double ty, const double tz,
                                                const double qx, const double qy, const double qz,
                                                const double nx, const double ny, const double nz,
                                                const Offset64 &off)
{
    const double p[3] = {tx + off.v[0], ty + off.v[1], tz + off.v[2]};
    const double q[3] = {qx + off.v[0], qy + off.v[1], qz + off.v[2]};
    const double r[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
    acc[0] += 1.0;
    if (!PLANE) {
        // rows J_k = [p x e_k | e_k] = k-th row of [-hat(p) | I]: only their
        // moments are kept here; finalize_kernel expands them to J^T J / J^T r
        acc[1] += r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
        acc[2] += p[0]; acc[3] += p[1]; acc[4] += p[2];
        acc[5] += q[0]; acc[6] += q[1]; acc[7] += q[2];
        acc[8] += p[0] * p[0]; acc[9] += p[0] * p[1]; acc[10] += p[0] * p[2];
        acc[11] += p[1] * p[1]; acc[12] += p[1] * p[2]; acc[13] += p[2] * p[2];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) acc[14 + a * 3 + b] += q[a] * p[b];
    } else {
        const double n[3] = {nx, ny, nz};
        const double rr = r[0] * n[0] + r[1] * n[1] + r[2] * n[2];
        // J = [p x n | n]  (TransformationEstimation.cpp:87-89); p x n = hat(p) n
        double J[6], H[9];
        hat(p, H);
        J[0] = H[0] * n[0] + H[1] * n[1] + H[2] * n[2];
        J[1] = H[3] * n[0] + H[4] * n[1] + H[5] * n[2];
        J[2] = H[6] * n[0] + H[7] * n[1] + H[8] * n[2];
        J[3] = n[0]; J[4] = n[1]; J[5] = n[2];
        // the result's inlier_rmse is the NEAREST-NEIGHBOUR distance whatever the estimator
        // (Registration.cpp:65-68,93: error2 += dists[0]), not the plane residual
        acc[1] += r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
        int o = 2;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = a; b < 6; b++) acc[o++] += J[a] * J[b];
#pragma unroll
        for (int a = 0; a < 6; a++) acc[23 + a] += J[a] * rr;
    }
}

template <bool PLANE>
__device__ __forceinline__ void accumulate_pair_d(double *acc, const double sx, const double sy, const double sz,
                                                  const double qx, const double qy, const double qz,
                                                  const double nx, const double ny, const double nz,
                                                  const Xform64 &T64, const Offset64 &off)
{
    accumulate_pq_d<PLANE>(acc, T64.m[0] * sx + T64.m[1] * sy + T64.m[2] * sz + T64.m[3],
                           T64.m[4] * sx + T64.m[5] * sy + T64.m[6] * sz + T64.m[7],
                           T64.m[8] * sx + T64.m[9] * sy + T64.m[10] * sz + T64.m[11], qx, qy, qz, nx, ny, nz, off);
}

template <bool PLANE>
__device__ __forceinline__ void accumulate_pair(double *acc, const float4 s4, const float4 q4,
                                                const float4 n4, const Xform64 &T64,
                                                const Offset64 &off)
{
    accumulate_pair_d<PLANE>(acc, (double)s4.x, (double)s4.y, (double)s4.z, (double)q4.x, (double)q4.y,
                             (double)q4.z, (double)n4.x, (double)n4.y, (double)n4.z, T64, off);
}

// Wave-wide sums of N values per lane with ~N shuffles instead of 6*N: at every
// butterfly step a lane hands HALF of its values to its partner and keeps the other
// half (the partner does the opposite), so the value count halves while the lane
// distance halves.  After the six steps lane L holds the total of ONE value,
// index multi_index<N>(L) (or -1: that lane ended up with padding).
template <int N>
__device__ __forceinline__ int multi_index(int lane)
{
    int n[7];
    n[0] = N;
#pragma unroll
    for (int s = 0; s < 6; s++) n[s + 1] = (n[s] + 1) / 2;
    int pos = 0;
    bool ok = true;
#pragma unroll
    for (int s = 5; s >= 0; s--) {                       // undo the steps, last one first
        if (lane & (32 >> s)) pos += n[s + 1];
        ok = ok && pos < n[s];
    }
    return ok ? pos : -1;
}

template <int N>
__device__ __forceinline__ double wave_sum_multi(double *v)
{
    const int lane = threadIdx.x & 63;
    int n = N;
#pragma unroll
    for (int s = 0; s < 6; s++) {
        const int m = 32 >> s, half = (n + 1) / 2;
        const bool up = (lane & m) != 0;
#pragma unroll
        for (int i = 0; i < half; i++) {
            const double lower = v[i], upper = (i + half < n) ? v[i + half] : 0.0;
            const double send = up ? lower : upper, keep = up ? upper : lower;
            v[i] = keep + __shfl_xor(send, m, 64);
        }
        n = half;
    }
    return v[0];
}

// wavefront shuffle reduction -> LDS across the 4 waves -> one partial row
// (row = nullptr: partials + blockIdx.x * kReduceAcc; agent: write-through stores for the fused fold)
// the thread's number in its workgroup; OPAQUE: as a value the compiler cannot see through (inside the loop of the
// persistent kernel everything derived from threadIdx.x alone was hoisted out of the loop and spilled)
template <bool OPAQUE, int NTH = kBlock>
__device__ __forceinline__ int thread_number()
{
    int t = threadIdx.x;
    if constexpr (OPAQUE) {
        asm volatile("" : "+v"(t));
        t &= NTH - 1;
    }
    return t;
}

typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
// a double as a granule {low half, tag, high half, tag}: each 8-byte half validates itself
__device__ __forceinline__ void store_granule_agent(u4_t *dst, double v, unsigned tag)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    u4_t g;
    g.x = (unsigned)b; g.y = tag; g.z = (unsigned)(b >> 32); g.w = tag;
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst), "v"(g) : "memory");
}

template <int NACC, int NWAVES = kBlock / 64, bool OPAQUE_TID = false>
__device__ __forceinline__ void block_reduce_store(double *acc, double *partials, bool agent = false, void *tagged_row = nullptr,
                                                   unsigned tag = 0u)
{
    __shared__ double wsum[NWAVES][NACC];
    const int tidx = thread_number<OPAQUE_TID, NWAVES * 64>();
    const int lane = tidx & 63, wave = tidx >> 6;
    const double tot = wave_sum_multi<NACC>(acc);
    const int slot = multi_index<NACC>(lane);
    if (slot >= 0) wsum[wave][slot] = tot;
    __syncthreads();
    if (tidx < NACC) {
        double v = wsum[0][tidx];
#pragma unroll
        for (int w = 1; w < NWAVES; w++) v += wsum[w][tidx];
        if (tagged_row) {
            // (persistent launches, polled fold: the row as self-validating granules, written through -- nobody waits for
            //  an acknowledgement, the group's reducer polls for the tag)
            store_granule_agent(reinterpret_cast<u4_t *>(tagged_row) + (long long)blockIdx.x * 32 + tidx, v, tag);
        } else {
            double *dst = partials + (long long)blockIdx.x * kReduceAcc + tidx;
            if (agent) store_agent_f64(dst, v);
            else *dst = v;
        }
    }
}

// The 38 statistics from the folded accumulator totals: point-to-plane totals are the
// statistics; the compact point-to-point moments are expanded into the literal 6x6 / 6x1
// normal equations (J^T J = [[sum(|p|^2 I - p p^T), hat(sum p)], [., K I]],
// J^T r = [-vee(sum q p^T); sum p - sum q]).
template <bool PLANE>
__device__ __forceinline__ void expand_moments(const double *tot, double *stats)
{
    if (PLANE) {
        for (int i = 0; i < 29; i++) stats[i] = tot[i];
        for (int i = 29; i < kNStats; i++) stats[i] = 0.0;
    } else {
        const double K = tot[0];
        const double Px = tot[2], Py = tot[3], Pz = tot[4];
        const double Qx = tot[5], Qy = tot[6], Qz = tot[7];
        const double Sxx = tot[8], Sxy = tot[9], Sxz = tot[10];
        const double Syy = tot[11], Syz = tot[12], Szz = tot[13];
        const double *M = &tot[14];
        stats[0] = K;
        stats[1] = tot[1];
        double *J = stats + 2;  // upper triangle, row by row
        // row 0: sum(|p|^2 I - p p^T) | hat(sum p)
        J[0] = Syy + Szz; J[1] = -Sxy; J[2] = -Sxz; J[3] = 0.0; J[4] = -Pz; J[5] = Py;
        J[6] = Sxx + Szz; J[7] = -Syz; J[8] = Pz; J[9] = 0.0; J[10] = -Px;
        J[11] = Sxx + Syy; J[12] = -Py; J[13] = Px; J[14] = 0.0;
        J[15] = K; J[16] = 0.0; J[17] = 0.0;
        J[18] = K; J[19] = 0.0;
        J[20] = K;
        double *r = stats + 23;  // J^T r = [ -vee(sum q p^T) ; sum p - sum q ]
        double v3[3];
        vee(M, v3);
        r[0] = -v3[0]; r[1] = -v3[1]; r[2] = -v3[2];
        r[3] = Px - Qx; r[4] = Py - Qy; r[5] = Pz - Qz;
        for (int i = 0; i < 9; i++) stats[29 + i] = M[i];
    }
}

// One workgroup; folds `nblocks` partial rows in a fixed order and expands the
// compact point-to-point moments into the 6x6 / 6x1 normal equations.
// NT = threads of the workgroup (a multiple of 32, at most 1024): 32 statistics x NT/32 row groups.
template <bool PLANE, int NT = 1024>
__device__ __forceinline__ void fold_partials(const double *__restrict__ partials, int nblocks,
                                              double *__restrict__ stats)
{
    constexpr int NACC = Acc<PLANE>::N;
    constexpr int NG = NT / 32;            // row groups (1024 threads = 32 stats x 32 groups)
    __shared__ double part[NG][33];
    __shared__ double tot[32];
    const int a = threadIdx.x & 31, g = threadIdx.x >> 5;
    double v = 0.0;
    if (a < NACC) {
        // rows g, g+32, g+64, ...: eight independent load/add chains in flight
        double c[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int b = g;
        for (; b + 7 * NG < nblocks; b += 8 * NG) {
#pragma unroll
            for (int u = 0; u < 8; u++) c[u] += partials[(long long)(b + u * NG) * kReduceAcc + a];
        }
        for (; b < nblocks; b += NG) c[0] += partials[(long long)b * kReduceAcc + a];
        v = ((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + (c[6] + c[7]));
    }
    part[g][a] = v;
    __syncthreads();
    if (threadIdx.x < 32) {
        double t = 0.0;
#pragma unroll
        for (int gg = 0; gg < NG; gg++) t += part[gg][threadIdx.x];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) expand_moments<PLANE>(tot, stats);
}


// Publish the 38 statistics to mapped (fine-grained, uncached) host memory as
// self-validating 16-byte granules {value, sequence tag}: each granule is ONE
// global_store_dwordx4, so the host can accept a value as soon as its tag shows
// the expected sequence number -- no system-scope fence (whose L2 write-back of
// the launch's ~2 MB of dirty index output cost ~10 us per iteration).
__device__ __forceinline__ void publish_tagged_stats(const double *stats, double *host_out,
                                                     unsigned long long seq)
{
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    __syncthreads();                                   // stats[] was written by thread 0
    if (threadIdx.x < kNStats) {
        const unsigned long long v = (unsigned long long)__double_as_longlong(stats[threadIdx.x]);
        u4 g;
        g.x = (unsigned)v; g.y = (unsigned)(v >> 32);
        g.z = (unsigned)seq; g.w = (unsigned)(seq >> 32);
        __builtin_nontemporal_store(g, reinterpret_cast<u4 *>(host_out) + threadIdx.x);
    }
}

// One rank's part of the one-shot all-reduce over xGMI (see ipc_allreduce_kernel in kernels.hip): lane a
// of ONE wave stores statistic a as a 16-byte granule (value + call-count tags) into slot [rank][a] of every
// peer's mailbox, then waits for the nranks granules of statistic a in its OWN mailbox and sums them in
// rank order.  The mailbox has two halves used alternately (call count parity): a rank can only start
// call k+2 after every peer has SENT call k+1, which a peer does only after it has finished reading call
// k, so a granule is never overwritten before its reader has seen it.
__device__ __forceinline__ u4_t load_granule_sys(const u4_t *p)
{
    u4_t v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// TABLE: the mailboxes come from `table` (device memory) instead of `peers` (a by-value argument)
template <bool TABLE = false>
__device__ __forceinline__ double ipc_exchange(int a, double mine, const IpcPeers &peers, int rank, int nranks,
                                               unsigned long long seq, int *timeout_flag, long long max_spins,
                                               bool &late, void *const *table = nullptr)
{
    auto box = [&](int r) -> void * {
        if constexpr (TABLE) return table[r];
        else return peers.box[r];
    };
    // granule = {value bits 0..31, tag, value bits 32..63, tag} with tag = the low 32 bits of the call count:
    // each 8-byte half validates itself, so the protocol only needs 8-byte stores to arrive whole (the
    // 16-byte store may be split in two on its way through the fabric)
    double sum = 0.0;
    if (a < kNStats) {
        const unsigned long long v = (unsigned long long)__double_as_longlong(mine);
        const unsigned tag = (unsigned)seq;
        u4_t g;
        g.x = (unsigned)v; g.y = tag;
        g.z = (unsigned)(v >> 32); g.w = tag;
        const size_t half = (size_t)(seq & 1ull) * kIpcMaxRanks * kNStats;
        for (int p = 0; p < nranks; p++)
        {
            u4_t *dst = reinterpret_cast<u4_t *>(box(p)) + half + rank * kNStats + a;
            if constexpr (TABLE) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(dst), "v"(g) : "memory");   // (the launch goes on: written through now)
            else __builtin_nontemporal_store(g, dst);
        }
        const u4_t *own = reinterpret_cast<const u4_t *>(box(rank)) + half;
        for (int r = 0; r < nranks; r++) {
            u4_t w;
            long long spins = 0;
            for (;;) {
                w = load_granule_sys(own + r * kNStats + a);
                if (w.y == tag && w.w == tag) break;
                if (++spins > max_spins) { *timeout_flag = 1 + r; late = true; break; }   // a peer never arrived
                __builtin_amdgcn_s_sleep(2);
            }
            sum += __longlong_as_double((long long)(((unsigned long long)w.z << 32) | w.x));
        }
    }
    return sum;
}

// Fused fold.  Every workgroup of a problem has written its partial row (kReduceAcc doubles,
// agent-scope stores) at partials[(row0 + lb) * kReduceAcc]; this folds them to the 38
// statistics INSIDE the launch: the last workgroup to arrive in each group of kFoldGroup rows
// sums that group (fixed order) into a level-2 row, the last group folder sums the level-2
// rows (fixed order), expands the moments and publishes.  Sums do not depend on arrival order:
// repeat runs are bit-identical.  Hand-off per the agent-scope recipe: write-through stores,
// every storing wave drains, one relaxed agent atomic as the ticket, one acquire by the reader.
// tickets: ticket_stride words per problem, zero before the first launch (the last arrivers
// re-arm them); [0] = level 2, [1 + g] = level 1 of group g.
// (kFoldGroup: kernels.h -- the host sizes the ticket and level-2 buffers by it)
// (128 since the end of round 4, 256 before: 65,536 queries = 256 rows fold 0.8 us faster in two levels of 16 x 16 than
//  with four batches of loads at one level -- 17.3 vs 18.1 us per iteration; 64: 128- and 79-row launches lose 0.5-1.2 us)
#ifndef VISMA_FOLD_SINGLE
#define VISMA_FOLD_SINGLE 128
#endif
constexpr int kFoldSingle = VISMA_FOLD_SINGLE;      // up to this many rows: one level
// Returns true on the ONE workgroup of the problem that finished the fold and published the statistics.
// LOOPED: called from the loop of the persistent kernel (no exchange with peers there; opaque thread number).
// SOLVE: the kernel honours FoldArgs::solve -- the publishing workgroup advances the problem's device-resident state
// (closed-form update, compose, stop test) right behind its fold, one thread on a copy of the state in LDS.
// (the persistent sweep launch: the one-thread closed-form update OUT of line -- inlined it spilled 150 registers of the
//  128-register search kernel around it; out of line it has an allocation of its own and the search path none of its cost)
__device__ __attribute__((noinline)) void advance_state_kabsch_outlined(DevIcpState *s) { advance_state<true>(s); }

template <bool PLANE, int NTH, bool LOOPED = false, bool SOLVE = false>
__device__ __forceinline__ bool fused_fold(const FoldArgs &f, const double *partials, long long row0, int lb,
                                           int bpp, int prob)
{
    constexpr int NACC = Acc<PLANE>::N;
    constexpr int NG = NTH / 32;
    __shared__ double f_part[NG][33];
    __shared__ double f_tot[32];
    __shared__ int f_flag[2];
    const int tid = thread_number<LOOPED, NTH>();
    // few rows (small clouds, the problems of a batch): ONE level -- the last workgroup sums them all;
    // otherwise groups of kFoldGroup rows, then the group sums
    const bool single = bpp <= kFoldSingle;
    const int ngroups = single ? 1 : (bpp + kFoldGroup - 1) / kFoldGroup;
    unsigned *tk = f.tickets + (long long)prob * f.ticket_stride;
    const int grp = single ? 0 : lb / kFoldGroup;
    const int gsize = single ? bpp : min(kFoldGroup, bpp - grp * kFoldGroup);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every storing wave drains
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(tk + 1 + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f_flag[0] = (t == (unsigned)(gsize - 1)) ? 1 : 0;
        if (f_flag[0]) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(tk + 1 + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!f_flag[0]) return false;
    const int sa = tid & 31, sg = tid >> 5;
    {
        const double *grows = partials + (row0 + (long long)grp * kFoldGroup) * kReduceAcc;
        double v = 0.0;
        if (sa < NACC)
            // rows sg, sg + NG, ...: eight loads in flight, added in row order
            for (int r0 = sg; r0 < gsize; r0 += 8 * NG) {
                double w[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int r = r0 + u * NG;
                    w[u] = r < gsize ? load_agent_f64(grows + (long long)r * kReduceAcc + sa) : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; u++) v += w[u];
            }
        f_part[sg][sa] = v;
    }
    __syncthreads();
    double *rows2 = f.partials2 + ((long long)prob * f.ticket_stride + grp) * kReduceAcc;
    if (tid < 32) {
        double t = 0.0;
#pragma unroll
        for (int gg = 0; gg < NG; gg++) t += f_part[gg][tid];
        if (single) f_tot[tid] = t;
        else if (tid < NACC) store_agent_f64(rows2 + tid, t);
    }
    if (!single) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f_flag[1] = (t == (unsigned)(ngroups - 1)) ? 1 : 0;
        if (f_flag[1]) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!f_flag[1]) return false;
    {
        const double *g2 = f.partials2 + (long long)prob * f.ticket_stride * kReduceAcc;
        double v = 0.0;
        if (sa < NACC)
            // rows sg, sg + NG, ...: all loads of a thread in flight (a dependent round trip per row cost 2 us at
            // 1024 workgroups), added in row order
            for (int r0 = sg; r0 < ngroups; r0 += 8 * NG) {
                double w[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int r = r0 + u * NG;
                    w[u] = r < ngroups ? load_agent_f64(g2 + (long long)r * kReduceAcc + sa) : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; u++) v += w[u];
            }
        f_part[sg][sa] = v;
    }
    __syncthreads();
    if (tid < 32) {
        double t = 0.0;
#pragma unroll
        for (int gg = 0; gg < NG; gg++) t += f_part[gg][tid];
        f_tot[tid] = t;
    }
    }
    __syncthreads();
    // the statistics are expanded into LDS and go out from there (device copy, exchange, host granules): reading
    // them back from global memory put a store -> load round trip into the tail of every launch
    double *stats = f.stats_out + (long long)prob * f.stats_stride;
    __shared__ double f_stats[kNStats + 2];
    if (tid == 0) expand_moments<PLANE>(f_tot, f_stats);
    __syncthreads();
    if (f.ipc_n > 1) {
        // source-sharded ranks: this workgroup's first wave exchanges the statistics with the peers
        // (remote stores over xGMI, rank-ordered sum) before anything is published
        // the number of THIS exchange: kept in device memory, advanced by the one workgroup that exchanges
        __shared__ unsigned long long f_seq;
        if (tid == 0) {
            f_flag[0] = 0;
            f_seq = __hip_atomic_load(f.ipc_seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
            __hip_atomic_store(f.ipc_seq_dev, f_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (tid < 64) {
            bool late = false;
            // (inside the persistent kernel the mailboxes are read from the table in device memory)
            const double sum = ipc_exchange<LOOPED>(tid, tid < kNStats ? f_stats[tid] : 0.0, f.peers, f.ipc_rank, f.ipc_n,
                                                    f_seq, f.ipc_flag, f.ipc_spins, late, f.peer_table);
            if (tid < kNStats) f_stats[tid] = late ? __longlong_as_double(0x7ff8000000000000ll) : sum;
            if (late) f_flag[0] = 1;
        }
        __syncthreads();
        if (f_flag[0]) {                                   // a peer was lost: nothing is published
            if (tid < kNStats) stats[tid] = f_stats[tid];
            return true;
        }
    }
    if (tid < kNStats) {
        const double sv = f_stats[tid];
        stats[tid] = sv;
        if (f.host_out) {
            const unsigned long long v = (unsigned long long)__double_as_longlong(sv);
            u4_t g;
            g.x = (unsigned)v; g.y = (unsigned)(v >> 32);
            g.z = (unsigned)f.seq; g.w = (unsigned)(f.seq >> 32);
            u4_t *dst = reinterpret_cast<u4_t *>(f.host_out) + tid;
            if constexpr (LOOPED) {
                // the launch goes on after this: a system-scope store (written through now -- a plain or non-temporal
                // one may stay in the L2 until the launch ends, and the host would wait for exactly that)
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(dst), "v"(g) : "memory");
            } else {
                __builtin_nontemporal_store(g, dst);
            }
        }
    }
    if constexpr (SOLVE) {
        if (f.solve && f.ipc_n <= 1) {                       // (workgroup-uniform)
            // Every workgroup of this problem has delivered its row -- each read the state when it began --, so the state
            // may move on now: the whole workgroup brings it into LDS (one round trip instead of one per field), the
            // statistics come from where the fold left them, thread 0 solves, the workgroup writes it back.  What
            // solve_state_kernel did in a launch of its own between every two search launches (18 % of config 3's GPU time).
            static_assert(sizeof(DevIcpState) % 8 == 0 && offsetof(DevIcpState, stats) % 8 == 0, "state copied as 8-byte words");
            constexpr int kWords = (int)(sizeof(DevIcpState) / 8), kStatsWord = (int)(offsetof(DevIcpState, stats) / 8);
            __shared__ unsigned long long f_sst[kWords];
            unsigned long long *gs = reinterpret_cast<unsigned long long *>(f.solve + prob);
            if constexpr (LOOPED) {
                // (the persistent sweep launch: the workgroup that advanced this state in the pass before may sit on
                //  another XCD -- the words come from and go to memory, past the L2s)
                // (f.sweep_passes: the pass counter this state must show -- what the pass before left; a state still on its
                //  way from that pass's folding workgroup shows an older one: read again.  The words of one state are stored
                //  by one workgroup in one go; the counter lies behind the transform and the bookkeeping in the struct, and a
                //  torn read -- new counter, old words before it -- is excluded by reading the counter LAST)
                constexpr int kPassWord = (int)(offsetof(DevIcpState, passes) / 8);
                static_assert(offsetof(DevIcpState, passes) % 8 == 4 || offsetof(DevIcpState, passes) % 8 == 0, "passes inside one word");
                for (int guard = 0; guard < (1 << 20); guard++) {
                    for (int k = tid; k < kWords; k += NTH)
                        if (k != kPassWord)
                            f_sst[k] = (k >= kStatsWord && k < kStatsWord + kNStats) ? (unsigned long long)__double_as_longlong(f_stats[k - kStatsWord])
                                                                                     : __hip_atomic_load(gs + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (tid == 0) f_sst[kPassWord] = __hip_atomic_load(gs + kPassWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __syncthreads();
                    if (reinterpret_cast<const DevIcpState *>(f_sst)->passes == f.sweep_passes) break;
                    __syncthreads();
                }
            } else {
                for (int k = tid; k < kWords; k += NTH)
                    f_sst[k] = (k >= kStatsWord && k < kStatsWord + kNStats) ? (unsigned long long)__double_as_longlong(f_stats[k - kStatsWord])
                                                                             : gs[k];
            }
            __syncthreads();
            if constexpr (LOOPED) { if (tid == 0) advance_state_kabsch_outlined(reinterpret_cast<DevIcpState *>(f_sst)); }
            else { if (tid == 0) advance_state<true>(reinterpret_cast<DevIcpState *>(f_sst)); }
            __syncthreads();
            if constexpr (LOOPED) {
                // the problem's other workgroups first: they wait for nothing but these 25 words.  The state follows; the
                // workgroup that folds the NEXT pass (a pass later, possibly on another XCD) validates what it reads by the
                // pass counter it expects and reads again until it sees it (above).
                if (f.sweep_relay && tid < kPersistWords) {
                    const DevIcpState *ns_ = reinterpret_cast<const DevIcpState *>(f_sst);
                    unsigned long long w;
                    if (tid < 24) {
                        const unsigned long long b = (unsigned long long)__double_as_longlong(ns_->Tc[tid >> 1]);
                        w = (tid & 1) ? (b >> 32) : (b & 0xFFFFFFFFull);
                    } else {
                        w = ns_->active ? kPersistGo : kPersistStop;
                    }
                    __hip_atomic_store(f.sweep_relay + 32ll * prob + tid, w | ((unsigned long long)f.sweep_tag << 32), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
                {
                    // (the word with the pass counter goes LAST, when the others are in memory: whoever reads it reads them)
                    constexpr int kPassWordW = (int)(offsetof(DevIcpState, passes) / 8);
                    for (int k = tid; k < kWords; k += NTH)
                        if (k != kPassWordW) __hip_atomic_store(gs + k, f_sst[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (tid == 0) __hip_atomic_store(gs + kPassWordW, f_sst[kPassWordW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                for (int k = tid; k < kWords; k += NTH) gs[k] = f_sst[k];
            }
        }
    }
    return true;
}

// ---- The POLLED fold of a persistent launch.  Every workgroup of the launch is resident, so nobody needs a ticket to
// learn that it is the last: the rows are granules that validate themselves with the pass's tag, the FIRST workgroup of
// every group of kFoldGroup rows polls its group's rows and writes the group row, workgroup 0 polls the group rows, sums,
// expands and publishes.  Behind the slowest workgroup that is two store-to-poll latencies instead of two acknowledged
// stores, two ticket round trips and two loads.  Sums in the order of fused_fold (rows sg, sg + NG, ... per thread, then
// the thread groups in order): the statistics are bit-identical to its.  A reducer whose rows do not come (the launch
// is dead: somebody left) gives up with everybody else.  Returns true on the workgroup that published.
template <bool PLANE, int NTH>
__device__ __forceinline__ bool polled_fold(const FoldArgs &f, int lb, int bpp, unsigned tag)
{
    constexpr int NACC = Acc<PLANE>::N;
    constexpr int NG = NTH / 32;
    __shared__ double p_part[NG][33];
    __shared__ double p_tot[32];
    __shared__ int p_dead;
    const int tid = thread_number<true, NTH>();
    const bool single = bpp <= kFoldSingle;
    const int ngroups = single ? 1 : (bpp + kFoldGroup - 1) / kFoldGroup;
    // Who reduces group g: workgroup g -- the lowest-numbered workgroups are the OLDEST waves of their SIMDs (blocks 0..255
    // are the first resident workgroup of every compute unit), which the instruction arbiter serves first: their bodies are
    // done 3 us before the youngest workgroups' (tools/persist_timeline.py, round 5: four plateaus by residency slot), and
    // their polls are not starved by three older waves.  (Until round 5: the first workgroup of each group, slots 0..3 alike.)
#ifndef VISMA_FOLD_REDUCERS_FIRST
#define VISMA_FOLD_REDUCERS_FIRST 1
#endif
    int grp;
    if (VISMA_FOLD_REDUCERS_FIRST) {
        if (lb >= ngroups) return false;                     // (not a reducer: the row is out, done)
        grp = lb;
    } else {
        grp = single ? 0 : lb / kFoldGroup;
        if (lb != grp * kFoldGroup) return false;
    }
    const int gsize = single ? bpp : min(kFoldGroup, bpp - grp * kFoldGroup);
    const int sa = tid & 31, sg = tid >> 5;
    if (tid == 0) p_dead = 0;
    __syncthreads();
    const long long t0 = (long long)wall_clock64();
    // the sum of `count` granule rows starting at `rows` (32 granules per row), rows sg, sg + NG, ... on this thread
    auto poll_sum = [&](const u4_t *rows, int count) {
        double v = 0.0;
        if (sa < NACC)
            for (int r0 = sg; r0 < count; r0 += 8 * NG) {
                double w[8];
                unsigned pending = 0u;
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    w[u] = 0.0;
                    if (r0 + u * NG < count) pending |= 1u << u;
                }
                while (pending) {
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (pending & (1u << u)) {
                            const u4_t g = load_granule_sys(rows + (long long)(r0 + u * NG) * 32 + sa);
                            if (g.y == tag && g.w == tag) {
                                w[u] = __longlong_as_double((long long)(((unsigned long long)g.z << 32) | g.x));
                                pending &= ~(1u << u);
                            }
                        }
                    if (pending) {
                        const bool gone = __hip_atomic_load(f.dead_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull ||
                                          (long long)wall_clock64() - t0 > f.poll_ticks;
                        if (gone) { p_dead = 1; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; u++) v += w[u];
            }
        return v;
    };
    p_part[sg][sa] = poll_sum(reinterpret_cast<const u4_t *>(f.rows_tagged) + (long long)grp * kFoldGroup * 32, gsize);
    __syncthreads();
    if (p_dead) {
        if (tid == 0) __hip_atomic_store(f.dead_flag, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
    }
    if (tid < 32) {
        double t = 0.0;
#pragma unroll
        for (int gg = 0; gg < NG; gg++) t += p_part[gg][tid];
        if (single) p_tot[tid] = t;
        else if (tid < NACC) store_granule_agent(reinterpret_cast<u4_t *>(f.rows2_tagged) + (long long)grp * 32 + tid, t, tag);
    }
    if (!single) {
        if (lb != 0) return false;
        __syncthreads();
        p_part[sg][sa] = poll_sum(reinterpret_cast<const u4_t *>(f.rows2_tagged), ngroups);
        __syncthreads();
        if (p_dead) {
            if (tid == 0) __hip_atomic_store(f.dead_flag, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        if (tid < 32) {
            double t = 0.0;
#pragma unroll
            for (int gg = 0; gg < NG; gg++) t += p_part[gg][tid];
            p_tot[tid] = t;
        }
    }
    __syncthreads();
    // the statistics: expanded, exchanged with the peers (source-sharded ranks), published -- as fused_fold does
    double *stats = f.stats_out;
    __shared__ double p_stats[kNStats + 2];
    __shared__ unsigned long long p_seq;
    if (tid == 0) expand_moments<PLANE>(p_tot, p_stats);
    __syncthreads();
    if (f.ipc_n > 1) {
        if (tid == 0) {
            p_dead = 0;
            p_seq = __hip_atomic_load(f.ipc_seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
            __hip_atomic_store(f.ipc_seq_dev, p_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (tid < 64) {
            bool late = false;
            const double sum = ipc_exchange<true>(tid, tid < kNStats ? p_stats[tid] : 0.0, f.peers, f.ipc_rank, f.ipc_n, p_seq,
                                                  f.ipc_flag, f.ipc_spins, late, f.peer_table);
            if (tid < kNStats) p_stats[tid] = late ? __longlong_as_double(0x7ff8000000000000ll) : sum;
            if (late) p_dead = 1;
        }
        __syncthreads();
        if (p_dead) {                                      // a peer was lost: nothing is published
            if (tid < kNStats) stats[tid] = p_stats[tid];
            return true;
        }
    }
    if (tid < kNStats) {
        const double sv = p_stats[tid];
        stats[tid] = sv;
        if (f.host_out) {
            const unsigned long long v = (unsigned long long)__double_as_longlong(sv);
            u4_t g;
            g.x = (unsigned)v; g.y = (unsigned)(v >> 32);
            g.z = (unsigned)f.seq; g.w = (unsigned)(f.seq >> 32);
            u4_t *dst = reinterpret_cast<u4_t *>(f.host_out) + tid;
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(dst), "v"(g) : "memory");
        }
    }
    return true;
}

}  // namespace visma
