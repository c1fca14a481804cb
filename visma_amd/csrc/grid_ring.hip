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
// in RINGS around its own row, nearest ring first, each row only over the x-extent that can still hold a candidate nearer
// than the best so far, and stops when the next ring lies beyond the best.  The best so far starts at the radius -- or, for
// every pass after the first, at the distance to the previous pass's winner under the new transform (the state of the
// cooperative kernels, same buffer), so a converging registration visits the few cells around its winner.  The cost of a
// query follows the number of target points nearer than its nearest neighbour's distance, not the radius.
//
// Exactness: candidates are the caller's f64 points (32 B each), the source is transformed in f64, d2 is the reference's
// f64 sum of squares in x, y, z order (flann dist.h:159-176), acceptance is d2 < (double)(float)(r*r)
// (KDTreeFlann.cpp:184-185), ties go to the lowest original index -- the arithmetic of nn_grid_reduce_kernel<F64>.  Rows
// and cells are pruned with the fp32 view and that kernel's margins (1e-3 cell on every slab distance, 1e-5 relative on
// the squared bound, the bound rounded up), strictly: a pruned candidate is strictly farther than the best, so it could
// neither win nor tie.  Eight lanes work on one query: each looks up one row of a ring, the octet then scans the eight
// x-ranges side by side, eight candidates per step; the partial minima meet in a butterfly on (d2, index).
// Statistics, outputs, fold: as nn_grid_reduce_kernel (same accumulators, block_reduce_store, fused_fold).
#include "device_common.h"

namespace visma {

namespace {

constexpr int kRingG = 8;                       // lanes per query
constexpr int kRingU = 2;                       // 32-byte candidates in flight per lane
constexpr unsigned kRingNone = 0xFFFFFFFFu;     // no winner (the largest index: loses every tie)
constexpr unsigned kRingState = 0xFFFFFFFEu;    // the winner is the point the state holds (no slot known)

template <bool PLANE>
__global__ __launch_bounds__(kBlock) void nn_ring_kernel(
    int ns, const Pt64 *__restrict__ src64, const Pt64 *__restrict__ sorted64, const unsigned *__restrict__ start,
    const GridParams g, const float4 *__restrict__ nrm, const Pt64 *__restrict__ nrm64, Xform64 T64, Offset64 off, float r2f,
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
    double acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++) acc[a] = 0.0;
    unsigned long long ncand = 0ull, nrows_seen = 0ull;

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

    int it = 0;
    for (long long i = i_begin; i < i_end; i++, it++) {
        // ---- the query: the reference's transform of a source point (PointCloud.cpp:75-80), in f64
        const Pt64 s8 = src64[i];
        const double pxd = T64.m[0] * s8.x + T64.m[1] * s8.y + T64.m[2] * s8.z + T64.m[3] * 1.0;
        const double pyd = T64.m[4] * s8.x + T64.m[5] * s8.y + T64.m[6] * s8.z + T64.m[7] * 1.0;
        const double pzd = T64.m[8] * s8.x + T64.m[9] * s8.y + T64.m[10] * s8.z + T64.m[11] * 1.0;
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

        // the best so far as fp32, rounded UP a little, the smallest of the octet
        auto octet_best = [&]() {
            float f = (float)bd;
            f += f * 1e-6f;
#pragma unroll
            for (int m = kRingG >> 1; m > 0; m >>= 1) f = fminf(f, __shfl_xor(f, m, 64));
            return f;
        };
        const float ux = (px - g.mn[0]) * g.inv_h, uy = (py - g.mn[1]) * g.inv_h, uz = (pz - g.mn[2]) * g.inv_h;
        // (written so that a NaN query is outside)
        const bool inside = ux >= -reach && ux <= (float)g.dim[0] + reach && uy >= -reach && uy <= (float)g.dim[1] + reach &&
                            uz >= -reach && uz <= (float)g.dim[2] + reach;
        if (inside) {
            const float flx = floorf(ux), fly = floorf(uy), flz = floorf(uz);
            const int cx = (int)flx, cy = (int)fly, cz = (int)flz;
            const float fx = ux - flx, fy = uy - fly, fz = uz - flz;      // position inside the cell, [0, 1)
            const float lo_x = fmaxf(fx - mgn, 0.f), hi_x = fmaxf(1.0f - fx - mgn, 0.f);
            const float lo_y = fmaxf(fy - mgn, 0.f), hi_y = fmaxf(1.0f - fy - mgn, 0.f);
            const float lo_z = fmaxf(fz - mgn, 0.f), hi_z = fmaxf(1.0f - fz - mgn, 0.f);
            const float face = fminf(fminf(lo_y, hi_y), fminf(lo_z, hi_z));
            float gbest = octet_best();
            for (int k = 0; k <= K; k++) {
                if (k > 0) {
                    // every row of ring k and beyond is at least (k - 1 + face) cells away in y or in z
                    const float m = (float)(k - 1) + face;
                    if (m * m * h2 > gbest) break;
                }
                const int nrows = k == 0 ? 1 : 8 * k;
                for (int r0 = 0; r0 < nrows; r0 += kRingG) {
                    // ---- one row of the ring per lane: its slab bound, its x-extent, its range of the sorted target
                    const int r = r0 + l8;
                    int dy = 0, dz = 0;
                    if (k > 0) {
                        const int side = r / (2 * k), t = r - side * 2 * k;      // the ring's perimeter, once round
                        dy = side == 0 ? -k + t : (side == 1 ? k : (side == 2 ? k - t : -k));
                        dz = side == 0 ? -k : (side == 1 ? -k + t : (side == 2 ? k : k - t));
                    }
                    const float ey = dy == 0 ? 0.f : (dy < 0 ? lo_y + (float)(-dy - 1) : hi_y + (float)(dy - 1));
                    const float ez = dz == 0 ? 0.f : (dz < 0 ? lo_z + (float)(-dz - 1) : hi_z + (float)(dz - 1));
                    const float bound = (ey * ey + ez * ez) * h2;
                    const int y = cy + dy, z = cz + dz;
                    const bool ok = r < nrows && y >= 0 && y < g.dim[1] && z >= 0 && z < g.dim[2] && !(bound > gbest);
                    unsigned rb = 0u, re = 0u;
                    if (ok) {
                        // cells of the row that can hold a point within the bound: x-cell cx + d (d >= 1) is at least
                        // hi_x + d - 1 cells away, cx - d at least lo_x + d - 1 (one cell more never hurts)
                        const float t = sqrtf(fmaxf(gbest - bound, 0.f) * inv_h2) * (1.0f + 1e-5f) + 1e-4f;
                        const int dxp = min(max((int)floorf(t - hi_x + 1.0f), 0), K + 1);
                        const int dxm = min(max((int)floorf(t - lo_x + 1.0f), 0), K + 1);
                        const int x0 = max(cx - dxm, 0), x1 = min(cx + dxp, g.dim[0] - 1);
                        if (x0 <= x1) {
                            const unsigned row = (unsigned)((z * g.dim[1] + y) * g.dim[0]);
                            rb = start[row + (unsigned)x0];
                            re = start[row + (unsigned)x1 + 1u];
                            nrows_seen += 1ull;
                        }
                    }
                    // ---- the octet scans the eight ranges one after the other, eight candidates per step
#pragma unroll 1
                    for (int j = 0; j < kRingG; j++) {
                        const unsigned b = (unsigned)__shfl((int)rb, obase + j, 64);
                        const unsigned e = (unsigned)__shfl((int)re, obase + j, 64);
                        const float bnd = __shfl(bound, obase + j, 64);
                        if (b >= e || bnd > gbest) continue;        // (octet-uniform)
                        if (l8 == 0) ncand += (unsigned long long)(e - b);
#pragma unroll 1
                        for (unsigned base = b; base < e; base += kRingG * kRingU) {
                            Pt64 q[kRingU];
                            unsigned jc[kRingU];
#pragma unroll
                            for (int u = 0; u < kRingU; u++) {
                                // a slot past the end re-reads the range's first point: evaluating a candidate twice
                                // cannot change the (d2, index) minimum
                                const unsigned ju = base + (unsigned)(l8 + u * kRingG);
                                jc[u] = ju < e ? ju : b;
                                q[u] = sorted64[jc[u]];
                            }
#pragma unroll
                            for (int u = 0; u < kRingU; u++) rank(q[u], jc[u]);
                        }
                        gbest = octet_best();
                    }
                }
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
        Pt64 w8 = prev;
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
        // lane (it mod 8) of the octet adds this query's correspondence to its sums
        if (found && (it & (kRingG - 1)) == l8) {
            double nx = 0.0, ny = 0.0, nz = 0.0;
            if (PLANE) {
                if (nrm64) { const Pt64 n8 = nrm64[(unsigned)w8.w]; nx = n8.x; ny = n8.y; nz = n8.z; }
                else { const float4 n4 = nrm[(unsigned)w8.w]; nx = n4.x; ny = n4.y; nz = n4.z; }
            }
            accumulate_pq_d<PLANE>(acc, pxd, pyd, pzd, w8.x, w8.y, w8.z, nx, ny, nz, off);
        }
    }
    block_reduce_store<NACC>(acc, partials, fold.tickets != nullptr);
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
    if (fold.tickets) fused_fold<PLANE, kBlock, false, kSolveInFold>(fold, partials, row0, lb, bpp, prob);
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

// `nblocks` workgroups per problem, `nprob` problems over shared clouds (st: their loop states or NULL); g.ring > 0.
// state_io: per query the winner's f64 point and original index (all bits set = none), laid out like idx_out: read when
// `warm` (every entry must be none or a point of THIS target), always written.
hipError_t launch_nn_ring(int nblocks, int nprob, int ns, const Pt64 *src64, const Pt64 *sorted64, const unsigned *start,
                          const GridParams &g, const float4 *nrm, const Pt64 *nrm64, const Xform64 &T64, const Offset64 &off,
                          float r2f, int point_to_plane, int32_t *idx_out, float *d2_out, double *d64_out, Pt64 *state_io,
                          int warm, double *partials, unsigned long long *cand_count, const DevIcpState *st,
                          long long out_stride, const FoldArgs &fold, hipStream_t stream)
{
    if (!src64 || !sorted64 || !start || g.ring < 1 || g.sub != 1 || nblocks < 1 || nprob < 1) return hipErrorInvalidValue;
    if (point_to_plane && !nrm && !nrm64) return hipErrorInvalidValue;
    if (point_to_plane)
        hipLaunchKernelGGL(nn_ring_kernel<true>, dim3(nblocks * nprob), dim3(kBlock), 0, stream, ns, src64, sorted64, start, g,
                           nrm, nrm64, T64, off, r2f, idx_out, d2_out, d64_out, state_io, warm, partials, cand_count, st, nblocks,
                           out_stride, fold);
    else
        hipLaunchKernelGGL(nn_ring_kernel<false>, dim3(nblocks * nprob), dim3(kBlock), 0, stream, ns, src64, sorted64, start, g,
                           nrm, nrm64, T64, off, r2f, idx_out, d2_out, d64_out, state_io, warm, partials, cand_count, st, nblocks,
                           out_stride, fold);
    return hipGetLastError();
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
