// icp_loop.hip -- the ICP loop advanced on the device.
//
// The synchronous API pays one stream synchronisation + one 304-byte D2H copy +
// one host solve per iteration (~25 us on MI355X, as much as the kernels of a
// 5k x 20k problem).  Here the per-iteration solve, the compose T <- update * T
// and the stop test of O3D/Core/Registration/Registration.cpp:169-184 run in a
// one-thread epilogue of the fold kernel, on the state kept in HBM
// (DevIcpState); the NN / reduction kernels read the transform from that state.
// The host only enqueues launches and reads the state back once per chunk.
//
// The solve is the SAME code as the host's (host_math.hpp is host+device):
// closed-form Kabsch/Umeyama from the reduced moments, or the 6x6 Gauss-Newton
// step with Euler / exponential-map retraction.
#include "device_common.h"
#include "icp_state.h"

namespace visma {

// 256 threads, not the 1024 of the plain fold kernel: the one-thread solve below needs more
// than the 128 VGPRs a 1024-thread workgroup leaves per lane (it spilled ~1500 scratch
// accesses per solve), and a device loop folds few partial rows per problem anyway.
constexpr int kSolveThreads = 256;

template <bool PLANE>
__global__ __launch_bounds__(kSolveThreads) void finalize_solve_kernel(const double *__restrict__ partials,
                                                              int nblocks, DevIcpState *st,
                                                              int do_solve)
{
    st += blockIdx.x;                                        // one workgroup per problem
    partials += (long long)blockIdx.x * nblocks * kReduceAcc;
    if (!st->active) return;
    fold_partials<PLANE, kSolveThreads>(partials, nblocks, st->stats);
    if (do_solve && threadIdx.x == 0) advance_state(st);   // same thread wrote the stats
}

// one workgroup per problem: the statistics are already in st->stats (fused fold / all-reduce)
__global__ __launch_bounds__(64) void solve_state_kernel(DevIcpState *st)
{
    // the whole wave brings the state into LDS (one round trip instead of one per field read by the
    // solving thread), thread 0 solves there, the wave writes it back
    static_assert(sizeof(DevIcpState) % 8 == 0, "state copied as 8-byte words");
    constexpr int kWords = (int)(sizeof(DevIcpState) / 8);
    __shared__ unsigned long long sst[kWords];
    st += blockIdx.x;
    if (!st->active) return;
    const unsigned long long *g = reinterpret_cast<const unsigned long long *>(st);
    for (int k = threadIdx.x; k < kWords; k += 64) sst[k] = g[k];
    __syncthreads();
    if (threadIdx.x == 0) advance_state(reinterpret_cast<DevIcpState *>(sst));
    __syncthreads();
    unsigned long long *o = reinterpret_cast<unsigned long long *>(st);
    for (int k = threadIdx.x; k < kWords; k += 64) o[k] = sst[k];
}

// `plane` is a host copy of st->plane (chooses the accumulator layout)
static hipError_t launch_fs(const double *partials, int nblocks, DevIcpState *st, int plane,
                            int do_solve, int nprob, hipStream_t stream)
{
    if (plane)
        hipLaunchKernelGGL(finalize_solve_kernel<true>, dim3(nprob), dim3(kSolveThreads), 0, stream, partials,
                           nblocks, st, do_solve);
    else
        hipLaunchKernelGGL(finalize_solve_kernel<false>, dim3(nprob), dim3(kSolveThreads), 0, stream, partials,
                           nblocks, st, do_solve);
    return hipGetLastError();
}

hipError_t launch_finalize_solve(const double *partials, int nblocks, DevIcpState *st, int plane,
                                 int nprob, hipStream_t stream)
{
    return launch_fs(partials, nblocks, st, plane, 1, nprob, stream);
}

hipError_t launch_finalize_state(const double *partials, int nblocks, DevIcpState *st, int plane,
                                 hipStream_t stream)
{
    return launch_fs(partials, nblocks, st, plane, 0, 1, stream);
}

template <bool PLANE>
__global__ __launch_bounds__(kSolveThreads) void finalize_solve_batch_kernel(const double *__restrict__ partials,
                                                                   const ProbDesc *__restrict__ descs,
                                                                   DevIcpState *st)
{
    const ProbDesc d = descs[blockIdx.x];                    // one workgroup per problem
    st += blockIdx.x;
    if (!st->active) return;
    fold_partials<PLANE, kSolveThreads>(partials + (long long)d.first_block * kReduceAcc, d.nblocks, st->stats);
    if (threadIdx.x == 0) advance_state(st);
}

hipError_t launch_finalize_solve_batch(const double *partials, const ProbDesc *descs, DevIcpState *st,
                                       int nprob, hipStream_t stream, int plane)
{
    if (plane)
        hipLaunchKernelGGL(finalize_solve_batch_kernel<true>, dim3(nprob), dim3(kSolveThreads), 0, stream, partials,
                           descs, st);
    else
        hipLaunchKernelGGL(finalize_solve_batch_kernel<false>, dim3(nprob), dim3(kSolveThreads), 0, stream, partials,
                           descs, st);
    return hipGetLastError();
}

hipError_t launch_solve_state(DevIcpState *st, int nprob, hipStream_t stream)
{
    hipLaunchKernelGGL(solve_state_kernel, dim3(nprob), dim3(64), 0, stream, st);
    return hipGetLastError();
}

}  // namespace visma
