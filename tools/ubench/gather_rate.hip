// gather_rate.hip -- what the memory system delivers for the grid search's access
// pattern: short contiguous runs (RUN bytes) at random places of a large buffer.
// Each group of L lanes reads one run (L x 16 B), `ILP` independent runs in flight
// per lane, addresses from a hash (no dependent chain).  Footprint and run length
// are parameters; prints useful GB/s (run bytes) -- compare with 8 TB/s streaming.
// Build: hipcc --offload-arch=gfx950 -O3 gather_rate.hip -o gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int L, int ILP>
__global__ __launch_bounds__(256) void gather(const float4 *__restrict__ buf, unsigned nrun_slots, int iters,
                                              int local, float4 *out)
{
    const unsigned gid = (blockIdx.x * 256u + threadIdx.x) / L, sub = threadIdx.x % L;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; it++) {
        float4 q[ILP];
#pragma unroll
        for (int u = 0; u < ILP; u++) {
            unsigned key = (gid * (unsigned)iters + it) * ILP + u;
            unsigned slot;
            if (local) {
                // neighbouring groups read neighbouring places (like Morton-ordered queries):
                // a random jump of +-`local` slots around a slowly moving centre
                const unsigned centre = (unsigned)(((unsigned long long)gid * nrun_slots) / (gridDim.x * 256u / L));
                slot = (centre + hash32(key) % (unsigned)local) % nrun_slots;
            } else {
                slot = hash32(key) % nrun_slots;
            }
            q[u] = buf[(size_t)slot * L + sub];
        }
#pragma unroll
        for (int u = 0; u < ILP; u++) { acc.x += q[u].x; acc.y += q[u].y; acc.z += q[u].z; acc.w += q[u].w; }
    }
    if (acc.x == 12345.f) out[0] = acc;
}

// One 128 B line per group and iteration, fetched either by 8 lanes in ONE instruction
// (L=8, PIECES=1) or by L lanes in 8/L separate instructions (the grid kernel's j, j+G, ...
// pattern): do in-flight loads to the same line merge in the L1, or does each go to L2?
template <int L>
__global__ __launch_bounds__(256) void sameline(const float4 *__restrict__ buf, unsigned nlines, int iters, float4 *out)
{
    constexpr int PIECES = 8 / L;
    const unsigned gid = (blockIdx.x * 256u + threadIdx.x) / L, sub = threadIdx.x % L;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; it++) {
        const unsigned line = hash32(gid * (unsigned)iters + it) % nlines;
        float4 q[PIECES];
#pragma unroll
        for (int u = 0; u < PIECES; u++) q[u] = buf[(size_t)line * 8 + u * L + sub];
#pragma unroll
        for (int u = 0; u < PIECES; u++) { acc.x += q[u].x; acc.y += q[u].y; acc.z += q[u].z; acc.w += q[u].w; }
    }
    if (acc.x == 12345.f) out[0] = acc;
}

template <int L>
static void run_sameline(const float4 *buf, size_t bytes, float4 *out)
{
    const unsigned nlines = (unsigned)(bytes / 128);
    const int blocks = 4096 * L / 8 * 4, iters = 64;      // same number of lines for every L
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((sameline<L>), dim3(blocks), dim3(256), 0, 0, buf, nlines, iters, out);
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((sameline<L>), dim3(blocks), dim3(256), 0, 0, buf, nlines, iters, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    const double lines = (double)blocks * 256 / L * iters;
    printf("same-line: %d lanes x %d loads per 128 B line: %8.1f M lines/s, %.3f ms\n", L, 8 / L, lines / ms * 1e-3, ms);
}

template <int L, int ILP>
static void run(const float4 *buf, size_t bytes, int local, float4 *out)
{
    const unsigned nrun_slots = (unsigned)(bytes / (16 * L));
    const int blocks = 4096, iters = 64 / ILP * 4;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((gather<L, ILP>), dim3(blocks), dim3(256), 0, 0, buf, nrun_slots, iters, local, out);
    hipEventRecord(a);
    for (int r = 0; r < 5; r++)
        hipLaunchKernelGGL((gather<L, ILP>), dim3(blocks), dim3(256), 0, 0, buf, nrun_slots, iters, local, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    const double runs = (double)blocks * 256 / L * iters * ILP;
    printf("footprint %5zu MB  run %4d B  ilp %d  local %7d : %8.1f GB/s useful, %7.1f M runs/s, %.3f ms\n",
           bytes >> 20, 16 * L, ILP, local, runs * 16 * L / ms * 1e-6, runs / ms * 1e-3, ms);
}

int main(int argc, char **argv)
{
    const size_t mb = argc > 1 ? atoi(argv[1]) : 200;
    const size_t bytes = mb << 20;
    float4 *buf, *out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 64);
    hipMemset(buf, 0, bytes);
    run_sameline<8>(buf, bytes, out);
    run_sameline<4>(buf, bytes, out);
    run_sameline<2>(buf, bytes, out);
    run_sameline<1>(buf, bytes, out);
    for (int local : {0}) {
        run<1, 4>(buf, bytes, local, out);
        run<2, 4>(buf, bytes, local, out);
        run<4, 4>(buf, bytes, local, out);
        run<8, 4>(buf, bytes, local, out);
        run<8, 8>(buf, bytes, local, out);
        run<16, 4>(buf, bytes, local, out);
        run<64, 4>(buf, bytes, local, out);
    }
    return 0;
}
