// fixedpoint_fold.hip -- VERDICT r5 item 5(a): would an order-independent accumulation of the 38 statistics in fixed-point
// integer atomics at the L2 (3 x 64-bit limbs per statistic: integer addition commutes, so no tree and no fixed order is
// needed for a reproducible sum) end a pass sooner than the fold the kernels run?  What decides is the TAIL: the time from
// the last workgroup's partial sums being ready to the totals being readable by one workgroup.  Three kernels over the
// geometry of a C4 pass (1,024 workgroups of 256 threads), each timed over many launches, an empty kernel subtracted:
//   ticket : what fused_fold does (device_common.h): every workgroup stores its row of 23 doubles, takes a ticket in its
//            group of 16; the last of a group sums the 16 rows, stores the group row, takes the second-level ticket; the last
//            of all sums the 64 group rows.
//   limbs  : every workgroup adds its 38 x 3 limbs with 114 device-scope atomic adds (no return value), waits for them,
//            takes ONE ticket; the last workgroup reads the 114 words.
//   limbs8 : the same into one of EIGHT copies of the 114 words (by workgroup mod 8: one per XCD's share), the last workgroup
//            reads 8 x 114 words.
// and the same three with a STRAGGLER: one workgroup arrives 20 us after the others (a pass ends with its slowest workgroup:
// everything the others did is long complete) -- the tail a pass really waits for.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/fixedpoint_fold.hip -o /tmp/fixedpoint_fold && /tmp/fixedpoint_fold
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kRows = 1024, kGroup = 16, kAcc = 23, kLimbs = 114;

__device__ __forceinline__ void straggle(int who, long long ticks)
{
    if ((int)blockIdx.x == who && ticks > 0) {
        const long long t0 = (long long)wall_clock64();
        while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    }
}

__global__ __launch_bounds__(256) void k_empty(int who, long long ticks) { straggle(who, ticks); }

__global__ __launch_bounds__(256) void k_ticket(double *rows, double *rows2, unsigned *tk, double *out, int who, long long ticks)
{
    __shared__ int flag;
    __shared__ double part[8][33];
    straggle(who, ticks);
    const int tid = threadIdx.x, b = blockIdx.x, grp = b / kGroup;
    if (tid < kAcc) __hip_atomic_store(rows + (long long)b * 32 + tid, (double)(b + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(tk + 1 + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag = t == kGroup - 1;
        if (flag) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); __hip_atomic_store(tk + 1 + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    __syncthreads();
    if (!flag) return;
    const int sa = tid & 31, sg = tid >> 5;
    {
        double v = 0.0;
        if (sa < kAcc)
            for (int r = sg; r < kGroup; r += 8) v += __hip_atomic_load(rows + (long long)(grp * kGroup + r) * 32 + sa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        part[sg][sa] = v;
    }
    __syncthreads();
    if (tid < kAcc) {
        double t = 0.0;
        for (int g = 0; g < 8; g++) t += part[g][tid];
        __hip_atomic_store(rows2 + (long long)grp * 32 + tid, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag = t == kRows / kGroup - 1;
        if (flag) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    __syncthreads();
    if (!flag) return;
    {
        double v = 0.0;
        if (sa < kAcc) {
            double w[8];
#pragma unroll
            for (int u = 0; u < 8; u++) w[u] = __hip_atomic_load(rows2 + (long long)(sg + u * 8) * 32 + sa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 8; u++) v += w[u];
        }
        part[sg][sa] = v;
    }
    __syncthreads();
    if (tid < kAcc) {
        double t = 0.0;
        for (int g = 0; g < 8; g++) t += part[g][tid];
        out[tid] = t;
    }
}

template <int COPIES>
__global__ __launch_bounds__(256) void k_limbs(unsigned long long *acc, unsigned *tk, unsigned long long *out, int who, long long ticks)
{
    __shared__ int flag;
    straggle(who, ticks);
    const int tid = threadIdx.x, b = blockIdx.x;
    unsigned long long *mine = acc + (long long)(b % COPIES) * 128;
    if (tid < kLimbs) __hip_atomic_fetch_add(mine + tid, (unsigned long long)(b + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag = t == kRows - 1;
        if (flag) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    __syncthreads();
    if (!flag) return;
    if (tid < kLimbs) {
        unsigned long long w[COPIES], t = 0ull;
#pragma unroll
        for (int c = 0; c < COPIES; c++) w[c] = __hip_atomic_load(acc + (long long)c * 128 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int c = 0; c < COPIES; c++) t += w[c];
        out[tid] = t;
        // (the accumulators must be zero again for the next pass: one more store per word)
#pragma unroll
        for (int c = 0; c < COPIES; c++) __hip_atomic_store(acc + (long long)c * 128 + tid, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <class F>
static double median_us(F launch, int reps)
{
    std::vector<float> t;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 20; i++) launch();
    (void)hipDeviceSynchronize();
    for (int i = 0; i < reps; i++) {
        (void)hipEventRecord(a, 0);
        launch();
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, a, b);
        t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return t[t.size() / 2];
}

int main()
{
    double *rows = nullptr, *rows2 = nullptr, *out = nullptr;
    unsigned long long *acc = nullptr, *out2 = nullptr;
    unsigned *tk = nullptr;
    CHECK(hipMalloc((void **)&rows, sizeof(double) * 32 * kRows));
    CHECK(hipMalloc((void **)&rows2, sizeof(double) * 32 * 64));
    CHECK(hipMalloc((void **)&out, sizeof(double) * 32));
    CHECK(hipMalloc((void **)&acc, sizeof(unsigned long long) * 128 * 8));
    CHECK(hipMalloc((void **)&out2, sizeof(unsigned long long) * 128));
    CHECK(hipMalloc((void **)&tk, sizeof(unsigned) * 128));
    CHECK(hipMemset(acc, 0, sizeof(unsigned long long) * 128 * 8));
    CHECK(hipMemset(tk, 0, sizeof(unsigned) * 128));
    const int reps = 300;
    for (long long ticks : {0ll, 2000ll}) {                 // 100 MHz: 2000 ticks = 20 us
        const int who = 517;
        const double e = median_us([&] { hipLaunchKernelGGL(k_empty, dim3(kRows), dim3(256), 0, 0, who, ticks); }, reps);
        const double t = median_us([&] { hipLaunchKernelGGL(k_ticket, dim3(kRows), dim3(256), 0, 0, rows, rows2, tk, out, who, ticks); }, reps);
        const double l1 = median_us([&] { hipLaunchKernelGGL(k_limbs<1>, dim3(kRows), dim3(256), 0, 0, acc, tk, out2, who, ticks); }, reps);
        const double l8 = median_us([&] { hipLaunchKernelGGL(k_limbs<8>, dim3(kRows), dim3(256), 0, 0, acc, tk, out2, who, ticks); }, reps);
        std::printf("%s: empty launch %.2f us | ticket fold (16 x 64 rows of 23 doubles) +%.2f us | 114 limb atomics per workgroup, one copy +%.2f us | eight copies +%.2f us\n",
                    ticks ? "one workgroup 20 us late " : "all workgroups together ", e, t - e, l1 - e, l8 - e);
    }
    // the sums are what they must be (integer: exact)
    unsigned long long h[128];
    CHECK(hipMemcpy(h, out2, sizeof(h), hipMemcpyDeviceToHost));
    unsigned long long want0 = 0;
    for (int b = 0; b < kRows; b++) want0 += (unsigned long long)b;
    std::printf("limb 0 = %llu (expected %llu), limb 113 = %llu (expected %llu)\n", h[0], want0, h[113], want0 + 113ull * kRows);
    return 0;
}
