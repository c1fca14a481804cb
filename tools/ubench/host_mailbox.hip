// host_mailbox.hip -- round trip of a word through mapped, coherent host memory: the host writes k, ONE wave of a
// running kernel polls for it (system-scope loads over PCIe) and answers with k in another word, the host polls for
// the answer.  The latency floor of a persistent kernel that takes its next transform from the host.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/host_mailbox.hip -o /tmp/host_mailbox && /tmp/host_mailbox
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void echo_kernel(const unsigned long long *cmd, unsigned long long *ack, int rounds, long long budget, int *gave_up)
{
    const int lane = threadIdx.x;
    for (int k = 1; k <= rounds; k++) {
        const long long t0 = (long long)wall_clock64();
        unsigned long long w = 0;
        for (;;) {
            if (lane < 25) w = __hip_atomic_load(cmd + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const bool ok = lane >= 25 || (unsigned)(w >> 32) == (unsigned)k;
            if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) break;
            if ((long long)wall_clock64() - t0 > budget) { if (lane == 0) *gave_up = k; return; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane < 25) __hip_atomic_store(ack + lane, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int main()
{
    unsigned long long *h = nullptr, *d = nullptr;
    int *flag_h = nullptr, *flag_d = nullptr;
    CHECK(hipHostMalloc((void **)&h, 1024, hipHostMallocMapped | hipHostMallocCoherent));
    CHECK(hipHostMalloc((void **)&flag_h, 64, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(h, 0, 1024);
    *flag_h = 0;
    CHECK(hipHostGetDevicePointer((void **)&d, h, 0));
    CHECK(hipHostGetDevicePointer((void **)&flag_d, flag_h, 0));
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int rounds = 2000;
    volatile unsigned long long *cmd = h, *ack = h + 64;
    hipLaunchKernelGGL(echo_kernel, dim3(1), dim3(64), 0, s, d, d + 64, rounds, 50000000ll, flag_d);
    CHECK(hipGetLastError());
    double worst = 0, total = 0;
    for (int k = 1; k <= rounds; k++) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 25; i++) cmd[i] = ((unsigned long long)k << 32) | (unsigned)i;
        bool seen = false;
        for (long long spin = 0; spin < 2000000000ll && !seen; spin++) {
            seen = true;
            for (int i = 0; i < 25; i++) if ((ack[i] >> 32) != (unsigned long long)k) { seen = false; break; }
            if (!seen && (spin & 0xFFFFF) == 0xFFFFF && hipStreamQuery(s) != hipErrorNotReady) break;
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (!seen) { std::printf("round %d: no answer (kernel gave up at %d)\n", k, *flag_h); break; }
        total += us;
        if (us > worst) worst = us;
    }
    CHECK(hipStreamSynchronize(s));
    std::printf("host -> polling wave -> host: %.2f us average over %d rounds, worst %.2f; gave_up %d\n", total / rounds, rounds, worst, *flag_h);
    return 0;
}
