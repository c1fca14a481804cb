// device_mailbox.hip -- can the HOST store straight into device memory (large BAR), and what does the round trip
// host store -> polling wave (device memory, no PCIe read) -> answer in host memory -> host cost?  Compare
// host_mailbox.hip (the wave polls HOST memory over PCIe: 2.7 us).  Tries fine-grained device memory
// (hipExtMallocWithFlags) and plain hipMalloc; a store the platform does not allow kills the CHILD process only.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/device_mailbox.hip -o /tmp/device_mailbox && /tmp/device_mailbox
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <sys/wait.h>
#include <unistd.h>
#include <immintrin.h>

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

static int run(int mode)
{
    int large_bar = -1;
    (void)hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0);
    unsigned long long *h = nullptr, *hd = nullptr, *dev = nullptr;
    int *flag_h = nullptr, *flag_d = nullptr;
    CHECK(hipHostMalloc((void **)&h, 1024, hipHostMallocMapped | hipHostMallocCoherent));
    CHECK(hipHostMalloc((void **)&flag_h, 64, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(h, 0, 1024);
    *flag_h = 0;
    CHECK(hipHostGetDevicePointer((void **)&hd, h, 0));
    CHECK(hipHostGetDevicePointer((void **)&flag_d, flag_h, 0));
    if (mode == 0) CHECK(hipExtMallocWithFlags((void **)&dev, 4096, hipDeviceMallocFinegrained));
    else if (mode == 1) CHECK(hipExtMallocWithFlags((void **)&dev, 4096, hipDeviceMallocUncached));
    else CHECK(hipMalloc((void **)&dev, 4096));
    CHECK(hipMemset(dev, 0, 4096));
    CHECK(hipDeviceSynchronize());
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int rounds = 2000;
    volatile unsigned long long *cmd = dev, *ack = h + 64;    // the HOST stores to `dev` directly
    std::printf("mode %d (%s), large BAR attribute %d: ", mode, mode == 0 ? "fine-grained device memory" : mode == 1 ? "uncached device memory" : "hipMalloc", large_bar);
    std::fflush(stdout);
    hipLaunchKernelGGL(echo_kernel, dim3(1), dim3(64), 0, s, dev, hd + 64, rounds, 50000000ll, flag_d);
    CHECK(hipGetLastError());
    double worst = 0, total = 0;
    for (int k = 1; k <= rounds; k++) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 25; i++) cmd[i] = ((unsigned long long)k << 32) | (unsigned)i;
        _mm_sfence();                                  // (the BAR is write-combining: without this the stores linger ~200 us)
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
    std::printf("host store -> polling wave -> host: %.2f us average over %d rounds, worst %.2f; gave_up %d\n", total / rounds, rounds, worst, *flag_h);
    return 0;
}

int main()
{
    for (int mode = 0; mode < 3; mode++) {
        std::fflush(stdout);
        const pid_t pid = fork();
        if (pid == 0) { const int r = run(mode); std::fflush(stdout); _exit(r); }
        int st = 0;
        waitpid(pid, &st, 0);
        if (WIFSIGNALED(st)) std::printf("... the child died with signal %d (the host cannot store there)\n", WTERMSIG(st));
    }
    return 0;
}
