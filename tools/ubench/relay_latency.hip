// relay_latency.hip -- what would the in-launch solve of the persistent kernel buy?  Today a pass ends with the statistics
// going to the host over PCIe, the host solving (0.45 us) and the next transform coming back through the BAR: 2.9 us from
// the slowest workgroup's end to the next pass's begin (tools/persist_timeline.py).  Solving inside the launch replaces
// that by: one thread's solve + a store to device memory that the first waves of the other 1,023 workgroups -- on all
// eight XCDs, each with its own L2 -- poll.  This measures the second part: the ping-pong of a 25-word block (the
// persistent kernel's command block: 24 lanes a word, self-validating tags) between workgroup 0 and a workgroup on
// ANOTHER XCD through device memory (agent-scope stores / loads, as the relay of grid_coop.hip), round trips / 2.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/relay_latency.hip -o /tmp/relay_latency && /tmp/relay_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void pingpong(unsigned long long *a, unsigned long long *b, int rounds, int partner, unsigned long long *ticks, int sleep)
{
    const int lane = threadIdx.x;
    if (blockIdx.x != 0 && (int)blockIdx.x != partner) return;
    const bool first = blockIdx.x == 0;
    unsigned long long *mine = first ? a : b, *theirs = first ? b : a;
    const unsigned long long t0 = wall_clock64();
    for (int k = 1; k <= rounds; k++) {
        if (first && lane < 25) __hip_atomic_store(theirs + lane, ((unsigned long long)k << 32) | (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (;;) {
            unsigned long long w = 0;
            if (lane < 25) w = __hip_atomic_load(mine + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool ok = lane >= 25 || (unsigned)(w >> 32) == (unsigned)k;
            if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) break;
            if (sleep) __builtin_amdgcn_s_sleep(1);
        }
        if (!first && lane < 25) __hip_atomic_store(theirs + lane, ((unsigned long long)k << 32) | (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (first && lane == 0) *ticks = wall_clock64() - t0;
}

int main()
{
    unsigned long long *buf = nullptr, *ticks = nullptr;
    CHECK(hipMalloc((void **)&buf, 8192));
    CHECK(hipMalloc((void **)&ticks, 8));
    const int rounds = 20000;
    for (int partner : {1, 2, 4, 8, 9, 255}) {
        for (int sleep = 0; sleep < 2; sleep++) {
            CHECK(hipMemset(buf, 0, 8192));
            hipLaunchKernelGGL(pingpong, dim3(256), dim3(64), 0, 0, buf, buf + 512, rounds, partner, ticks, sleep);
            CHECK(hipDeviceSynchronize());
            unsigned long long t = 0;
            CHECK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
            std::printf("workgroup 0 <-> workgroup %3d (%s): one way %.2f us (100 MHz clock, %d round trips)\n", partner,
                        sleep ? "s_sleep 1 between polls" : "tight polls", (double)t / rounds / 2.0 / 100.0, rounds);
        }
    }
    return 0;
}
