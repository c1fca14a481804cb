// grid_coop_probe.h -- the measurement builds of grid_coop.hip.  The product build defines none of the macros below
// and gets no-ops; side libraries built by tools/coop_phases.py and tools/truncate_probe.py define
//   -DVISMA_COOP_DEBUG_PHASES=1   clocks between the phase borders of wave 0 of every workgroup, summed over the
//                                 workgroups (shared atomics and forced waits: perturbs the launch by ~30 %)
//   -DVISMA_COOP_DEBUG_PHASES=2   per-wave clocks: one store per wave per phase border, nothing shared, nothing forced
//                                 (the build DESIGN 4.1c's per-wave picture comes from)
//   -DVISMA_COOP_STOP_AFTER=k     every query's work ends after phase k (k = 0, 1, 4, 5; the fold still runs, the
//                                 results are garbage): launch time as a function of how far the queries get
// COOP_PHASE(k, u, f) sits at the border after phase k inside coop_body's per-query lambda and names a 32-bit and a
// float value that the work so far produced (kept alive by the stop / the forced wait); COOP_MARK(k) only stamps.
// Marks of a round (round 4: certificate + compaction): 9 phase A done | 10 past the first barrier | 0..5 the search
// (searching waves only) | 11 search done | 12 past the second barrier | 13 phase C done | 6 rounds done | 7 partial row | 8 fold.
#pragma once

#if defined(VISMA_COOP_DEBUG_PHASES) || defined(VISMA_COOP_STOP_AFTER)

#ifndef VISMA_COOP_DEBUG_PHASES
#define VISMA_COOP_DEBUG_PHASES 0
#endif
#ifndef VISMA_COOP_STOP_AFTER
#define VISMA_COOP_STOP_AFTER 99
#endif

namespace visma {
__device__ unsigned long long g_coop_phase[16];             // mode 1: clocks per phase, summed over workgroups
__device__ unsigned long long g_coop_span[4 * 8192];        // per wave of the last launch: first and last clock
__device__ unsigned long long g_coop_marks[16 * 8192];      // mode 2: per wave, the clock at every phase border
}  // namespace visma

#if VISMA_COOP_DEBUG_PHASES == 2
#define COOP_PROBE_CLEAR()                                                                                      \
    do {                                                                                                        \
        if ((threadIdx.x & 63) < 14 && blockIdx.x < 2048)      /* (a wave that skips a phase leaves no stale stamp) */ \
            g_coop_marks[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (threadIdx.x & 63)] = 0ull;               \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048) {    /* where the wave runs: HW_ID (4), XCC_ID (20) */   \
            g_coop_marks[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + 14] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); \
            g_coop_marks[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + 15] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); \
        }                                                                                                       \
    } while (0)
#else
#define COOP_PROBE_CLEAR() do { } while (0)
#endif
#define COOP_PROBE_BEGIN()                                                                                      \
    unsigned long long stamp_ = __builtin_amdgcn_s_memrealtime();                                               \
    const unsigned long long stamp0_ = stamp_;                                                                  \
    (void)stamp0_;                                                                                              \
    COOP_PROBE_CLEAR()

#if VISMA_COOP_DEBUG_PHASES == 1
#define COOP_MARK(k)                                                                                            \
    do {                                                                                                        \
        const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();                                       \
        if (threadIdx.x == 0) atomicAdd(&g_coop_phase[k], now_ - stamp_);                                       \
        stamp_ = now_;                                                                                          \
    } while (0)
#define COOP_KEEP(u, f) asm volatile("" ::"v"(u), "v"(f))
#elif VISMA_COOP_DEBUG_PHASES == 2
#define COOP_MARK(k)                                                                                            \
    do {                                                                                                        \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048)                                                       \
            g_coop_marks[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (k)] = __builtin_amdgcn_s_memrealtime();  \
    } while (0)
#define COOP_KEEP(u, f) do { } while (0)
#else
#define COOP_MARK(k) do { (void)stamp_; } while (0)
#define COOP_KEEP(u, f) do { } while (0)
#endif

#define COOP_PHASE(k, u, f)                                                                                     \
    do {                                                                                                        \
        COOP_KEEP(u, f);                                                                                        \
        COOP_MARK(k);                                                                                           \
        if (VISMA_COOP_STOP_AFTER == (k)) {                                                                     \
            if (active) { idx_out[i] = (int)(u); d2_out[i] = (float)(f); }                                      \
            return;                                                                                             \
        }                                                                                                       \
    } while (0)

#define COOP_WAVE_DONE()                                                                                        \
    do {                                                                                                        \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048) {                                                     \
            const int w_ = blockIdx.x * 4 + (threadIdx.x >> 6);                                                 \
            g_coop_span[2 * w_] = stamp0_;                                                                      \
            g_coop_span[2 * w_ + 1] = __builtin_amdgcn_s_memrealtime();                                         \
        }                                                                                                       \
    } while (0)

extern "C" __attribute__((visibility("default"))) int visma_debug_coop_marks(unsigned long long *out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(visma::g_coop_marks), sizeof(unsigned long long) * n) != hipSuccess;
}
extern "C" __attribute__((visibility("default"))) int visma_debug_coop_spans(unsigned long long *out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(visma::g_coop_span), sizeof(unsigned long long) * n) != hipSuccess;
}
extern "C" __attribute__((visibility("default"))) int visma_debug_coop_phases(unsigned long long *out16, int reset)
{
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(visma::g_coop_phase), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (reset) {
        const unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(visma::g_coop_phase), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}

#else   /* the product build */

#define COOP_PROBE_BEGIN() do { } while (0)
#define COOP_MARK(k) do { } while (0)
#define COOP_PHASE(k, u, f) do { } while (0)
#define COOP_WAVE_DONE() do { } while (0)

#endif
