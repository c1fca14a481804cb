// grid_coop_probe.h -- the measurement build of grid_coop.hip.  The product build defines none of the macros below
// and gets no-ops; the side library built by tools/coop_phases.py defines
//   -DVISMA_COOP_DEBUG_PHASES=2   per-wave clocks: one store per wave per phase border, nothing shared, nothing forced
//                                 (the build DESIGN 4.1c's per-wave pictures come from)
//   -DVISMA_COOP_DEBUG_PHASES=1   the same clocks of wave 0 of every workgroup, summed over the workgroups (shared
//                                 atomics: perturbs the launch)
// COOP_MARK(k) stamps the wave's clock into slot k.  Marks of a round (round 4: certificate + compaction + the
// workgroup's chunk list): 9 phase A done | 10 past the first barrier | 0 query taken over | 1 rows asked for |
// 2 list written | 3 chunks worked off | 4 merged | 5 f64 winner ranked | 11 search done (0..5, 11: searching waves
// only) | 12 past the last barrier | 13 phase C done | 6 rounds done | 7 partial row | 8 fold; 14, 15: HW_ID, XCC_ID.
// (The truncation builds of round 3 -- every query's work cut off after phase k -- went with the barriers the phases
// now hold.)
#pragma once

#if defined(VISMA_COOP_DEBUG_PHASES)

namespace visma {
__device__ unsigned long long g_coop_phase[16];             // mode 1: clocks per phase, summed over workgroups
__device__ unsigned long long g_coop_span[4 * 8192];        // per wave of the last launch: first and last clock
__device__ unsigned long long g_coop_marks[16 * 8192];      // mode 2: per wave, the clock at every phase border
}  // namespace visma

#define COOP_WAVE_SLOT() ((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)))   /* waves of the launch, in order */
#if VISMA_COOP_DEBUG_PHASES == 2
#define COOP_PROBE_CLEAR()                                                                                      \
    do {                                                                                                        \
        if ((threadIdx.x & 63) < 14 && COOP_WAVE_SLOT() < 8192)      /* (a wave that skips a phase leaves no stale stamp) */ \
            g_coop_marks[COOP_WAVE_SLOT() * 16 + (threadIdx.x & 63)] = 0ull;               \
        if ((threadIdx.x & 63) == 0 && COOP_WAVE_SLOT() < 8192) {    /* where the wave runs: HW_ID (4), XCC_ID (20) */   \
            g_coop_marks[COOP_WAVE_SLOT() * 16 + 14] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); \
            g_coop_marks[COOP_WAVE_SLOT() * 16 + 15] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); \
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
#elif VISMA_COOP_DEBUG_PHASES == 2
#define COOP_MARK(k)                                                                                            \
    do {                                                                                                        \
        if ((threadIdx.x & 63) == 0 && COOP_WAVE_SLOT() < 8192)                                                       \
            g_coop_marks[COOP_WAVE_SLOT() * 16 + (k)] = __builtin_amdgcn_s_memrealtime();  \
    } while (0)
#else
#define COOP_MARK(k) do { (void)stamp_; } while (0)
#endif

#define COOP_WAVE_DONE()                                                                                        \
    do {                                                                                                        \
        if ((threadIdx.x & 63) == 0 && COOP_WAVE_SLOT() < 8192) {                                                     \
            const int w_ = COOP_WAVE_SLOT();                                                 \
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
#define COOP_WAVE_DONE() do { } while (0)

#endif
