// valu_rate.hip -- issue-rate micro-benchmark of the fp32 VALU ops the NN kernels use.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 4096
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, float a0, float b0)
{
    typedef float v2 __attribute__((ext_vector_type(2)));
    float x0 = a0 + threadIdx.x, x1 = a0 * 2, x2 = a0 * 3, x3 = a0 * 4, x4 = a0 * 5, x5 = a0 * 6, x6 = a0 * 7, x7 = a0 * 8;
    v2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    v2 pb = {b0, b0 * 0.5f};
    for (int i = 0; i < REP; i++) {
        if (OP == 0) {  // v_fma_f32 x8 independent chains
            asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                         "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b0));
        } else if (OP == 1) {  // v_pk_fma_f32 x4 (8 scalar fmas)
            asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
        } else if (OP == 2) {  // v_pk_add_f32 x4
            asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
        } else if (OP == 3) {  // v_sub_f32 x8
            asm volatile("v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %8, %1\n v_sub_f32 %2, %8, %2\n v_sub_f32 %3, %8, %3\n"
                         "v_sub_f32 %4, %8, %4\n v_sub_f32 %5, %8, %5\n v_sub_f32 %6, %8, %6\n v_sub_f32 %7, %8, %7"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b0));
        } else if (OP == 4) {  // v_min3_f32 x8
            asm volatile("v_min3_f32 %0, %0, %8, %1\n v_min3_f32 %1, %1, %8, %2\n v_min3_f32 %2, %2, %8, %3\n v_min3_f32 %3, %3, %8, %4\n"
                         "v_min3_f32 %4, %4, %8, %5\n v_min3_f32 %5, %5, %8, %6\n v_min3_f32 %6, %6, %8, %7\n v_min3_f32 %7, %7, %8, %0"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b0));
        } else if (OP == 5) {  // v_pk_mul_f32 x4
            asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int OP>
double run(float *d, const char *name, int insts_per_iter, int lane_ops_per_inst)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.9999f);
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.9999f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    const double wave_insts = (double)blocks * 4 * REP * insts_per_iter;
    const double lane_ops = wave_insts * 64 * lane_ops_per_inst;
    printf("%-14s %8.3f ms  %7.2f G wave-inst/s  %7.2f T scalar-ops/s  (cycles per wave-inst per SIMD at 2.4 GHz: %.2f)\n", name, ms,
           wave_insts / ms / 1e6, lane_ops / ms / 1e9, 2.4e9 * 1024 / (wave_insts / (ms * 1e-3)));
    return ms;
}

int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>(d, "v_fma_f32", 8, 1);
    run<1>(d, "v_pk_fma_f32", 4, 2);
    run<2>(d, "v_pk_add_f32", 4, 2);
    run<5>(d, "v_pk_mul_f32", 4, 2);
    run<3>(d, "v_sub_f32", 8, 1);
    run<4>(d, "v_min3_f32", 8, 1);
    return 0;
}
