// order.hip -- the source cloud put into Morton order on the device.
//
// The search kernels want neighbouring lanes to hold neighbouring queries (shared cache lines, similar
// trip counts), so the source is uploaded in Morton order and the outputs are un-permuted by the
// driver.  Round 1 ordered it on the host (keys + 3-pass LSD radix sort on 16 threads: 2.3 ms for
// 262,144 points, a third of a C4 upload); here the caller's f64 values go up as they are
// (24 bytes per point) and everything else happens on the GPU:
//   expand   (float)(x - c) per point                       (same rounding as the host packing)
//   bbox     grid.hip's order-preserving atomic min / max
//   keys     10 bits per axis over the bounding box, interleaved (30-bit Morton key)
//   sort     stable radix sort of (key, index)  -- rocPRIM through hipCUB, a plain library sort
//   gather   fp32 copy and, if wanted, the f64 copy {x - c, original index} in that order
// The order itself (int32 per point) is copied back for the un-permutation of the results.
#include "device_common.h"

#include <hipcub/hipcub.hpp>

namespace visma {

namespace {

__device__ __forceinline__ float ord2f(unsigned o)
{
    const unsigned u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(u);
}

__device__ __forceinline__ unsigned spread10(unsigned v)      // 10 bits -> every third bit
{
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ void order_expand_kernel(const double *__restrict__ xyz, int n, double cx, double cy, double cz,
                                    float4 *__restrict__ f4)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f4[i] = make_float4((float)(xyz[3ll * i] - cx), (float)(xyz[3ll * i + 1] - cy), (float)(xyz[3ll * i + 2] - cz), 0.f);
}

__global__ void order_keys_kernel(const float4 *__restrict__ f4, int n, const unsigned *__restrict__ box6,
                                  unsigned *__restrict__ keys, int *__restrict__ idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = f4[i];
    const float v[3] = {p.x, p.y, p.z};
    unsigned q[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float mn = ord2f(box6[a]), mx = ord2f(box6[3 + a]);
        const float ext = mx - mn;
        float t = ext > 0.f ? (v[a] - mn) / ext * 1024.0f : 0.f;
        t = fminf(fmaxf(t, 0.f), 1023.0f);                     // (NaN -> 0)
        q[a] = (unsigned)t;
    }
    keys[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    idx[i] = i;
}

__global__ void order_gather_kernel(const float4 *__restrict__ f4, const double *__restrict__ xyz, int n,
                                    const int *__restrict__ order, double cx, double cy, double cz,
                                    float4 *__restrict__ src, Pt64 *__restrict__ src64)
{
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n) return;
    const int i = order[pos];
    src[pos] = f4[i];
    if (src64) src64[pos] = Pt64{xyz[3ll * i] - cx, xyz[3ll * i + 1] - cy, xyz[3ll * i + 2] - cz, (unsigned long long)i};
}

constexpr size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

// scratch layout: f4[n] | keys[n] keys2[n] | idx[n] | box[8] | cub temp
size_t order_source_scratch_bytes(int64_t n)
{
    size_t tmp = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, (unsigned *)nullptr, (unsigned *)nullptr, (int *)nullptr,
                                            (int *)nullptr, (int)std::max<int64_t>(n, 1), 0, 30, (hipStream_t) nullptr);
    const size_t m = (size_t)std::max<int64_t>(n, 1);
    return align256(16 * m) + 2 * align256(4 * m) + align256(4 * m) + 256 + align256(tmp) + 256;
}

// d_xyz: the caller's points (3 doubles each) on the device; d_src (float4[n]), d_src64 (Pt64[n] or NULL) and
// d_order (int32[n]: original index of the point at each position) receive the Morton-ordered cloud
hipError_t order_source_device(const double *d_xyz, int64_t n, const double c[3], float4 *d_src, Pt64 *d_src64,
                               int32_t *d_order, void *scratch, size_t scratch_bytes, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    if (n > 0x7fffffff || scratch_bytes < order_source_scratch_bytes(n)) return hipErrorInvalidValue;
    char *p = (char *)scratch;
    const size_t m = (size_t)n;
    float4 *f4 = (float4 *)p; p += align256(16 * m);
    unsigned *keys = (unsigned *)p; p += align256(4 * m);
    unsigned *keys2 = (unsigned *)p; p += align256(4 * m);
    int *idx = (int *)p; p += align256(4 * m);
    unsigned *box = (unsigned *)p; p += 256;
    size_t tmp = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, keys, keys2, idx, (int *)d_order, (int)n, 0, 30, stream);
    if (e != hipSuccess) return e;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(order_expand_kernel, dim3(blocks), dim3(256), 0, stream, d_xyz, (int)n, c[0], c[1], c[2], f4);
    e = launch_grid_bbox(f4, n, box, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(order_keys_kernel, dim3(blocks), dim3(256), 0, stream, f4, (int)n, box, keys, idx);
    e = hipcub::DeviceRadixSort::SortPairs((void *)p, tmp, keys, keys2, idx, (int *)d_order, (int)n, 0, 30, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(order_gather_kernel, dim3(blocks), dim3(256), 0, stream, f4, d_xyz, (int)n, (const int *)d_order,
                       c[0], c[1], c[2], d_src, d_src64);
    return hipGetLastError();
}

// ---- exclusive prefix sum of n u32 values in ONE pass over the data (hipCUB, decoupled look-back) -------------
size_t exclusive_scan_u32_tmp_bytes(long long n)
{
    size_t tmp = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, (const unsigned *)nullptr, (unsigned *)nullptr,
                                          (int)std::min<long long>(std::max<long long>(n, 1), 0x7fffffff), (hipStream_t) nullptr);
    return tmp;
}

hipError_t launch_exclusive_scan_u32_lib(const unsigned *in, long long n, unsigned *out, void *tmp, size_t tmp_bytes,
                                         hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    if (n > 0x7fffffff) return hipErrorInvalidValue;
    return hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, in, out, (int)n, stream);
}

// ---- the clouds as uploaded -> the fp32 / f64 copies the searches read ----------------------------------
__global__ void promote_pt64_kernel(const float4 *__restrict__ src, Pt64 *__restrict__ dst, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 q = src[i];
    Pt64 o;
    o.x = (double)q.x; o.y = (double)q.y; o.z = (double)q.z;
    o.w = (unsigned long long)i;
    dst[i] = o;
}

// RAW = double: the caller's f64 values; RAW = float: values that fp32 holds exactly, widened back first
template <typename RAW>
__global__ void expand_raw_kernel(const RAW *__restrict__ xyz, long long n, long long index0, double cx, double cy, double cz,
                                  float4 *__restrict__ f4, Pt64 *__restrict__ p8)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    // the same f64 subtraction and rounding as the host packing (pack_f64_to / set_clouds_f64)
    const double x = (double)xyz[3 * j] - cx, y = (double)xyz[3 * j + 1] - cy, z = (double)xyz[3 * j + 2] - cz;
    f4[j] = make_float4((float)x, (float)y, (float)z, 0.f);
    if (p8) p8[j] = Pt64{x, y, z, (unsigned long long)(index0 + j)};
}

hipError_t launch_expand_f64(const double *xyz, int64_t n, const double c[3], float4 *f4, Pt64 *p8, hipStream_t stream,
                             int64_t index0)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(expand_raw_kernel<double>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, xyz, (long long)n,
                       (long long)index0, c[0], c[1], c[2], f4, p8);
    return hipGetLastError();
}

hipError_t launch_expand_f32(const float *xyz, int64_t n, int64_t index0, const double c[3], float4 *f4, Pt64 *p8,
                             hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(expand_raw_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, xyz, (long long)n,
                       (long long)index0, c[0], c[1], c[2], f4, p8);
    return hipGetLastError();
}

hipError_t launch_promote_pt64(const float4 *src, Pt64 *dst, int64_t n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(promote_pt64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, dst,
                       (long long)n);
    return hipGetLastError();
}

}  // namespace visma
