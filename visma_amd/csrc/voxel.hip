// voxel.hip -- open3d::VoxelDownSample on the GPU
// (O3D/Core/Geometry/DownSample.cpp:179-220), the step both callers run right
// before ICP (src/annotation.cpp:112, src/evaluation.cpp:258).
//
// The reference hashes voxel indices into an unordered_map and accumulates
// points in input order.  Here: f64 bounding box (atomic min/max on
// order-preserving keys) -> voxel index per point with the reference's exact
// f64 expression floor((p - (min - voxel/2)) / voxel) -> 64-bit key
// (ix*ny + iy)*nz + iz -> STABLE radix sort of (key, point index) (rocPRIM via
// hipCUB) -> run heads + exclusive scan -> one thread per voxel sums its run
// SEQUENTIALLY in point-index order.  Because the sort is stable the sums are
// taken in exactly the reference's order, so every output value is bit-identical
// to the reference's; only the ORDER of the output voxels differs (ascending key
// here, hash-map iteration order there).
#include "device_common.h"

#include <hipcub/hipcub.hpp>

namespace visma {

__device__ __forceinline__ unsigned long long d2ord(double d)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}

__global__ void vox_bbox_init_kernel(unsigned long long *box)
{
    if (threadIdx.x < 3) box[threadIdx.x] = ~0ull;
    else if (threadIdx.x < 6) box[threadIdx.x] = 0ull;
}

__global__ __launch_bounds__(256) void vox_bbox_kernel(const double *__restrict__ xyz, long long n,
                                                       unsigned long long *__restrict__ box)
{
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        for (int a = 0; a < 3; a++) {
            const double v = xyz[3 * i + a];
            // std::min_element semantics: comparisons with NaN are false
            if (v < mn[a]) mn[a] = v;
            if (v > mx[a]) mx[a] = v;
        }
    for (int a = 0; a < 3; a++) {
        for (int off = 32; off > 0; off >>= 1) {
            const double o1 = __shfl_down(mn[a], off, 64), o2 = __shfl_down(mx[a], off, 64);
            if (o1 < mn[a]) mn[a] = o1;
            if (o2 > mx[a]) mx[a] = o2;
        }
    }
    // six atomics per workgroup, not per wave (they all hit the same six words)
    __shared__ double wmn[4][3], wmx[4][3];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; a++) { wmn[w][a] = mn[a]; wmx[w][a] = mx[a]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        double lo = wmn[0][a], hi = wmx[0][a];
        for (int k = 1; k < 4; k++) {                       // same comparisons as above: NaN never replaces
            if (wmn[k][a] < lo) lo = wmn[k][a];
            if (wmx[k][a] > hi) hi = wmx[k][a];
        }
        atomicMin(&box[a], d2ord(lo));
        atomicMax(&box[3 + a], d2ord(hi));
    }
}

struct VoxParams {
    double vmin[3];
    double voxel;
    long long ny, nz;
};

__global__ __launch_bounds__(256) void vox_key_kernel(const double *__restrict__ xyz, long long n,
                                                      VoxParams vp,
                                                      unsigned long long *__restrict__ key,
                                                      unsigned *__restrict__ val)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    long long v[3];
    for (int a = 0; a < 3; a++)      // DownSample.cpp:201-204, same f64 expression
        v[a] = (long long)(int)floor((xyz[3 * i + a] - vp.vmin[a]) / vp.voxel);
    key[i] = (unsigned long long)((v[0] * vp.ny + v[1]) * vp.nz + v[2]);
    val[i] = (unsigned)i;
}

__global__ __launch_bounds__(256) void vox_head_kernel(const unsigned long long *__restrict__ key,
                                                       long long n, unsigned *__restrict__ head)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    head[i] = (i == 0 || key[i] != key[i - 1]) ? 1u : 0u;
}

// vid = exclusive scan of head; a head at sorted position i starts voxel vid[i]
__global__ __launch_bounds__(256) void vox_starts_kernel(const unsigned *__restrict__ head,
                                                         const unsigned *__restrict__ vid, long long n,
                                                         unsigned *__restrict__ vstart)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (head[i]) vstart[vid[i]] = (unsigned)i;
}

__global__ __launch_bounds__(256) void vox_reduce_kernel(
    const double *__restrict__ xyz, const double *__restrict__ nrm, const double *__restrict__ col,
    const unsigned *__restrict__ sorted_idx, const unsigned *__restrict__ vstart, long long nvox,
    long long n, double *__restrict__ out_xyz, double *__restrict__ out_nrm, double *__restrict__ out_col)
{
    const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
    if (v >= nvox) return;
    const long long b = vstart[v], e = (v + 1 < nvox) ? (long long)vstart[v + 1] : n;
    double p[3] = {0, 0, 0}, nn[3] = {0, 0, 0}, c[3] = {0, 0, 0};
    for (long long j = b; j < e; j++) {            // ascending point index (stable sort)
        const long long i = sorted_idx[j];
        p[0] += xyz[3 * i]; p[1] += xyz[3 * i + 1]; p[2] += xyz[3 * i + 2];
        if (nrm) {
            const double a0 = nrm[3 * i], a1 = nrm[3 * i + 1], a2 = nrm[3 * i + 2];
            if (!isnan(a0) && !isnan(a1) && !isnan(a2)) { nn[0] += a0; nn[1] += a1; nn[2] += a2; }
        }
        if (col) { c[0] += col[3 * i]; c[1] += col[3 * i + 1]; c[2] += col[3 * i + 2]; }
    }
    const double cnt = (double)(e - b);
    out_xyz[3 * v] = p[0] / cnt; out_xyz[3 * v + 1] = p[1] / cnt; out_xyz[3 * v + 2] = p[2] / cnt;
    if (nrm) {                                       // Eigen normalized(): v / sqrt(|v|^2) if |v|^2 > 0
        const double z = nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2];
        if (z > 0.0) { const double s = sqrt(z); nn[0] /= s; nn[1] /= s; nn[2] /= s; }
        out_nrm[3 * v] = nn[0]; out_nrm[3 * v + 1] = nn[1]; out_nrm[3 * v + 2] = nn[2];
    }
    if (col) { out_col[3 * v] = c[0] / cnt; out_col[3 * v + 1] = c[1] / cnt; out_col[3 * v + 2] = c[2] / cnt; }
}

// host-side decode of the order-preserving keys
static double ord2d(unsigned long long o)
{
    const unsigned long long u = (o & 0x8000000000000000ull) ? (o & 0x7FFFFFFFFFFFFFFFull) : ~o;
    double d;
    memcpy(&d, &u, sizeof(d));
    return d;
}

// Exclusive scan helpers live in grid.hip
hipError_t launch_exclusive_scan_u32(const unsigned *in, long long n, unsigned *bsum, unsigned *out,
                                     hipStream_t stream);

#define VOX_TRY(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { rc = e__; goto done; } } while (0)

// The device part: d_xyz (+ d_nrm, d_col or NULL), n points resident -> *d_oxyz (+ *d_onrm, *d_ocol) hipMalloc'ed
// here with *n_out voxels (NULL / 0 when the reference would return an empty cloud); the caller frees them.
hipError_t voxel_down_sample_core(const double *d_xyz, const double *d_nrm, const double *d_col, int64_t n, double voxel,
                                  double **d_oxyz_out, double **d_onrm_out, double **d_ocol_out, int64_t *n_out,
                                  int *too_fine, hipStream_t stream)
{
    *n_out = 0;
    *too_fine = 0;
    *d_oxyz_out = nullptr;
    if (d_onrm_out) *d_onrm_out = nullptr;
    if (d_ocol_out) *d_ocol_out = nullptr;
    if (!(voxel > 0.0) || n <= 0) return hipSuccess;       // DownSample.cpp:183-186
    hipError_t rc = hipSuccess;
    double *d_oxyz = nullptr, *d_onrm = nullptr, *d_ocol = nullptr;
    unsigned long long *d_box = nullptr, *d_key = nullptr, *d_key2 = nullptr;
    unsigned *d_val = nullptr, *d_val2 = nullptr, *d_head = nullptr, *d_vid = nullptr, *d_bsum = nullptr, *d_vstart = nullptr;
    void *d_tmp = nullptr;
    size_t tmp_bytes = 0;
    unsigned long long box[6];
    const int pb = (int)((n + 255) / 256);
    VoxParams vp;
    double mn[3], mx[3], ext = 0.0, dims[3], cells = 1.0;
    unsigned last_vid = 0, last_head = 0;
    int64_t nvox = 0;
    int end_bit = 64;
    bool keep = false;

    VOX_TRY(hipMalloc(&d_box, sizeof(unsigned long long) * 8));
    hipLaunchKernelGGL(vox_bbox_init_kernel, dim3(1), dim3(64), 0, stream, d_box);
    hipLaunchKernelGGL(vox_bbox_kernel, dim3(pb > 1024 ? 1024 : pb), dim3(256), 0, stream, d_xyz, (long long)n, d_box);
    VOX_TRY(hipMemcpyAsync(box, d_box, sizeof(box), hipMemcpyDeviceToHost, stream));
    VOX_TRY(hipStreamSynchronize(stream));
    for (int a = 0; a < 3; a++) { mn[a] = ord2d(box[a]); mx[a] = ord2d(box[3 + a]); }
    for (int a = 0; a < 3; a++) {                              // :189-190
        vp.vmin[a] = mn[a] - voxel * 0.5;
        const double e = (mx[a] + voxel * 0.5) - vp.vmin[a];
        if (e > ext) ext = e;
    }
    if (voxel * 2147483647.0 < ext) goto done;                 // :191-195 -> empty cloud
    for (int a = 0; a < 3; a++) {
        dims[a] = floor(((mx[a] - vp.vmin[a]) / voxel)) + 2.0;
        cells *= dims[a];
    }
    if (!(cells < 4.0e18)) { *too_fine = 1; goto done; }
    vp.voxel = voxel;
    vp.ny = (long long)dims[1];
    vp.nz = (long long)dims[2];
    end_bit = 1;
    while (end_bit < 64 && ldexp(1.0, end_bit) < cells) end_bit++;

    VOX_TRY(hipMalloc(&d_key, sizeof(unsigned long long) * n));
    VOX_TRY(hipMalloc(&d_key2, sizeof(unsigned long long) * n));
    VOX_TRY(hipMalloc(&d_val, sizeof(unsigned) * n));
    VOX_TRY(hipMalloc(&d_val2, sizeof(unsigned) * n));
    hipLaunchKernelGGL(vox_key_kernel, dim3(pb), dim3(256), 0, stream, d_xyz, (long long)n, vp, d_key, d_val);
    VOX_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_key, d_key2, d_val, d_val2, (int)n, 0, end_bit, stream));
    VOX_TRY(hipMalloc(&d_tmp, tmp_bytes > 0 ? tmp_bytes : 16));
    VOX_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_key, d_key2, d_val, d_val2, (int)n, 0, end_bit, stream));
    VOX_TRY(hipMalloc(&d_head, sizeof(unsigned) * (n + 1)));
    VOX_TRY(hipMalloc(&d_vid, sizeof(unsigned) * (n + 1)));
    VOX_TRY(hipMalloc(&d_bsum, sizeof(unsigned) * (n / 2048 + 2)));
    hipLaunchKernelGGL(vox_head_kernel, dim3(pb), dim3(256), 0, stream, d_key2, (long long)n, d_head);
    VOX_TRY(launch_exclusive_scan_u32(d_head, (long long)n, d_bsum, d_vid, stream));
    VOX_TRY(hipMemcpyAsync(&last_vid, d_vid + (n - 1), sizeof(unsigned), hipMemcpyDeviceToHost, stream));
    VOX_TRY(hipMemcpyAsync(&last_head, d_head + (n - 1), sizeof(unsigned), hipMemcpyDeviceToHost, stream));
    VOX_TRY(hipStreamSynchronize(stream));
    nvox = (int64_t)last_vid + (int64_t)last_head;
    VOX_TRY(hipMalloc(&d_vstart, sizeof(unsigned) * (nvox + 1)));
    hipLaunchKernelGGL(vox_starts_kernel, dim3(pb), dim3(256), 0, stream, d_head, d_vid, (long long)n, d_vstart);
    VOX_TRY(hipMalloc(&d_oxyz, sizeof(double) * 3 * nvox));
    if (d_nrm && d_onrm_out) VOX_TRY(hipMalloc(&d_onrm, sizeof(double) * 3 * nvox));
    if (d_col && d_ocol_out) VOX_TRY(hipMalloc(&d_ocol, sizeof(double) * 3 * nvox));
    hipLaunchKernelGGL(vox_reduce_kernel, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, stream, d_xyz,
                       d_onrm ? d_nrm : nullptr, d_ocol ? d_col : nullptr, d_val2, d_vstart, (long long)nvox, (long long)n,
                       d_oxyz, d_onrm, d_ocol);
    VOX_TRY(hipGetLastError());
    VOX_TRY(hipStreamSynchronize(stream));                     // (the scratch below is freed on return)
    *n_out = nvox;
    *d_oxyz_out = d_oxyz;
    if (d_onrm_out) *d_onrm_out = d_onrm;
    if (d_ocol_out) *d_ocol_out = d_ocol;
    keep = true;
done:
    if (!keep) { (void)hipFree(d_oxyz); (void)hipFree(d_onrm); (void)hipFree(d_ocol); }
    (void)hipFree(d_box); (void)hipFree(d_key); (void)hipFree(d_key2); (void)hipFree(d_val);
    (void)hipFree(d_val2); (void)hipFree(d_head); (void)hipFree(d_vid); (void)hipFree(d_bsum); (void)hipFree(d_vstart);
    (void)hipFree(d_tmp);
    return rc;
}

// Host arrays in, host arrays out.  Returns hipSuccess and *n_out (0 when the reference would return an empty
// cloud); *too_fine is set when the voxel grid cannot be keyed in 62 bits.
hipError_t voxel_down_sample_device(const double *h_xyz, const double *h_nrm, const double *h_col,
                                    int64_t n, double voxel, double *h_out_xyz, double *h_out_nrm,
                                    double *h_out_col, int64_t *n_out, int *too_fine,
                                    hipStream_t stream)
{
    *n_out = 0;
    *too_fine = 0;
    if (!(voxel > 0.0) || n <= 0) return hipSuccess;       // DownSample.cpp:183-186
    hipError_t rc = hipSuccess;
    double *d_xyz = nullptr, *d_nrm = nullptr, *d_col = nullptr, *d_oxyz = nullptr, *d_onrm = nullptr, *d_ocol = nullptr;
    int64_t nvox = 0;
    VOX_TRY(hipMalloc(&d_xyz, sizeof(double) * 3 * n));
    VOX_TRY(hipMemcpyAsync(d_xyz, h_xyz, sizeof(double) * 3 * n, hipMemcpyHostToDevice, stream));
    if (h_nrm) { VOX_TRY(hipMalloc(&d_nrm, sizeof(double) * 3 * n)); VOX_TRY(hipMemcpyAsync(d_nrm, h_nrm, sizeof(double) * 3 * n, hipMemcpyHostToDevice, stream)); }
    if (h_col) { VOX_TRY(hipMalloc(&d_col, sizeof(double) * 3 * n)); VOX_TRY(hipMemcpyAsync(d_col, h_col, sizeof(double) * 3 * n, hipMemcpyHostToDevice, stream)); }
    VOX_TRY(voxel_down_sample_core(d_xyz, d_nrm, d_col, n, voxel, &d_oxyz, &d_onrm, &d_ocol, &nvox, too_fine, stream));
    if (nvox > 0) {
        VOX_TRY(hipMemcpyAsync(h_out_xyz, d_oxyz, sizeof(double) * 3 * nvox, hipMemcpyDeviceToHost, stream));
        if (h_nrm && h_out_nrm) VOX_TRY(hipMemcpyAsync(h_out_nrm, d_onrm, sizeof(double) * 3 * nvox, hipMemcpyDeviceToHost, stream));
        if (h_col && h_out_col) VOX_TRY(hipMemcpyAsync(h_out_col, d_ocol, sizeof(double) * 3 * nvox, hipMemcpyDeviceToHost, stream));
        VOX_TRY(hipStreamSynchronize(stream));
    }
    *n_out = nvox;
done:
    (void)hipFree(d_xyz); (void)hipFree(d_nrm); (void)hipFree(d_col); (void)hipFree(d_oxyz); (void)hipFree(d_onrm);
    (void)hipFree(d_ocol);
    return rc;
}

// The centroid of n device-resident points as centroid_f64 (driver.cpp) computes it on the host: sequential f64
// sums over fixed chunks of `chunk` points, the chunk sums added in chunk order, divided by n -- the same value,
// bit for bit (no contraction on either side).  part: ceil(n / chunk) * 3 doubles of scratch; out: 3 doubles.
__global__ void centroid_chunks_kernel(const double *__restrict__ xyz, long long n, long long chunk, double *__restrict__ part)
{
    const long long ch = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long lo = ch * chunk;
    if (lo >= n) return;
    const long long hi = lo + chunk < n ? lo + chunk : n;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (long long i = lo; i < hi; i++) { s0 += xyz[3 * i]; s1 += xyz[3 * i + 1]; s2 += xyz[3 * i + 2]; }
    part[3 * ch] = s0; part[3 * ch + 1] = s1; part[3 * ch + 2] = s2;
}
__global__ void centroid_combine_kernel(const double *__restrict__ part, long long nch, long long n, double *__restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double c0 = 0.0, c1 = 0.0, c2 = 0.0;
    for (long long ch = 0; ch < nch; ch++) { c0 += part[3 * ch]; c1 += part[3 * ch + 1]; c2 += part[3 * ch + 2]; }
    out[0] = c0 / (double)n; out[1] = c1 / (double)n; out[2] = c2 / (double)n;
}
hipError_t centroid_device(const double *d_xyz, int64_t n, int64_t chunk, double *d_part, double *d_out, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    const long long nch = (n + chunk - 1) / chunk;
    hipLaunchKernelGGL(centroid_chunks_kernel, dim3((unsigned)((nch + 63) / 64)), dim3(64), 0, stream, d_xyz, (long long)n,
                       (long long)chunk, d_part);
    hipLaunchKernelGGL(centroid_combine_kernel, dim3(1), dim3(64), 0, stream, d_part, nch, (long long)n, d_out);
    return hipGetLastError();
}

}  // namespace visma
