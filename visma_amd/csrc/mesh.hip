// mesh.hip -- the mesh steps either side of ICP:
//
//  sample_mesh_kernel   feh::SamplePointCloudFromMesh (include/geometry.h:29-64):
//                       area-weighted surface sampling, the step that produces the
//                       ICP source (src/evaluation.cpp:252, src/annotation.cpp:126).
//  point_mesh_kernel    point -> triangle-mesh squared distance, the body of
//                       feh::MeasureSurfaceError (include/geometry.h:117-141, which
//                       uses igl::AABB::squared_distance): brute force over the
//                       faces, LDS-tiled, f64, same closest-point algorithm
//                       (Ericson, Real-Time Collision Detection 5.1.5).
//
// Everything is f64: these steps run once per evaluation, their results are
// compared value for value with the reference.
#include "device_common.h"

#include <math.h>
#include <vector>

namespace visma {

// ---------------------------------------------------------------------------
// point -> mesh distance
// ---------------------------------------------------------------------------
constexpr int kTriTile = 128;       // triangles staged per LDS fill (9 doubles each)
constexpr int kMeshBlock = 128;

__device__ __forceinline__ double dot3d(const double *a, const double *b)
{
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

// Closest point on triangle (a,b,c) to p; returns the squared distance.  Same
// region order and expressions as the oracle (vo_point_triangle_sqdist).
__device__ __forceinline__ double point_triangle(const double p[3], const double *a, const double *b,
                                                 const double *c, double q[3])
{
    double ab[3], ac[3], ap[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; ap[i] = p[i] - a[i]; }
    const double d1 = dot3d(ab, ap), d2 = dot3d(ac, ap);
    bool done = false;
    if (d1 <= 0.0 && d2 <= 0.0) { q[0] = a[0]; q[1] = a[1]; q[2] = a[2]; done = true; }          // vertex a
    if (!done) {
        double bp[3];
#pragma unroll
        for (int i = 0; i < 3; i++) bp[i] = p[i] - b[i];
        const double d3 = dot3d(ab, bp), d4 = dot3d(ac, bp);
        if (d3 >= 0.0 && d4 <= d3) { q[0] = b[0]; q[1] = b[1]; q[2] = b[2]; done = true; }      // vertex b
        if (!done) {
            const double vc = d1 * d4 - d3 * d2;
            if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {                                           // edge ab
                const double v = d1 / (d1 - d3);
#pragma unroll
                for (int i = 0; i < 3; i++) q[i] = a[i] + v * ab[i];
                done = true;
            }
            if (!done) {
                double cp[3];
#pragma unroll
                for (int i = 0; i < 3; i++) cp[i] = p[i] - c[i];
                const double d5 = dot3d(ab, cp), d6 = dot3d(ac, cp);
                if (d6 >= 0.0 && d5 <= d6) { q[0] = c[0]; q[1] = c[1]; q[2] = c[2]; done = true; }  // vertex c
                if (!done) {
                    const double vb = d5 * d2 - d1 * d6;
                    if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {                                   // edge ac
                        const double w = d2 / (d2 - d6);
#pragma unroll
                        for (int i = 0; i < 3; i++) q[i] = a[i] + w * ac[i];
                        done = true;
                    }
                    if (!done) {
                        const double va = d3 * d6 - d5 * d4;
                        if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {                 // edge bc
                            const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
#pragma unroll
                            for (int i = 0; i < 3; i++) q[i] = b[i] + w * (c[i] - b[i]);
                        } else {                                                                 // interior
                            const double denom = 1.0 / (va + vb + vc);
                            const double v = vb * denom, w = vc * denom;
#pragma unroll
                            for (int i = 0; i < 3; i++) q[i] = a[i] + ab[i] * v + ac[i] * w;
                        }
                    }
                }
            }
        }
    }
    const double dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
    return dx * dx + dy * dy + dz * dz;
}

__global__ __launch_bounds__(kMeshBlock) void point_mesh_kernel(
    const double *__restrict__ P, long long np, const double *__restrict__ tri /* nf x 9 */,
    long long nf, double *__restrict__ d2_out, int *__restrict__ face_out,
    double *__restrict__ closest_out)
{
    __shared__ double lds[kTriTile * 9];
    const long long i = (long long)blockIdx.x * kMeshBlock + threadIdx.x;
    double p[3] = {0, 0, 0};
    if (i < np) { p[0] = P[3 * i]; p[1] = P[3 * i + 1]; p[2] = P[3 * i + 2]; }
    double best = INFINITY, bq[3] = {0, 0, 0};
    int bf = -1;
    for (long long f0 = 0; f0 < nf; f0 += kTriTile) {
        const int cnt = (int)min((long long)kTriTile, nf - f0);
        for (int k = threadIdx.x; k < cnt * 9; k += kMeshBlock) lds[k] = tri[f0 * 9 + k];
        __syncthreads();
        for (int t = 0; t < cnt; t++) {
            const double *a = &lds[t * 9];
            double q[3];
            const double d = point_triangle(p, a, a + 3, a + 6, q);
            if (d < best) { best = d; bf = (int)(f0 + t); bq[0] = q[0]; bq[1] = q[1]; bq[2] = q[2]; }
        }
        __syncthreads();
    }
    if (i < np) {
        d2_out[i] = best;
        if (face_out) face_out[i] = bf;
        if (closest_out) { closest_out[3 * i] = bq[0]; closest_out[3 * i + 1] = bq[1]; closest_out[3 * i + 2] = bq[2]; }
    }
}

#define MESH_TRY(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { rc = e__; goto done; } } while (0)

hipError_t point_mesh_distance_device(const double *h_P, int64_t np, const double *h_V, int64_t nv,
                                      const int32_t *h_F, int64_t nf, double *h_d2, int32_t *h_face,
                                      double *h_closest, float *kernel_ms, hipStream_t stream)
{
    hipError_t rc = hipSuccess;
    if (np <= 0) return hipSuccess;
    double *d_P = nullptr, *d_tri = nullptr, *d_d2 = nullptr, *d_cl = nullptr;
    int *d_face = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::vector<double> tri((size_t)(nf > 0 ? nf : 1) * 9);
    for (int64_t f = 0; f < nf; f++)
        for (int c = 0; c < 3; c++) {
            const int64_t v = h_F[3 * f + c];
            if (v < 0 || v >= nv) return hipErrorInvalidValue;
            for (int a = 0; a < 3; a++) tri[(size_t)f * 9 + c * 3 + a] = h_V[3 * v + a];
        }
    MESH_TRY(hipMalloc(&d_P, sizeof(double) * 3 * np));
    MESH_TRY(hipMalloc(&d_tri, sizeof(double) * 9 * (nf > 0 ? nf : 1)));
    MESH_TRY(hipMalloc(&d_d2, sizeof(double) * np));
    MESH_TRY(hipMalloc(&d_face, sizeof(int) * np));
    MESH_TRY(hipMalloc(&d_cl, sizeof(double) * 3 * np));
    MESH_TRY(hipMemcpyAsync(d_P, h_P, sizeof(double) * 3 * np, hipMemcpyHostToDevice, stream));
    if (nf > 0) MESH_TRY(hipMemcpyAsync(d_tri, tri.data(), sizeof(double) * 9 * nf, hipMemcpyHostToDevice, stream));
    MESH_TRY(hipEventCreate(&e0));
    MESH_TRY(hipEventCreate(&e1));
    MESH_TRY(hipEventRecord(e0, stream));
    hipLaunchKernelGGL(point_mesh_kernel, dim3((unsigned)((np + kMeshBlock - 1) / kMeshBlock)), dim3(kMeshBlock), 0,
                       stream, d_P, (long long)np, d_tri, (long long)nf, d_d2, d_face, d_cl);
    MESH_TRY(hipGetLastError());
    MESH_TRY(hipEventRecord(e1, stream));
    MESH_TRY(hipMemcpyAsync(h_d2, d_d2, sizeof(double) * np, hipMemcpyDeviceToHost, stream));
    if (h_face) MESH_TRY(hipMemcpyAsync(h_face, d_face, sizeof(int) * np, hipMemcpyDeviceToHost, stream));
    if (h_closest) MESH_TRY(hipMemcpyAsync(h_closest, d_cl, sizeof(double) * 3 * np, hipMemcpyDeviceToHost, stream));
    MESH_TRY(hipStreamSynchronize(stream));
    if (kernel_ms) MESH_TRY(hipEventElapsedTime(kernel_ms, e0, e1));
done:
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d_P); (void)hipFree(d_tri); (void)hipFree(d_d2); (void)hipFree(d_face); (void)hipFree(d_cl);
    return rc;
}

// ---------------------------------------------------------------------------
// mesh sampling
// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11): counter-based, so sample i's uniforms
// depend on (seed, i) only.
__device__ __forceinline__ void philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                           unsigned k0, unsigned k1, unsigned out[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double u53(unsigned hi, unsigned lo)
{
    const unsigned long long v = (((unsigned long long)hi << 32) | lo) >> 11;     // 53 bits
    return (double)v * (1.0 / 9007199254740992.0);                                 // [0, 1)
}

__global__ __launch_bounds__(256) void sample_mesh_kernel(
    const double *__restrict__ V, const int *__restrict__ F, const double *__restrict__ cdf,
    long long nf, long long n, int quirks, unsigned long long seed,
    const double *__restrict__ uniforms, double *__restrict__ pts, unsigned *__restrict__ valid)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double r, a, b;
    if (uniforms) {
        r = uniforms[3 * i]; a = uniforms[3 * i + 1]; b = uniforms[3 * i + 2];
    } else {
        unsigned w0[4], w1[4];
        philox4x32((unsigned)i, (unsigned)(i >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), w0);
        philox4x32((unsigned)i, (unsigned)(i >> 32), 1u, 0u, (unsigned)seed, (unsigned)(seed >> 32), w1);
        r = u53(w0[0], w0[1]); a = u53(w0[2], w0[3]); b = u53(w1[0], w1[1]);
    }
    // first index whose cdf exceeds r
    long long lo = 0, hi = nf;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (cdf[mid] > r) hi = mid; else lo = mid + 1;
    }
    long long k;
    if (quirks) {
        // geometry.h:53-54 picks k for r in [cdf[k], cdf[k+1]), k <= nf-2: one face late,
        // never the last face, nothing at all for r < cdf[0]
        k = (lo >= 1 && lo <= nf - 1) ? lo - 1 : -1;
    } else {
        k = lo < nf ? lo : nf - 1;
        if (a + b > 1.0) { a = 1.0 - a; b = 1.0 - b; }      // fold the parallelogram onto the triangle
    }
    valid[i] = k >= 0 ? 1u : 0u;
    if (k < 0) return;
    const double *v0 = V + 3 * (long long)F[3 * k], *v1 = V + 3 * (long long)F[3 * k + 1],
                 *v2 = V + 3 * (long long)F[3 * k + 2];
#pragma unroll
    for (int d = 0; d < 3; d++) pts[3 * i + d] = v0[d] + a * (v1[d] - v0[d]) + b * (v2[d] - v0[d]);
}

__global__ __launch_bounds__(256) void compact_points_kernel(const double *__restrict__ pts,
                                                             const unsigned *__restrict__ valid,
                                                             const unsigned *__restrict__ pos, long long n,
                                                             double *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !valid[i]) return;
    const long long o = pos[i];
    out[3 * o] = pts[3 * i]; out[3 * o + 1] = pts[3 * i + 1]; out[3 * o + 2] = pts[3 * i + 2];
}

hipError_t launch_exclusive_scan_u32(const unsigned *in, long long n, unsigned *bsum, unsigned *out,
                                     hipStream_t stream);

hipError_t sample_mesh_device(const double *h_V, int64_t nv, const int32_t *h_F, int64_t nf, int64_t n,
                              int quirks, unsigned long long seed, const double *h_uniforms,
                              double *h_out, int64_t *n_out, hipStream_t stream)
{
    *n_out = 0;
    if (nf <= 0 || n <= 0) return hipSuccess;
    hipError_t rc = hipSuccess;
    // The cumulative area table is built on the host with the reference's own
    // sequential f64 arithmetic (geometry.h:33-43) so that face selection is identical.
    std::vector<double> cdf((size_t)nf);
    double total = 0.0;
    for (int64_t i = 0; i < nf; i++) {
        for (int c = 0; c < 3; c++)
            if (h_F[3 * i + c] < 0 || h_F[3 * i + c] >= nv) return hipErrorInvalidValue;
        const double *a = h_V + 3 * (int64_t)h_F[3 * i], *b = h_V + 3 * (int64_t)h_F[3 * i + 1],
                     *c = h_V + 3 * (int64_t)h_F[3 * i + 2];
        const double e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
        const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2],
                     cz = e1[0] * e2[1] - e1[1] * e2[0];
        cdf[i] = 0.5 * sqrt(cx * cx + cy * cy + cz * cz);
        total += cdf[i];
    }
    cdf[0] /= total;
    for (int64_t i = 1; i < nf; i++) cdf[i] = cdf[i - 1] + cdf[i] / total;

    double *d_V = nullptr, *d_cdf = nullptr, *d_u = nullptr, *d_pts = nullptr, *d_out = nullptr;
    int *d_F = nullptr;
    unsigned *d_valid = nullptr, *d_pos = nullptr, *d_bsum = nullptr;
    unsigned last_pos = 0, last_valid = 0;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    int64_t m = 0;
    MESH_TRY(hipMalloc(&d_V, sizeof(double) * 3 * nv));
    MESH_TRY(hipMalloc(&d_F, sizeof(int) * 3 * nf));
    MESH_TRY(hipMalloc(&d_cdf, sizeof(double) * nf));
    MESH_TRY(hipMalloc(&d_pts, sizeof(double) * 3 * n));
    MESH_TRY(hipMalloc(&d_out, sizeof(double) * 3 * n));
    MESH_TRY(hipMalloc(&d_valid, sizeof(unsigned) * (n + 1)));
    MESH_TRY(hipMalloc(&d_pos, sizeof(unsigned) * (n + 1)));
    MESH_TRY(hipMalloc(&d_bsum, sizeof(unsigned) * (n / 2048 + 2)));
    MESH_TRY(hipMemcpyAsync(d_V, h_V, sizeof(double) * 3 * nv, hipMemcpyHostToDevice, stream));
    MESH_TRY(hipMemcpyAsync(d_F, h_F, sizeof(int) * 3 * nf, hipMemcpyHostToDevice, stream));
    MESH_TRY(hipMemcpyAsync(d_cdf, cdf.data(), sizeof(double) * nf, hipMemcpyHostToDevice, stream));
    if (h_uniforms) {
        MESH_TRY(hipMalloc(&d_u, sizeof(double) * 3 * n));
        MESH_TRY(hipMemcpyAsync(d_u, h_uniforms, sizeof(double) * 3 * n, hipMemcpyHostToDevice, stream));
    }
    hipLaunchKernelGGL(sample_mesh_kernel, dim3(blocks), dim3(256), 0, stream, d_V, d_F, d_cdf, (long long)nf,
                       (long long)n, quirks, seed, d_u, d_pts, d_valid);
    MESH_TRY(launch_exclusive_scan_u32(d_valid, (long long)n, d_bsum, d_pos, stream));
    hipLaunchKernelGGL(compact_points_kernel, dim3(blocks), dim3(256), 0, stream, d_pts, d_valid, d_pos,
                       (long long)n, d_out);
    MESH_TRY(hipGetLastError());
    MESH_TRY(hipMemcpyAsync(&last_pos, d_pos + (n - 1), sizeof(unsigned), hipMemcpyDeviceToHost, stream));
    MESH_TRY(hipMemcpyAsync(&last_valid, d_valid + (n - 1), sizeof(unsigned), hipMemcpyDeviceToHost, stream));
    MESH_TRY(hipStreamSynchronize(stream));
    m = (int64_t)last_pos + (int64_t)last_valid;
    if (m > 0) MESH_TRY(hipMemcpy(h_out, d_out, sizeof(double) * 3 * m, hipMemcpyDeviceToHost));
    *n_out = m;
done:
    (void)hipFree(d_V); (void)hipFree(d_F); (void)hipFree(d_cdf); (void)hipFree(d_u); (void)hipFree(d_pts);
    (void)hipFree(d_out); (void)hipFree(d_valid); (void)hipFree(d_pos); (void)hipFree(d_bsum);
    return rc;
}

}  // namespace visma
