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

#include <hipcub/hipcub.hpp>

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

// ---------------------------------------------------------------------------
// BVH over the triangles (exact: prunes only what cannot win)
// ---------------------------------------------------------------------------
// Triangles are sorted by the 30-bit Morton code of their centroid (stable radix
// sort, ties by face index); kLeaf consecutive triangles form a leaf; above the
// leaves sits an IMPLICIT complete binary tree in heap order (node i has children
// 2i, 2i+1; the leaf level starts at index P = the leaf count rounded up to a
// power of two; missing leaves carry an inverted box).  A query walks it depth
// first, nearer child first, with a per-thread stack in LDS, and skips a node only
// when the box's lower bound is strictly above the best squared distance so far.
// Leaf boxes are inflated by 2^-40 of the mesh extent, which keeps every computed
// closest point inside its box, so the bound (monotone floating-point arithmetic,
// same x,y,z summation order as the distance itself) never exceeds a computed
// distance: the result is the same minimum, face for face, as the brute-force scan.
constexpr int kLeaf = 4;
constexpr int kBvhBlock = 64;

struct MortonParams {
    double lo[3], inv[3];            // code = min(1023, (c - lo) * inv)
};

__device__ __forceinline__ unsigned spread10(unsigned v)
{
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(256) void tri_code_kernel(const double *__restrict__ V, const int *__restrict__ F,
                                                       long long nf, MortonParams mp,
                                                       unsigned *__restrict__ key, unsigned *__restrict__ val)
{
    const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
    if (f >= nf) return;
    unsigned code = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double c = (V[3 * (long long)F[3 * f] + a] + V[3 * (long long)F[3 * f + 1] + a] +
                          V[3 * (long long)F[3 * f + 2] + a]) * (1.0 / 3.0);
        double q = (c - mp.lo[a]) * mp.inv[a];
        q = q >= 0.0 ? q : 0.0;                       // also catches NaN
        const unsigned qi = q < 1023.0 ? (unsigned)q : 1023u;
        code |= spread10(qi) << (2 - a);
    }
    key[f] = code;
    val[f] = (unsigned)f;
}

// gather the triangles in sorted order (or in face order when `order` is null)
__global__ __launch_bounds__(256) void tri_gather_kernel(const double *__restrict__ V, const int *__restrict__ F,
                                                         const unsigned *__restrict__ order, long long nf,
                                                         double *__restrict__ tri)
{
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j >= nf) return;
    const long long f = order ? (long long)order[j] : j;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const long long v = F[3 * f + c];
#pragma unroll
        for (int a = 0; a < 3; a++) tri[9 * j + 3 * c + a] = V[3 * v + a];
    }
}

__global__ __launch_bounds__(256) void bvh_leaf_kernel(const double *__restrict__ tri, long long nf, long long P,
                                                       double inflate, double *__restrict__ nodes)
{
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= P) return;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (long long t = g * kLeaf; t < (g + 1) * kLeaf && t < nf; t++)
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const double v = tri[9 * t + 3 * c + a];
                lo[a] = fmin(lo[a], v);
                hi[a] = fmax(hi[a], v);
            }
    double *n = nodes + 6 * (P + g);
#pragma unroll
    for (int a = 0; a < 3; a++) { n[a] = lo[a] - inflate; n[3 + a] = hi[a] + inflate; }
}

__global__ __launch_bounds__(256) void bvh_level_kernel(long long first, long long count, double *__restrict__ nodes)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const long long n = first + i;
    const double *l = nodes + 6 * (2 * n), *r = l + 6;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        nodes[6 * n + a] = fmin(l[a], r[a]);
        nodes[6 * n + 3 + a] = fmax(l[3 + a], r[3 + a]);
    }
}

__device__ __forceinline__ double box_lower_bound(const double *__restrict__ n, const double p[3])
{
    double s = 0.0;
    double t[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double below = n[a] - p[a], above = p[a] - n[3 + a];
        t[a] = fmax(0.0, fmax(below, above));        // +inf for an empty (inverted) box
    }
    s = t[0] * t[0] + t[1] * t[1] + t[2] * t[2];
    return s;
}

__device__ __forceinline__ float round_down_f32(double x)
{
    float f = (float)x;
    if ((double)f > x) f = nextafterf(f, -INFINITY);
    return f;
}

// Dynamic LDS: `depth` stack levels of (node u32, lower bound f32 rounded DOWN) per thread.
__global__ __launch_bounds__(kBvhBlock) void point_mesh_bvh_kernel(
    const double *__restrict__ Pq, long long np, const double *__restrict__ tri, const unsigned *__restrict__ faceid,
    const double *__restrict__ nodes, long long nf, long long P, int depth, const unsigned *__restrict__ order,
    double *__restrict__ d2_out, int *__restrict__ face_out, double *__restrict__ closest_out)
{
    extern __shared__ unsigned bvh_smem[];
    unsigned *stk_node = bvh_smem;
    float *stk_lb = (float *)(bvh_smem + depth * kBvhBlock);
    const long long slot = (long long)blockIdx.x * kBvhBlock + threadIdx.x;
    if (slot >= np) return;
    // queries are walked in Morton order (neighbouring lanes take the same path
    // through the tree); results go back to the caller's positions
    const long long i = order ? (long long)order[slot] : slot;
    const double p[3] = {Pq[3 * i], Pq[3 * i + 1], Pq[3 * i + 2]};
    double best = INFINITY, bq[3] = {0, 0, 0};
    int bf = -1;
    int sp = 1;
    stk_node[threadIdx.x] = 1u;
    stk_lb[threadIdx.x] = 0.f;
    while (sp > 0) {
        --sp;
        const unsigned node = stk_node[sp * kBvhBlock + threadIdx.x];
        if ((double)stk_lb[sp * kBvhBlock + threadIdx.x] > best) continue;
        if ((long long)node >= P) {
            const long long t0 = ((long long)node - P) * kLeaf;
            for (long long t = t0; t < t0 + kLeaf && t < nf; t++) {
                const double *a = tri + 9 * t;
                double q[3];
                const double d = point_triangle(p, a, a + 3, a + 6, q);
                const int f = (int)faceid[t];
                if (d < best || (d == best && f < bf)) { best = d; bf = f; bq[0] = q[0]; bq[1] = q[1]; bq[2] = q[2]; }
            }
        } else {
            const unsigned c0 = 2u * node, c1 = c0 + 1u;
            const double l0 = box_lower_bound(nodes + 6 * (long long)c0, p);
            const double l1 = box_lower_bound(nodes + 6 * (long long)c1, p);
            const bool first0 = l0 <= l1;
            const double lfar = first0 ? l1 : l0, lnear = first0 ? l0 : l1;
            if (!(lfar > best)) {                    // farther child below the nearer one
                stk_node[sp * kBvhBlock + threadIdx.x] = first0 ? c1 : c0;
                stk_lb[sp * kBvhBlock + threadIdx.x] = round_down_f32(lfar);
                ++sp;
            }
            if (!(lnear > best)) {
                stk_node[sp * kBvhBlock + threadIdx.x] = first0 ? c0 : c1;
                stk_lb[sp * kBvhBlock + threadIdx.x] = round_down_f32(lnear);
                ++sp;
            }
        }
    }
    d2_out[i] = best;
    if (face_out) face_out[i] = bf;
    if (closest_out) { closest_out[3 * i] = bq[0]; closest_out[3 * i + 1] = bq[1]; closest_out[3 * i + 2] = bq[2]; }
}

__global__ __launch_bounds__(256) void point_code_kernel(const double *__restrict__ Pq, long long np, MortonParams mp,
                                                         unsigned *__restrict__ key, unsigned *__restrict__ val)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= np) return;
    unsigned code = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        double q = (Pq[3 * i + a] - mp.lo[a]) * mp.inv[a];
        q = q >= 0.0 ? q : 0.0;
        const unsigned qi = q < 1023.0 ? (unsigned)q : 1023u;
        code |= spread10(qi) << (2 - a);
    }
    key[i] = code;
    val[i] = (unsigned)i;
}

__global__ __launch_bounds__(256) void sqrt_kernel(double *__restrict__ d, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = sqrt(d[i]);                     // correctly rounded, as std::sqrt / Eigen cwiseSqrt
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

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
namespace {

struct DBuf {                                       // device allocation, freed on scope exit
    void *p = nullptr;
    ~DBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <typename T> T *as() const { return (T *)p; }
};

#define MESH_TRY(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) return e__; } while (0)

struct DevMesh {
    DBuf V, F;
    int64_t nv = 0, nf = 0;
    double lo[3], hi[3];                             // vertex bounding box (host)
};

hipError_t upload_mesh(const double *h_V, int64_t nv, const int32_t *h_F, int64_t nf, DevMesh &m, hipStream_t stream)
{
    for (int64_t i = 0; i < 3 * nf; i++)
        if (h_F[i] < 0 || h_F[i] >= nv) return hipErrorInvalidValue;
    for (int a = 0; a < 3; a++) { m.lo[a] = INFINITY; m.hi[a] = -INFINITY; }
    for (int64_t i = 0; i < nv; i++)
        for (int a = 0; a < 3; a++) {
            const double v = h_V[3 * i + a];
            if (v < m.lo[a]) m.lo[a] = v;
            if (v > m.hi[a]) m.hi[a] = v;
        }
    m.nv = nv; m.nf = nf;
    MESH_TRY(m.V.alloc(sizeof(double) * 3 * nv));
    MESH_TRY(m.F.alloc(sizeof(int) * 3 * nf));
    if (nv > 0) MESH_TRY(hipMemcpyAsync(m.V.p, h_V, sizeof(double) * 3 * nv, hipMemcpyHostToDevice, stream));
    if (nf > 0) MESH_TRY(hipMemcpyAsync(m.F.p, h_F, sizeof(int) * 3 * nf, hipMemcpyHostToDevice, stream));
    return hipSuccess;
}

struct DevBvh {
    DBuf tri, faceid, nodes;
    int64_t nf = 0, P = 0;
    bool tree = false;
    MortonParams mp;
};

// method: 0 = choose, 1 = brute force (face order, no tree), 2 = BVH
hipError_t build_search(const DevMesh &m, int method, DevBvh &b, hipStream_t stream)
{
    const int64_t nf = m.nf;
    b.nf = nf;
    b.tree = method == 2 || (method == 0 && nf >= 64);
    MESH_TRY(b.tri.alloc(sizeof(double) * 9 * nf));
    if (nf == 0) { b.tree = false; return hipSuccess; }
    const unsigned fb = (unsigned)((nf + 255) / 256);
    if (!b.tree) {
        hipLaunchKernelGGL(tri_gather_kernel, dim3(fb), dim3(256), 0, stream, m.V.as<double>(), m.F.as<int>(),
                           (const unsigned *)nullptr, (long long)nf, b.tri.as<double>());
        return hipGetLastError();
    }
    if (nf >= ((int64_t)1 << 25)) return hipErrorInvalidValue;
    MortonParams &mp = b.mp;
    double ext = 0.0;
    for (int a = 0; a < 3; a++) {
        const double e = m.hi[a] - m.lo[a];
        mp.lo[a] = m.lo[a];
        mp.inv[a] = e > 0.0 ? 1024.0 / e : 0.0;
        ext = fmax(ext, fmax(fabs(m.lo[a]), fabs(m.hi[a])));
    }
    DBuf key, key2, val, tmp;
    size_t tmp_bytes = 0;
    MESH_TRY(key.alloc(sizeof(unsigned) * nf));
    MESH_TRY(key2.alloc(sizeof(unsigned) * nf));
    MESH_TRY(val.alloc(sizeof(unsigned) * nf));
    MESH_TRY(b.faceid.alloc(sizeof(unsigned) * nf));
    hipLaunchKernelGGL(tri_code_kernel, dim3(fb), dim3(256), 0, stream, m.V.as<double>(), m.F.as<int>(), (long long)nf,
                       mp, key.as<unsigned>(), val.as<unsigned>());
    MESH_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, key.as<unsigned>(), key2.as<unsigned>(),
                                                val.as<unsigned>(), b.faceid.as<unsigned>(), (int)nf, 0, 30, stream));
    MESH_TRY(tmp.alloc(tmp_bytes));
    MESH_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, key.as<unsigned>(), key2.as<unsigned>(),
                                                val.as<unsigned>(), b.faceid.as<unsigned>(), (int)nf, 0, 30, stream));
    hipLaunchKernelGGL(tri_gather_kernel, dim3(fb), dim3(256), 0, stream, m.V.as<double>(), m.F.as<int>(),
                       b.faceid.as<unsigned>(), (long long)nf, b.tri.as<double>());
    const int64_t nleaf = (nf + kLeaf - 1) / kLeaf;
    int64_t P = 1;
    while (P < nleaf) P <<= 1;
    b.P = P;
    MESH_TRY(b.nodes.alloc(sizeof(double) * 6 * 2 * P));
    hipLaunchKernelGGL(bvh_leaf_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, b.tri.as<double>(),
                       (long long)nf, (long long)P, ldexp(ext, -40), b.nodes.as<double>());
    for (int64_t first = P >> 1; first >= 1; first >>= 1)
        hipLaunchKernelGGL(bvh_level_kernel, dim3((unsigned)((first + 255) / 256)), dim3(256), 0, stream,
                           (long long)first, (long long)first, b.nodes.as<double>());
    MESH_TRY(hipGetLastError());
    // the temporaries above are freed when this function returns: wait for the kernels using them
    return hipStreamSynchronize(stream);
}

hipError_t launch_query(const DevBvh &b, const double *d_P, int64_t np, double *d_d2, int *d_face, double *d_cl,
                        hipStream_t stream)
{
    if (np <= 0) return hipSuccess;
    if (!b.tree) {
        hipLaunchKernelGGL(point_mesh_kernel, dim3((unsigned)((np + kMeshBlock - 1) / kMeshBlock)), dim3(kMeshBlock), 0,
                           stream, d_P, (long long)np, b.tri.as<double>(), (long long)b.nf, d_d2, d_face, d_cl);
        return hipGetLastError();
    }
    DBuf key, key2, val, order, tmp;
    const bool sorted = np >= 4096 && np < ((int64_t)1 << 31);
    if (sorted) {
        size_t tmp_bytes = 0;
        MESH_TRY(key.alloc(sizeof(unsigned) * np));
        MESH_TRY(key2.alloc(sizeof(unsigned) * np));
        MESH_TRY(val.alloc(sizeof(unsigned) * np));
        MESH_TRY(order.alloc(sizeof(unsigned) * np));
        hipLaunchKernelGGL(point_code_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, stream, d_P,
                           (long long)np, b.mp, key.as<unsigned>(), val.as<unsigned>());
        MESH_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, key.as<unsigned>(), key2.as<unsigned>(),
                                                    val.as<unsigned>(), order.as<unsigned>(), (int)np, 0, 30, stream));
        MESH_TRY(tmp.alloc(tmp_bytes));
        MESH_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, key.as<unsigned>(), key2.as<unsigned>(),
                                                    val.as<unsigned>(), order.as<unsigned>(), (int)np, 0, 30, stream));
    }
    int depth = 2;                                   // levels below the root + slack
    for (int64_t q = b.P; q > 1; q >>= 1) depth++;
    hipLaunchKernelGGL(point_mesh_bvh_kernel, dim3((unsigned)((np + kBvhBlock - 1) / kBvhBlock)), dim3(kBvhBlock),
                       (size_t)depth * kBvhBlock * 8, stream, d_P, (long long)np, b.tri.as<double>(),
                       b.faceid.as<unsigned>(), b.nodes.as<double>(), (long long)b.nf, (long long)b.P, depth,
                       sorted ? order.as<unsigned>() : (const unsigned *)nullptr, d_d2, d_face, d_cl);
    MESH_TRY(hipGetLastError());
    return sorted ? hipStreamSynchronize(stream) : hipSuccess;     // the temporaries die here
}

struct Events {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Events() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    hipError_t create() { MESH_TRY(hipEventCreate(&e0)); return hipEventCreate(&e1); }
};

// sample on the device; leaves the compacted points in `out` (n rows allocated), *m rows valid
hipError_t sample_on_device(const double *h_V, const int32_t *h_F, const DevMesh &mesh, int64_t n, int quirks,
                            unsigned long long seed, const double *h_uniforms, DBuf &out, int64_t *m,
                            hipStream_t stream)
{
    const int64_t nf = mesh.nf;
    *m = 0;
    // The cumulative area table is built on the host with the reference's own
    // sequential f64 arithmetic (geometry.h:33-43) so that face selection is identical.
    std::vector<double> cdf((size_t)nf);
    double total = 0.0;
    for (int64_t i = 0; i < nf; i++) {
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

    DBuf d_cdf, d_u, d_pts, d_valid, d_pos, d_bsum;
    unsigned last_pos = 0, last_valid = 0;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    MESH_TRY(d_cdf.alloc(sizeof(double) * nf));
    MESH_TRY(d_pts.alloc(sizeof(double) * 3 * n));
    MESH_TRY(out.alloc(sizeof(double) * 3 * n));
    MESH_TRY(d_valid.alloc(sizeof(unsigned) * (n + 1)));
    MESH_TRY(d_pos.alloc(sizeof(unsigned) * (n + 1)));
    MESH_TRY(d_bsum.alloc(sizeof(unsigned) * (n / 2048 + 2)));
    MESH_TRY(hipMemcpyAsync(d_cdf.p, cdf.data(), sizeof(double) * nf, hipMemcpyHostToDevice, stream));
    if (h_uniforms) {
        MESH_TRY(d_u.alloc(sizeof(double) * 3 * n));
        MESH_TRY(hipMemcpyAsync(d_u.p, h_uniforms, sizeof(double) * 3 * n, hipMemcpyHostToDevice, stream));
    }
    hipLaunchKernelGGL(sample_mesh_kernel, dim3(blocks), dim3(256), 0, stream, mesh.V.as<double>(), mesh.F.as<int>(),
                       d_cdf.as<double>(), (long long)nf, (long long)n, quirks, seed, d_u.as<double>(),
                       d_pts.as<double>(), d_valid.as<unsigned>());
    if (quirks) {
        MESH_TRY(launch_exclusive_scan_u32(d_valid.as<unsigned>(), (long long)n, d_bsum.as<unsigned>(),
                                           d_pos.as<unsigned>(), stream));
        hipLaunchKernelGGL(compact_points_kernel, dim3(blocks), dim3(256), 0, stream, d_pts.as<double>(),
                           d_valid.as<unsigned>(), d_pos.as<unsigned>(), (long long)n, out.as<double>());
        MESH_TRY(hipGetLastError());
        MESH_TRY(hipMemcpyAsync(&last_pos, d_pos.as<unsigned>() + (n - 1), sizeof(unsigned), hipMemcpyDeviceToHost, stream));
        MESH_TRY(hipMemcpyAsync(&last_valid, d_valid.as<unsigned>() + (n - 1), sizeof(unsigned), hipMemcpyDeviceToHost, stream));
        MESH_TRY(hipStreamSynchronize(stream));
        *m = (int64_t)last_pos + (int64_t)last_valid;
    } else {                                        // every draw yields a point: no compaction
        MESH_TRY(hipGetLastError());
        MESH_TRY(hipMemcpyAsync(out.p, d_pts.p, sizeof(double) * 3 * n, hipMemcpyDeviceToDevice, stream));
        MESH_TRY(hipStreamSynchronize(stream));
        *m = n;
    }
    return hipSuccess;
}

}  // namespace

hipError_t point_mesh_distance_device(const double *h_P, int64_t np, const double *h_V, int64_t nv,
                                      const int32_t *h_F, int64_t nf, int method, double *h_d2, int32_t *h_face,
                                      double *h_closest, float *kernel_ms, float *build_ms, hipStream_t stream)
{
    if (kernel_ms) *kernel_ms = 0.f;
    if (build_ms) *build_ms = 0.f;
    DevMesh mesh;
    DevBvh bvh;
    Events ev;
    DBuf d_P, d_d2, d_face, d_cl;
    float ms = 0.f;
    MESH_TRY(upload_mesh(h_V, nv, h_F, nf, mesh, stream));
    if (np <= 0) return hipSuccess;
    MESH_TRY(ev.create());
    MESH_TRY(hipEventRecord(ev.e0, stream));
    MESH_TRY(build_search(mesh, method, bvh, stream));
    MESH_TRY(hipEventRecord(ev.e1, stream));
    MESH_TRY(hipEventSynchronize(ev.e1));
    MESH_TRY(hipEventElapsedTime(&ms, ev.e0, ev.e1));
    if (build_ms) *build_ms = ms;
    MESH_TRY(d_P.alloc(sizeof(double) * 3 * np));
    MESH_TRY(d_d2.alloc(sizeof(double) * np));
    MESH_TRY(d_face.alloc(sizeof(int) * np));
    MESH_TRY(d_cl.alloc(sizeof(double) * 3 * np));
    MESH_TRY(hipMemcpyAsync(d_P.p, h_P, sizeof(double) * 3 * np, hipMemcpyHostToDevice, stream));
    MESH_TRY(hipEventRecord(ev.e0, stream));
    MESH_TRY(launch_query(bvh, d_P.as<double>(), np, d_d2.as<double>(), d_face.as<int>(), d_cl.as<double>(), stream));
    MESH_TRY(hipEventRecord(ev.e1, stream));
    MESH_TRY(hipMemcpyAsync(h_d2, d_d2.p, sizeof(double) * np, hipMemcpyDeviceToHost, stream));
    if (h_face) MESH_TRY(hipMemcpyAsync(h_face, d_face.p, sizeof(int) * np, hipMemcpyDeviceToHost, stream));
    if (h_closest) MESH_TRY(hipMemcpyAsync(h_closest, d_cl.p, sizeof(double) * 3 * np, hipMemcpyDeviceToHost, stream));
    MESH_TRY(hipStreamSynchronize(stream));
    if (kernel_ms) MESH_TRY(hipEventElapsedTime(kernel_ms, ev.e0, ev.e1));
    return hipSuccess;
}

hipError_t sample_mesh_device(const double *h_V, int64_t nv, const int32_t *h_F, int64_t nf, int64_t n,
                              int quirks, unsigned long long seed, const double *h_uniforms,
                              double *h_out, int64_t *n_out, hipStream_t stream)
{
    *n_out = 0;
    if (nf <= 0 || n <= 0) return hipSuccess;
    DevMesh mesh;
    DBuf pts;
    int64_t m = 0;
    MESH_TRY(upload_mesh(h_V, nv, h_F, nf, mesh, stream));
    MESH_TRY(sample_on_device(h_V, h_F, mesh, n, quirks, seed, h_uniforms, pts, &m, stream));
    if (m > 0) MESH_TRY(hipMemcpy(h_out, pts.p, sizeof(double) * 3 * m, hipMemcpyDeviceToHost));
    *n_out = m;
    return hipSuccess;
}

// The source cloud of feh::ICPRefinement (src/evaluation.cpp:252-259) without a host round trip: n draws of
// feh::SamplePointCloudFromMesh on the mesh, moved by PointCloud::Transform (PointCloud.cpp:75-80:
// transformation * (x, y, z, 1), first three rows; T16 row-major, NULL = identity), written to d_out (device,
// room for `room` points, 3 doubles each).  *m_out = points written (<= n; < n only with reference_quirks).
__global__ void transform_points_kernel(const double *__restrict__ in, long long m, const double *__restrict__ T,
                                        double *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
    if (T) {
        out[3 * i] = T[0] * x + T[1] * y + T[2] * z + T[3];
        out[3 * i + 1] = T[4] * x + T[5] * y + T[6] * z + T[7];
        out[3 * i + 2] = T[8] * x + T[9] * y + T[10] * z + T[11];
    } else {
        out[3 * i] = x; out[3 * i + 1] = y; out[3 * i + 2] = z;
    }
}

hipError_t sample_mesh_transformed_device(const double *h_V, int64_t nv, const int32_t *h_F, int64_t nf, int64_t n,
                                          int quirks, unsigned long long seed, const double *T16, double *d_out,
                                          int64_t room, int64_t *m_out, hipStream_t stream)
{
    *m_out = 0;
    if (nf <= 0 || n <= 0) return hipSuccess;
    if (n > room) return hipErrorInvalidValue;
    DevMesh mesh;
    DBuf pts, d_T;
    int64_t m = 0;
    MESH_TRY(upload_mesh(h_V, nv, h_F, nf, mesh, stream));
    MESH_TRY(sample_on_device(h_V, h_F, mesh, n, quirks, seed, nullptr, pts, &m, stream));
    if (m > 0) {
        if (T16) {
            MESH_TRY(d_T.alloc(sizeof(double) * 16));
            MESH_TRY(hipMemcpyAsync(d_T.p, T16, sizeof(double) * 16, hipMemcpyHostToDevice, stream));
        }
        hipLaunchKernelGGL(transform_points_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream,
                           pts.as<double>(), (long long)m, T16 ? d_T.as<double>() : (const double *)nullptr, d_out);
        MESH_TRY(hipGetLastError());
        MESH_TRY(hipStreamSynchronize(stream));          // (pts and d_T are freed on return)
    }
    *m_out = m;
    return hipSuccess;
}

// feh::MeasureSurfaceError (geometry.h:117-141) with the samples kept on the
// device: sample the source mesh -> distance to the target mesh -> sqrt; the
// distances come back to the host for the statistics.  h_dist holds n entries.
hipError_t surface_distances_device(const double *h_Vs, int64_t nvs, const int32_t *h_Fs, int64_t nfs,
                                    const double *h_Vt, int64_t nvt, const int32_t *h_Ft, int64_t nft, int64_t n,
                                    int quirks, unsigned long long seed, int method, double *h_dist,
                                    int64_t *n_out, float *kernel_ms, float *build_ms, hipStream_t stream)
{
    *n_out = 0;
    if (kernel_ms) *kernel_ms = 0.f;
    if (build_ms) *build_ms = 0.f;
    if (nfs <= 0 || n <= 0) return hipSuccess;
    DevMesh src, tgt;
    DevBvh bvh;
    Events ev;
    DBuf pts, d_d2;
    int64_t m = 0;
    float ms = 0.f;
    MESH_TRY(upload_mesh(h_Vs, nvs, h_Fs, nfs, src, stream));
    MESH_TRY(upload_mesh(h_Vt, nvt, h_Ft, nft, tgt, stream));
    MESH_TRY(ev.create());
    MESH_TRY(hipEventRecord(ev.e0, stream));
    MESH_TRY(build_search(tgt, method, bvh, stream));
    MESH_TRY(hipEventRecord(ev.e1, stream));
    MESH_TRY(sample_on_device(h_Vs, h_Fs, src, n, quirks, seed, nullptr, pts, &m, stream));
    MESH_TRY(hipEventElapsedTime(&ms, ev.e0, ev.e1));
    if (build_ms) *build_ms = ms;
    if (m == 0) return hipSuccess;
    MESH_TRY(d_d2.alloc(sizeof(double) * m));
    MESH_TRY(hipEventRecord(ev.e0, stream));
    MESH_TRY(launch_query(bvh, pts.as<double>(), m, d_d2.as<double>(), nullptr, nullptr, stream));
    MESH_TRY(hipEventRecord(ev.e1, stream));
    hipLaunchKernelGGL(sqrt_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, d_d2.as<double>(), (long long)m);
    MESH_TRY(hipMemcpyAsync(h_dist, d_d2.p, sizeof(double) * m, hipMemcpyDeviceToHost, stream));
    MESH_TRY(hipStreamSynchronize(stream));
    if (kernel_ms) MESH_TRY(hipEventElapsedTime(kernel_ms, ev.e0, ev.e1));
    *n_out = m;
    return hipSuccess;
}

}  // namespace visma
