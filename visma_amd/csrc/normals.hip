// normals.hip -- open3d::EstimateNormals on the GPU
// (O3D/Core/Geometry/EstimateNormals.cpp:38-163; declared O3D/Core/Geometry/PointCloud.h:140-146):
// the step a caller needs before the point-to-plane estimator when its clouds carry no normals
// (Registration.cpp:152-157 refuses to run without them).
//
// Per point: neighbours by one of the three KDTreeFlann searches (KDTreeFlann.cpp:114-189) --
//   KNN     the knn nearest points (the point itself included), ascending distance,
//   Radius  every point with d2 < (double)(float)(r*r)  (flann radiusSearch, strict),
//   Hybrid  the max_nn nearest of those --
// then the covariance of the neighbours from their first and second moments summed in the order of
// the result list (EstimateNormals.cpp:86-112), the eigenvector of its smallest eigenvalue by the
// closed form of FastEigen3x3 (:40-83), (0,0,1) when fewer than 3 neighbours or a zero vector,
// and the sign of an existing normal kept (:133-146).  All f64, flann's sum of squares.
//
// The search runs on the radius-cell grid of grid.hip: cells of edge 1.001 r (Radius / Hybrid: the
// 27 cells cover the radius), or of about the distance to the knn-th neighbour (KNN: rings of cells
// are added until the knn-th distance is covered -- exact for any cell size).  One thread per
// point, in cell order; the result list (d2, index) lives in LDS, sorted by (d2, index) -- the
// reference's order up to exactly equal distances, which flann orders by tree traversal.  Lists longer than
// kNormalsMaxList entries (the LDS capacity at 64 threads) live in global memory as a max-heap per point that is
// heap-sorted at the end: any knn / max_nn.  Radius searches (no bound on the list) first COUNT every point's
// neighbours, then run as a Hybrid search whose max_nn is the largest count: the moments are summed in list order
// like the reference's (round 2 summed them in scan order: 3e-7 off where two eigenvalues nearly coincide).
#include "device_common.h"

#include <math.h>

#include <algorithm>
#include <vector>

namespace visma {

namespace {

struct NormalArgs {
    const float4 *sorted;        // cell-sorted fp32 copy (cell binning only)
    const Pt64 *sorted64;        // cell-sorted points, w = original index
    const unsigned *start;
    GridParams g;
    const Pt64 *pts;             // original order
    const double *nrm_in;        // n x 3 or NULL
    double *nrm_out;             // n x 3
    int n;
    int cap;                     // list capacity (knn / max_nn); unused for Radius
    double r2d;                  // (double)(float)(r * r); unused for KNN
    unsigned long long *n27;     // (sampling pass) sum of 27-cell populations, queries counted
    int *count_out;              // (counting pass) neighbours within the radius, per point in cell order
    double *spill_d2;            // (spilled lists) [cap][q_count] squared distances ...
    int *spill_id;               // ... and indices, one max-heap per query of the slab
    int q_begin, q_count;        // the slab of queries (cell order) this launch works on
};

__device__ __forceinline__ double sqr(double x) { return x * x; }

// EstimateNormals.cpp:40-83.  A symmetric; returns false for a zero vector.
__device__ bool fast_eigen_3x3(const double A[3][3], double out[3])
{
    const double p1 = sqr(A[0][1]) + sqr(A[0][2]) + sqr(A[1][2]);
    double ev0, ev1, ev2;
    const double trace = A[0][0] + A[1][1] + A[2][2];
    if (p1 == 0.0) {
        ev2 = fmin(A[0][0], fmin(A[1][1], A[2][2]));
        ev0 = fmax(A[0][0], fmax(A[1][1], A[2][2]));
        ev1 = trace - ev0 - ev2;
    } else {
        const double q = trace / 3.0;
        const double p2 = sqr(A[0][0] - q) + sqr(A[1][1] - q) + sqr(A[2][2] - q) + 2 * p1;
        const double p = sqrt(p2 / 6.0);
        const double ip = 1.0 / p;
        double B[3][3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) B[i][j] = ip * (A[i][j] - (i == j ? q : 0.0));
        // Eigen's 3x3 determinant (Determinant.h: bruteforce_det3_helper)
        const double det = B[0][0] * (B[1][1] * B[2][2] - B[1][2] * B[2][1]) -
                           B[0][1] * (B[1][0] * B[2][2] - B[1][2] * B[2][0]) +
                           B[0][2] * (B[1][0] * B[2][1] - B[1][1] * B[2][0]);
        const double r = det / 2.0;
        double phi;
        if (r <= -1) phi = M_PI / 3.0;
        else if (r >= 1) phi = 0.0;
        else phi = acos(r) / 3.0;
        ev0 = q + 2.0 * p * cos(phi);
        ev2 = q + 2.0 * p * cos(phi + 2.0 * M_PI / 3.0);
        ev1 = q * 3.0 - ev0 - ev2;
    }
    (void)ev2;
    // (A - I ev0) * (A.col(0) - (ev1, 0, 0))
    const double v[3] = {A[0][0] - ev1, A[1][0], A[2][0]};
    double e[3];
    for (int i = 0; i < 3; i++) {
        const double m0 = A[i][0] - (i == 0 ? ev0 : 0.0), m1 = A[i][1] - (i == 1 ? ev0 : 0.0),
                     m2 = A[i][2] - (i == 2 ? ev0 : 0.0);
        e[i] = m0 * v[0] + m1 * v[1] + m2 * v[2];
    }
    const double len = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    if (len == 0.0) return false;
    out[0] = e[0] / len; out[1] = e[1] / len; out[2] = e[2] / len;
    return true;
}

// TYPE: 0 KNN, 1 Radius (counting pass only), 2 Hybrid.  MODE 1: only count the 27-cell population of every
// 64th point (cell-size tuning); MODE 2: only count the neighbours within the radius.  SPILL: the list is a
// max-heap in global memory instead of a sorted array in LDS.
template <int TYPE, int NTH, int MODE, bool SPILL>
__global__ __launch_bounds__(NTH) void estimate_normals_kernel(NormalArgs a)
{
    constexpr bool SAMPLE = MODE == 1, COUNT = MODE == 2;
    extern __shared__ double lds_raw[];
    const int tid = threadIdx.x;
    const long long u = (long long)blockIdx.x * NTH + tid;          // query of the slab
    if (u >= a.q_count) return;
    const long long t = a.q_begin + u;
    // element j of this thread's list: LDS [cap][NTH] or global [cap][q_count] (coalesced across threads)
    double *ld2 = SPILL ? a.spill_d2 + u : lds_raw + tid;
    int *lid = SPILL ? a.spill_id + u : reinterpret_cast<int *>(lds_raw + (size_t)a.cap * NTH) + tid;
    const size_t lstride = SPILL ? (size_t)a.q_count : (size_t)NTH;
    const Pt64 q = a.sorted64[t];                                 // queries in cell order
    const float4 qf = a.sorted[t];
    const GridParams g = a.g;
    const int cx = cell_coord(qf.x, g.mn[0], g.inv_h, g.dim[0]);
    const int cy = cell_coord(qf.y, g.mn[1], g.inv_hs, g.dim[1]);
    const int cz = cell_coord(qf.z, g.mn[2], g.inv_hs, g.dim[2]);
    if (SAMPLE) {
        if ((t & 63) != 0) return;
        unsigned long long tot = 0;
        for (int dz = -1; dz <= 1; dz++)
            for (int dy = -1; dy <= 1; dy++) {
                const int z = cz + dz, y = cy + dy;
                if (z < 0 || z >= g.dim[2] || y < 0 || y >= g.dim[1]) continue;
                const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dim[0] - 1);
                const long long row = ((long long)z * g.dim[1] + y) * g.dim[0];
                tot += a.start[row + x1 + 1] - a.start[row + x0];
            }
        atomicAdd(a.n27, tot);
        atomicAdd(a.n27 + 1, 1ull);
        return;
    }
    int cnt = 0;
    double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto moments = [&](const double x, const double y, const double z) {
        // EstimateNormals.cpp:95-105
        c[0] += x; c[1] += y; c[2] += z;
        c[3] += x * x; c[4] += x * y; c[5] += x * z;
        c[6] += y * y; c[7] += y * z; c[8] += z * z;
    };
    auto less = [&](double d, int id, int j) {
        const double dj = ld2[(size_t)j * lstride];
        return d < dj || (d == dj && id < lid[(size_t)j * lstride]);
    };
    // (SPILL) max-heap on (d2, index): the root is the entry a nearer candidate replaces
    auto sift_down = [&](int pos, int len, double d, int id) {
        for (;;) {
            int ch = 2 * pos + 1;
            if (ch >= len) break;
            if (ch + 1 < len && less(ld2[(size_t)ch * lstride], lid[(size_t)ch * lstride], ch + 1)) ch++;
            if (!less(d, id, ch)) break;
            ld2[(size_t)pos * lstride] = ld2[(size_t)ch * lstride];
            lid[(size_t)pos * lstride] = lid[(size_t)ch * lstride];
            pos = ch;
        }
        ld2[(size_t)pos * lstride] = d;
        lid[(size_t)pos * lstride] = id;
    };
    auto consider = [&](const Pt64 &p) {
        // flann L2 (dist.h:159-176): result += diff * diff over x, y, z
        const double dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
        double d = dx * dx;
        d += dy * dy;
        d += dz * dz;
        if (TYPE != 0 && !(d < a.r2d)) return;
        if (COUNT) { cnt++; return; }
        const int id = (int)p.w;
        if (SPILL) {
            if (cnt < a.cap) {
                // sift up: parents smaller than the new entry move down
                int pos = cnt++;
                while (pos > 0) {
                    const int par = (pos - 1) >> 1;
                    const double dp = ld2[(size_t)par * lstride];
                    const int ip = lid[(size_t)par * lstride];
                    if (!(dp < d || (dp == d && ip < id))) break;
                    ld2[(size_t)pos * lstride] = dp;
                    lid[(size_t)pos * lstride] = ip;
                    pos = par;
                }
                ld2[(size_t)pos * lstride] = d;
                lid[(size_t)pos * lstride] = id;
            } else if (less(d, id, 0)) {
                sift_down(0, a.cap, d, id);
            }
            return;
        }
        int pos;
        if (cnt < a.cap) pos = cnt++;
        else if (less(d, id, a.cap - 1)) pos = a.cap - 1;
        else return;
        while (pos > 0 && less(d, id, pos - 1)) {
            ld2[(size_t)pos * lstride] = ld2[(size_t)(pos - 1) * lstride];
            lid[(size_t)pos * lstride] = lid[(size_t)(pos - 1) * lstride];
            pos--;
        }
        ld2[(size_t)pos * lstride] = d;
        lid[(size_t)pos * lstride] = id;
    };
    auto scan_cells = [&](int z, int y, int xa, int xb) {              // cells xa..xb of row (y, z), clipped
        if (z < 0 || z >= g.dim[2] || y < 0 || y >= g.dim[1]) return;
        xa = max(xa, 0); xb = min(xb, g.dim[0] - 1);
        if (xa > xb) return;
        const long long row = ((long long)z * g.dim[1] + y) * g.dim[0];
        const unsigned b = a.start[row + xa], e = a.start[row + xb + 1];
        for (unsigned j = b; j < e; j++) consider(a.sorted64[j]);
    };
    const int rmax = TYPE == 0 ? max(max(g.dim[0], g.dim[1]), g.dim[2]) : 1;
    for (int R = 0; R <= rmax; R++) {
        // the shell of cells at Chebyshev distance R: full x-runs on its y/z faces, two cells elsewhere
        for (int dz = -R; dz <= R; dz++)
            for (int dy = -R; dy <= R; dy++) {
                if (max(abs(dy), abs(dz)) == R) scan_cells(cz + dz, cy + dy, cx - R, cx + R);
                else { scan_cells(cz + dz, cy + dy, cx - R, cx - R); scan_cells(cz + dz, cy + dy, cx + R, cx + R); }
            }
        if (TYPE != 0) continue;
        // every point nearer than R cell edges has been seen (0.1 % slack for the fp32 binning)
        if (cnt == a.cap && R >= 1) {
            const double cover = (double)R * (double)g.h * 0.999;
            const double farthest = ld2[(size_t)(SPILL ? 0 : a.cap - 1) * lstride];      // heap root / end of the sorted list
            if (farthest <= cover * cover) break;
        }
        if (cx - R <= 0 && cy - R <= 0 && cz - R <= 0 && cx + R >= g.dim[0] - 1 && cy + R >= g.dim[1] - 1 &&
            cz + R >= g.dim[2] - 1)
            break;                                                  // the whole grid has been scanned
    }
    if (COUNT) { a.count_out[t] = cnt; return; }
    const int me = (int)q.w;
    double nrm[3] = {0.0, 0.0, 1.0};                                // fewer than 3 neighbours (:147-149)
    if (cnt >= 3) {
        if (SPILL) {
            // heap sort in place: ascending (d2, index) = the order of the reference's result list
            for (int end = cnt - 1; end > 0; end--) {
                const double dl = ld2[(size_t)end * lstride];
                const int il = lid[(size_t)end * lstride];
                ld2[(size_t)end * lstride] = ld2[0];
                lid[(size_t)end * lstride] = lid[0];
                sift_down(0, end, dl, il);
            }
        }
        for (int j = 0; j < cnt; j++) {
            const Pt64 p = a.pts[lid[(size_t)j * lstride]];
            moments(p.x, p.y, p.z);
        }
        const double inv = (double)cnt;
        for (int k = 0; k < 9; k++) c[k] /= inv;
        double A[3][3];
        A[0][0] = c[3] - c[0] * c[0];
        A[1][1] = c[6] - c[1] * c[1];
        A[2][2] = c[8] - c[2] * c[2];
        A[0][1] = A[1][0] = c[4] - c[0] * c[1];
        A[0][2] = A[2][0] = c[5] - c[0] * c[2];
        A[1][2] = A[2][1] = c[7] - c[1] * c[2];
        const bool ok = fast_eigen_3x3(A, nrm);
        if (!ok) {
            if (a.nrm_in) { nrm[0] = a.nrm_in[3ll * me]; nrm[1] = a.nrm_in[3ll * me + 1]; nrm[2] = a.nrm_in[3ll * me + 2]; }
            else { nrm[0] = 0.0; nrm[1] = 0.0; nrm[2] = 1.0; }
        }
        if (a.nrm_in) {
            const double dot = nrm[0] * a.nrm_in[3ll * me] + nrm[1] * a.nrm_in[3ll * me + 1] + nrm[2] * a.nrm_in[3ll * me + 2];
            if (dot < 0.0) { nrm[0] *= -1.0; nrm[1] *= -1.0; nrm[2] *= -1.0; }
        }
    }
    a.nrm_out[3ll * me] = nrm[0];
    a.nrm_out[3ll * me + 1] = nrm[1];
    a.nrm_out[3ll * me + 2] = nrm[2];
}

// The fp32 copy is used for BINNING only and is taken relative to a point of the cloud (subtracted in f64): its
// rounding error then scales with the cloud's extent, not with its distance from the origin -- a scan 100 m from
// the origin with a 5 mm radius would otherwise put neighbours one cell off (0.1 % slack between cell and radius).
// Distances and moments use the caller's f64 coordinates, so the results do not depend on the shift.
__global__ void pack_points_kernel(const double *__restrict__ xyz, long long n, double cx, double cy, double cz,
                                   float4 *__restrict__ f4, Pt64 *__restrict__ p8)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    f4[i] = make_float4((float)(x - cx), (float)(y - cy), (float)(z - cz), __uint_as_float((unsigned)i));
    p8[i] = Pt64{x, y, z, (unsigned long long)i};
}

struct DevBufs {
    std::vector<void *> ptrs;
    ~DevBufs() { for (void *p : ptrs) (void)hipFree(p); }
    template <class T>
    hipError_t alloc(T **out, size_t count)
    {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, sizeof(T) * std::max<size_t>(count, 1));
        if (e != hipSuccess) return e;
        ptrs.push_back(p);
        *out = (T *)p;
        return hipSuccess;
    }
};

template <int TYPE, int MODE, bool SPILL>
hipError_t launch_normals(const NormalArgs &a, hipStream_t stream)
{
    // list bytes per thread: cap * 12 (LDS lists only); keep a workgroup's list within 128 KiB of LDS
    const size_t per_thread = (MODE != 0 || SPILL) ? 0 : (size_t)a.cap * 12;
    int nth = 256;
    while (nth > 64 && per_thread * nth > 60 * 1024) nth >>= 1;      // (64 threads: up to 128 KiB of the 160)
    const size_t lds = per_thread * nth + 64;
    const unsigned blocks = (unsigned)((a.q_count + nth - 1) / nth);
    if (a.q_count <= 0) return hipSuccess;
#define VISMA_NRM_LAUNCH(NTH_)                                                                                  \
    do {                                                                                                        \
        if (lds > 48 * 1024) {                                                                                  \
            hipError_t e = hipFuncSetAttribute((const void *)estimate_normals_kernel<TYPE, NTH_, MODE, SPILL>,  \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
            if (e != hipSuccess) return e;                                                                      \
        }                                                                                                       \
        hipLaunchKernelGGL((estimate_normals_kernel<TYPE, NTH_, MODE, SPILL>), dim3(blocks), dim3(NTH_), lds, stream, a); \
    } while (0)
    if (nth == 256) VISMA_NRM_LAUNCH(256);
    else if (nth == 128) VISMA_NRM_LAUNCH(128);
    else VISMA_NRM_LAUNCH(64);
#undef VISMA_NRM_LAUNCH
    return hipGetLastError();
}

}  // namespace

#define NRM_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

// search_type 0 KNN(knn) | 1 Radius(radius) | 2 Hybrid(radius, knn = max_nn); h_nrm_in may be NULL
hipError_t estimate_normals_device(const double *h_xyz, int64_t n, const double *h_nrm_in, int search_type,
                                   int knn, double radius, double *h_out, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    if (search_type < 0 || search_type > 2 || n > 0x7fffffff) return hipErrorInvalidValue;
    const bool use_r = search_type != 0;
    // fewer than 3 neighbours for every point (KDTreeFlann.cpp:121-126,171-176 return -1 / k < 3)
    const bool trivial = (search_type != 1 && knn < 3) || (use_r && !(radius > 0.0));
    if (trivial) {
        for (int64_t i = 0; i < n; i++) { h_out[3 * i] = 0.0; h_out[3 * i + 1] = 0.0; h_out[3 * i + 2] = 1.0; }
        return hipSuccess;
    }
    int cap = search_type == 1 ? 1 : (int)std::min<int64_t>(knn, n);           // (Radius: set from the counting pass)
    DevBufs B;
    double *d_xyz = nullptr, *d_nin = nullptr, *d_out = nullptr;
    float4 *d_f4 = nullptr, *d_sorted = nullptr;
    Pt64 *d_p8 = nullptr, *d_sorted64 = nullptr;
    unsigned *d_box = nullptr, *d_cell_of = nullptr;
    unsigned long long *d_n27 = nullptr;
    NRM_TRY(B.alloc(&d_xyz, 3 * (size_t)n));
    NRM_TRY(B.alloc(&d_out, 3 * (size_t)n));
    NRM_TRY(B.alloc(&d_f4, (size_t)n));
    NRM_TRY(B.alloc(&d_sorted, (size_t)n + kSortedSlack));
    NRM_TRY(B.alloc(&d_p8, (size_t)n));
    NRM_TRY(B.alloc(&d_sorted64, (size_t)n));
    NRM_TRY(B.alloc(&d_box, 8));
    NRM_TRY(B.alloc(&d_cell_of, 2 * (size_t)n));   // (cell, rank in the cell)
    NRM_TRY(B.alloc(&d_n27, 2));
    if (h_nrm_in) {
        NRM_TRY(B.alloc(&d_nin, 3 * (size_t)n));
        NRM_TRY(hipMemcpyAsync(d_nin, h_nrm_in, sizeof(double) * 3 * n, hipMemcpyHostToDevice, stream));
    }
    NRM_TRY(hipMemcpyAsync(d_xyz, h_xyz, sizeof(double) * 3 * n, hipMemcpyHostToDevice, stream));
    {
        // a finite point of the cloud as the binning origin
        double c0[3] = {0.0, 0.0, 0.0};
        for (int64_t i = 0; i < n; i++)
            if (std::isfinite(h_xyz[3 * i]) && std::isfinite(h_xyz[3 * i + 1]) && std::isfinite(h_xyz[3 * i + 2])) {
                c0[0] = h_xyz[3 * i]; c0[1] = h_xyz[3 * i + 1]; c0[2] = h_xyz[3 * i + 2];
                break;
            }
        hipLaunchKernelGGL(pack_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_xyz,
                           (long long)n, c0[0], c0[1], c0[2], d_f4, d_p8);
    }
    NRM_TRY(launch_grid_bbox(d_f4, n, d_box, stream));
    unsigned box[6];
    NRM_TRY(hipMemcpyAsync(box, d_box, sizeof(box), hipMemcpyDeviceToHost, stream));
    NRM_TRY(hipStreamSynchronize(stream));
    float mn[3], mx[3];
    grid_decode_bbox(box, mn, mx);
    double diag = 0.0;
    for (int k = 0; k < 3; k++) diag += ((double)mx[k] - mn[k]) * ((double)mx[k] - mn[k]);
    diag = std::sqrt(diag);
    // KNN: a first guess of the distance to the knn-th neighbour on a surface; refined below from
    // the measured population of the 27 cells
    double cell = use_r ? radius : 0.6 * diag * std::sqrt((double)cap / (double)n);
    if (!(cell > 0.0) || !std::isfinite(cell)) cell = 1.0;
    NormalArgs a{};
    unsigned *d_count = nullptr, *d_start = nullptr, *d_bsum = nullptr;
    int64_t cell_cap = 0;
    for (int attempt = 0;; attempt++) {
        const int64_t max_cells = std::min<int64_t>(kGridMaxCells, std::max<int64_t>(4096, 8 * n));
        const GridParams g = grid_plan(mn, mx, cell, max_cells);
        if (g.ncell + 1 > cell_cap) {
            NRM_TRY(B.alloc(&d_count, (size_t)g.ncell + 1));
            NRM_TRY(B.alloc(&d_start, (size_t)g.ncell + 9));
            NRM_TRY(B.alloc(&d_bsum, (size_t)grid_scan_blocks(g.ncell) + 1));
            cell_cap = g.ncell + 1;
        }
        NRM_TRY(launch_grid_build(d_f4, n, g, d_cell_of, d_count, d_bsum, d_start, d_sorted, stream, d_p8, d_sorted64));
        a.sorted = d_sorted; a.sorted64 = d_sorted64; a.start = d_start; a.g = g; a.pts = d_p8;
        a.nrm_in = d_nin; a.nrm_out = d_out; a.n = (int)n; a.cap = cap;
        a.q_begin = 0; a.q_count = (int)n;
        const float r2f = (float)(radius * radius);
        a.r2d = (double)r2f;
        a.n27 = d_n27;
        if (use_r || attempt >= 3) break;
        // KNN: aim at 2.5 knn points in the 27 cells (one ring is then enough for most points)
        NRM_TRY(hipMemsetAsync(d_n27, 0, 2 * sizeof(unsigned long long), stream));
        NRM_TRY((launch_normals<0, 1, false>(a, stream)));
        unsigned long long h27[2];
        NRM_TRY(hipMemcpyAsync(h27, d_n27, sizeof(h27), hipMemcpyDeviceToHost, stream));
        NRM_TRY(hipStreamSynchronize(stream));
        const double mean27 = h27[1] ? (double)h27[0] / h27[1] : 0.0, want = 2.5 * cap;
        if (mean27 >= 0.6 * want && mean27 <= 2.5 * want) break;
        const double f = mean27 > 0.0 ? std::sqrt(want / mean27) : 2.0;      // surface scaling; rings cover the rest
        const double next = cell * std::min(std::max(f, 0.25), 4.0);
        if (g.h > (float)(cell * 1.01) && next < cell) break;              // the cell cap already enlarged the cells
        cell = next;
    }
    int type = search_type;
    std::vector<int> cnt;                                        // (Radius) neighbours per query, cell order
    if (search_type == 1) {
        // Radius: count every point's neighbours, then run as a Hybrid search that keeps them all
        int *d_cnt = nullptr;
        NRM_TRY(B.alloc(&d_cnt, (size_t)n));
        a.count_out = d_cnt;
        NRM_TRY((launch_normals<1, 2, false>(a, stream)));
        cnt.resize((size_t)n);
        NRM_TRY(hipMemcpyAsync(cnt.data(), d_cnt, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, stream));
        NRM_TRY(hipStreamSynchronize(stream));
        type = 2;
    }
    // Runs of queries (cell order) and the list capacity each needs.  KNN / Hybrid: one run, the caller's bound.
    // Radius: the capacity of a run is the largest count AMONG ITS OWN queries (one dense corner of a scan
    // used to give every query of the cloud a list -- and, beyond the LDS, a global-memory heap -- of that corner's
    // length); runs of kSlab queries, neighbouring runs whose lists fit the LDS merged into one launch.
    struct Run { int64_t q0, qn; int cap; };
    std::vector<Run> runs;
    if (search_type == 1) {
        const int64_t kSlab = 16384;
        for (int64_t q0 = 0; q0 < n; q0 += kSlab) {
            const int64_t qn = std::min<int64_t>(kSlab, n - q0);
            const int c = std::max(1, *std::max_element(cnt.begin() + q0, cnt.begin() + q0 + qn));
            if (!runs.empty() && c <= kNormalsMaxList && runs.back().cap <= kNormalsMaxList) {
                runs.back().qn += qn;
                runs.back().cap = std::max(runs.back().cap, c);
            } else {
                runs.push_back(Run{q0, qn, c});
            }
        }
    } else {
        runs.push_back(Run{0, n, cap});
    }
    // lists longer than the LDS holds: one max-heap per query in global memory, a slab of queries at a time; the
    // slab's heaps take at most 1 GiB or half of what the device has free
    size_t budget = (size_t)1 << 30;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min(budget, free_b / 2);
    }
    size_t spill_entries = 0;
    for (Run &r : runs)
        if (r.cap > kNormalsMaxList) {
            if ((size_t)r.cap * 12 * 64 > budget) return hipErrorOutOfMemory;   // (64 queries' heaps do not fit)
            const int64_t slab = std::min<int64_t>(r.qn, std::max<int64_t>(64, (int64_t)(budget / ((size_t)r.cap * 12))));
            spill_entries = std::max(spill_entries, (size_t)r.cap * (size_t)slab);
        }
    double *d_sd = nullptr;
    int *d_si = nullptr;
    if (spill_entries) {
        NRM_TRY(B.alloc(&d_sd, spill_entries));
        NRM_TRY(B.alloc(&d_si, spill_entries));
        a.spill_d2 = d_sd; a.spill_id = d_si;
    }
    for (const Run &r : runs) {
        a.cap = r.cap;
        if (r.cap <= kNormalsMaxList) {
            a.q_begin = (int)r.q0;
            a.q_count = (int)r.qn;
            if (type == 0) NRM_TRY((launch_normals<0, 0, false>(a, stream)));
            else NRM_TRY((launch_normals<2, 0, false>(a, stream)));
            continue;
        }
        const int64_t slab = std::min<int64_t>(r.qn, std::max<int64_t>(64, (int64_t)(budget / ((size_t)r.cap * 12))));
        for (int64_t q0 = r.q0; q0 < r.q0 + r.qn; q0 += slab) {
            a.q_begin = (int)q0;
            a.q_count = (int)std::min<int64_t>(slab, r.q0 + r.qn - q0);
            if (type == 0) NRM_TRY((launch_normals<0, 0, true>(a, stream)));
            else NRM_TRY((launch_normals<2, 0, true>(a, stream)));
        }
    }
    NRM_TRY(hipMemcpyAsync(h_out, d_out, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, stream));
    NRM_TRY(hipStreamSynchronize(stream));
    return hipSuccess;
}

}  // namespace visma
