// hip_engine_clouds.cpp -- HipEngine: uploads and layout of the clouds, the radius-cell grid, device buffers and their pool, timing read-back.
#include "hip_engine.hpp"

namespace visma {
namespace drv {

int HipEngine::init()
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        err_ = "no HIP device visible (this library has no CPU fallback)";
        return VISMA_ICP_ERR_NO_DEVICE;
    }
    if (device_ < 0 || device_ >= count) {
        err_ = "device index out of range";
        return VISMA_ICP_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(device_));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        err_ = std::string("device is ") + prop.gcnArchName + ", this build targets gfx950 only";
        return VISMA_ICP_ERR_NO_DEVICE;
    }
    HIP_TRY(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&stream_src_, hipStreamNonBlocking));
    if (const char *e = std::getenv("VISMA_ICP_UPLOAD_OVERLAP")) upload_overlap_ = std::atoi(e) != 0;
    if (const char *e = std::getenv("VISMA_ICP_GRID_SUB")) {
        const int v = std::atoi(e);
        if (v == 1 || v == 2) grid_sub_ = v;
    }
    if (const char *e = std::getenv("VISMA_ICP_GRID_BLOCKS")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= kGridMaxBlocks) grid_blocks_env_ = v;
    }
    HIP_TRY(hipMalloc(&d_partials_, sizeof(double) * kReduceAcc * kGridMaxBlocks));
    partial_rows_ = (size_t)kGridMaxBlocks;
    HIP_TRY(hipMalloc(&d_stats_, sizeof(double) * kNStats));
    HIP_TRY(hipMalloc(&d_cand_, 3 * 4096 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(d_cand_, 0, 3 * 4096 * sizeof(unsigned long long)));
    if (const char *e = std::getenv("VISMA_ICP_COOP")) coop_enabled_ = std::atoi(e) != 0;
    if (const char *e = std::getenv("VISMA_ICP_CERT")) cert_enabled_ = std::atoi(e) != 0;
    if (const char *e = std::getenv("VISMA_ICP_RUNNER_UP")) runner_up_ = std::atoi(e) != 0;
    if (const char *e = std::getenv("VISMA_ICP_GRID_LANES")) {
        const int v = std::atoi(e);
        if (v > 0) grid_lanes_ = v;   // G + 100*U (lanes per query, loads in flight per lane)
    }
    if (const char *e = std::getenv("VISMA_ICP_TILE")) tile_enabled_ = std::atoi(e) != 0;
    if (const char *e = std::getenv("VISMA_ICP_TILE_CONFIG")) { const int v = std::atoi(e); if (v >= 0 && v <= 10) tile_config_ = v; }
    if (const char *e = std::getenv("VISMA_ICP_TILE_FALLBACK")) tile_fallback_ = std::atoi(e) != 0;
    if (const char *e = std::getenv("VISMA_ICP_TILE_FOLD")) tile_fused_fold_ = std::atoi(e) != 0;
    if (const char *e = std::getenv("VISMA_ICP_FUSED_FOLD")) fused_fold_ = std::atoi(e) != 0;
    if (const char *e = std::getenv("VISMA_ICP_SOLVE_IN_FOLD")) solve_in_fold_ = std::atoi(e) != 0;
    HIP_TRY(hipHostMalloc(&h_stats_, sizeof(double) * 2 * kNStats,     // {value, tag} granules
                          hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(h_stats_, 0, sizeof(double) * 2 * kNStats);
    HIP_TRY(hipHostGetDevicePointer((void **)&h_stats_dev_, h_stats_, 0));
    if (const char *e = std::getenv("VISMA_ICP_PERSIST")) persist_enabled_ = std::atoi(e) != 0;
    if (const char *e = std::getenv("VISMA_ICP_PERSIST_EARLY")) persist_early_ = std::atoi(e) != 0;
    if (const char *e = std::getenv("VISMA_ICP_SWEEP_PERSIST")) sweep_persist_ = std::atoi(e) != 0;
    if (const char *e = std::getenv("VISMA_ICP_RING")) ring_mode_ = std::atoi(e) > 0 ? 1 : 0;
    if (const char *e = std::getenv("VISMA_ICP_RING_LANES")) { const int v = std::atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8) { ring_lanes_ = v; ring_lanes_auto_ = false; } }
    if (const char *e = std::getenv("VISMA_ICP_RING_T84")) { const double v = std::atof(e); if (v >= 0.0) ring_lanes_t84_ = v; }
    if (const char *e = std::getenv("VISMA_ICP_RING_T42")) { const double v = std::atof(e); if (v >= 0.0) ring_lanes_t42_ = v; }
    if (const char *e = std::getenv("VISMA_ICP_RING_OCCUPANCY")) { const double v = std::atof(e); if (v >= 1.0) ring_occ_min_ = v; }
    if (const char *e = std::getenv("VISMA_ICP_RING_TARGET")) { const double v = std::atof(e); if (v >= 1.0 && v <= 1024.0) ring_occ_target_ = v; }
    if (const char *e = std::getenv("VISMA_ICP_PERSIST_RANKS")) persist_ranks_ = std::atoi(e) != 0;
    if (const char *e = std::getenv("VISMA_ICP_COLD_IN_LAUNCH")) cold_in_launch_ = std::atoi(e) != 0;
    if (const char *e = std::getenv("VISMA_ICP_COLD_IN_LAUNCH_MIN_NS")) cold_in_launch_min_ns_ = std::max<long long>(0, std::atoll(e));
    if (const char *e = std::getenv("VISMA_ICP_PERSIST_TIMEOUT_MS")) { const double v = std::atof(e); if (v >= 1.0 && v <= 5000.0) persist_timeout_ms_ = v; }
    if (const char *e = std::getenv("VISMA_ICP_PERSIST_PRIO")) { const int v = std::atoi(e); if (v >= 0 && v <= 2) persist_prio_ = v; }
    if (const char *e = std::getenv("VISMA_ICP_PERSIST_START_MS")) { const double v = std::atof(e); if (v >= 0.0 && v <= 5000.0) persist_start_ms_ = v; }
    if (const char *e = std::getenv("VISMA_ICP_PERSIST_TIMELINE")) timeline_path_ = e;
    // The command block of the persistent launch.  On a large-BAR system it lies in fine-grained DEVICE memory the host
    // stores into through the BAR (write-combining: post_command ends with a store fence): every workgroup polls it
    // where it is -- no PCIe read per poll, no relay through a poller (tools/ubench/device_mailbox.hip: 2.3 us host ->
    // wave -> host against 2.7 with the block in host memory, and the relay hop on top of that).  Otherwise, or with
    // VISMA_ICP_PERSIST_CMD=host: mapped host memory, polled by the workgroup that published, relayed to the others.
    {
        int large_bar = 0;
        if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device_) != hipSuccess) { (void)hipGetLastError(); large_bar = 0; }
        const char *e = std::getenv("VISMA_ICP_PERSIST_CMD");
        const bool want_direct = e ? (e[0] == 'd') : large_bar != 0;
        if (want_direct && large_bar &&
            hipExtMallocWithFlags(&d_cmd_block_, sizeof(unsigned long long) * 64, hipDeviceMallocFinegrained) == hipSuccess) {
            HIP_TRY(hipMemset(d_cmd_block_, 0, sizeof(unsigned long long) * 64));
            HIP_TRY(hipDeviceSynchronize());
            h_cmd_ = h_cmd_dev_ = (unsigned long long *)d_cmd_block_;
            cmd_direct_ = true;
        } else {
            (void)hipGetLastError();
            d_cmd_block_ = nullptr;
            HIP_TRY(hipHostMalloc((void **)&h_cmd_, sizeof(unsigned long long) * 64, hipHostMallocMapped | hipHostMallocCoherent));
            std::memset(h_cmd_, 0, sizeof(unsigned long long) * 64);
            HIP_TRY(hipHostGetDevicePointer((void **)&h_cmd_dev_, h_cmd_, 0));
        }
        HIP_TRY(hipHostMalloc((void **)&h_flag_, 64, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(h_flag_, 0, 64);
        HIP_TRY(hipHostGetDevicePointer((void **)&h_flag_dev_, h_flag_, 0));
    }
    HIP_TRY(hipMalloc(&d_relay_, sizeof(unsigned long long) * 64));
    HIP_TRY(hipMemset(d_relay_, 0, sizeof(unsigned long long) * 64));
    inited_ = true;
    return VISMA_ICP_OK;
}

// The target as the caller holds it (f64 AoS) -> device, in 1 M-point pieces whose DMA runs while the next
// piece is staged.  Staging does three things in ONE pass over the caller's memory:
//  * the centroid's chunk sums (centre_out != NULL: the fixed 16 k-point chunks of centroid_f64, combined in
//    chunk order afterwards -- the same value, bit for bit, as a separate pass would give),
//  * the copy into pinned memory,
//  * and, while every value so far is exactly representable in fp32 (scans read from float PLY / PCD files,
//    depth maps: the common case), the copy is the fp32 value -- half the bytes to stage and to send; the
//    device widens it back to the identical double.  The first piece that holds a value fp32 cannot hold is
//    re-staged as f64 and the rest of the upload stays f64.
int HipEngine::set_target_f64(const double *xyz, int64_t nt, int stride, double *c, bool compute_centre, bool want64)
{
    HIP_TRY(hipSetDevice(device_));
    // (the pinned staging area and d_raw_ are about to be written: whatever an earlier upload still has in flight from
    //  them must be through -- since round 6 an upload no longer drains the stream on its way out)
    HIP_TRY(hipStreamSynchronize(stream_));
    raw_source_points_ = 0;                              // (d_raw_ is about to be reused)
    int rc = ensure_target(nt);
    if (rc) return rc;
    if (want64) { rc = pool_alloc(&d_tgt64_, sizeof(Pt64) * (size_t)std::max<int64_t>(nt, 1)); if (rc) return rc; }
    if (nt > 0) {
        if ((size_t)nt * 24 > raw_bytes_) {
            free_dev(d_raw_);
            rc = pool_alloc(&d_raw_, (size_t)nt * 24);
            if (rc) return rc;
            raw_bytes_ = (size_t)nt * 24;
        }
        double *pin = reinterpret_cast<double *>(staging(2, (size_t)nt * 6));
        const int64_t nch_all = (nt + kHostChunk - 1) / kHostChunk;
        std::vector<double> part(compute_centre ? (size_t)nch_all * 3 : 0, 0.0);
        // bounding box of the caller's values, per chunk (the grid build then needs no kernel and no round trip
        // for it: x -> (float)(x - c) is monotone, so the box of the fp32 target is the image of this one)
        std::vector<double> lohi((size_t)nch_all * 6);
        // (pieces of 1 M points: each is staged by the host threads and goes out while the next one is staged;
        //  VISMA_ICP_UPLOAD_PIECE = another multiple of 16,384 for experiments)
        static const int64_t piece = [] {
            const char *e = std::getenv("VISMA_ICP_UPLOAD_PIECE");
            const long long v = e ? std::atoll(e) : 0;
            return (v >= kHostChunk && v % kHostChunk == 0) ? (int64_t)v : (int64_t)1 << 20;
        }();
        struct Piece { int64_t lo, hi; bool f32; };
        std::vector<Piece> pieces;
        bool try32 = true;
        static const bool trace_pieces = std::getenv("VISMA_ICP_UPLOAD_TRACE") != nullptr;
        const auto t_up0 = std::chrono::steady_clock::now();
        double stage_ms = 0.0;
        for (int64_t lo = 0; lo < nt; lo += piece) {
            const auto t_p0 = std::chrono::steady_clock::now();
            const int64_t hi = std::min(nt, lo + piece);
            const int64_t nch = (hi - lo + kHostChunk - 1) / kHostChunk;      // (piece is a multiple of kHostChunk)
            std::atomic<bool> exact(true);
            bool as32 = try32;
            for (int pass = 0; pass < 2; pass++) {
                parallel_for(nch, 1, [&](int64_t ch) {
                    const int64_t a = lo + ch * kHostChunk, b = std::min(hi, a + kHostChunk);
                    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
                    double lo3[3] = {INFINITY, INFINITY, INFINITY}, hi3[3] = {-INFINITY, -INFINITY, -INFINITY};
                    for (int64_t j = a; j < b; j++) {
                        const double *q = xyz + (size_t)j * stride;
                        for (int k = 0; k < 3; k++) {
                            if (q[k] < lo3[k]) lo3[k] = q[k];
                            if (q[k] > hi3[k]) hi3[k] = q[k];
                        }
                    }
                    {
                        const int64_t g = a / kHostChunk;
                        for (int k = 0; k < 3; k++) { lohi[6 * g + k] = lo3[k]; lohi[6 * g + 3 + k] = hi3[k]; }
                    }
                    if (as32) {
                        // the piece's fp32 copy lives at the start of its own f64 area of the staging buffer
                        float *f = reinterpret_cast<float *>(pin + 3 * lo) + 3 * (a - lo);
                        bool ok = true;
                        for (int64_t j = a; j < b; j++, f += 3) {
                            const double *q = xyz + (size_t)j * stride;
                            const float x = (float)q[0], y = (float)q[1], z = (float)q[2];
                            ok = ok && (double)x == q[0] && (double)y == q[1] && (double)z == q[2];
                            f[0] = x; f[1] = y; f[2] = z;
                            s0 += q[0]; s1 += q[1]; s2 += q[2];
                        }
                        if (!ok) exact.store(false, std::memory_order_relaxed);
                    } else {
                        for (int64_t j = a; j < b; j++) {
                            const double *q = xyz + (size_t)j * stride;
                            pin[3 * j] = q[0]; pin[3 * j + 1] = q[1]; pin[3 * j + 2] = q[2];
                            s0 += q[0]; s1 += q[1]; s2 += q[2];
                        }
                    }
                    if (compute_centre) {
                        const int64_t g = a / kHostChunk;
                        part[3 * g] = s0; part[3 * g + 1] = s1; part[3 * g + 2] = s2;
                    }
                });
                if (!as32 || exact.load()) break;
                as32 = false;                                // a value fp32 cannot hold: this piece again, as f64
                try32 = false;
            }
            stage_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_p0).count();
            // (an fp32 piece of a LATER upload may still be in flight from this area: the stream is in order,
            //  and the previous upload ended with a synchronise)
            if (as32)
                HIP_TRY(hipMemcpyAsync((char *)d_raw_ + (size_t)lo * 24, pin + 3 * lo, sizeof(float) * 3 * (size_t)(hi - lo),
                                       hipMemcpyHostToDevice, stream_));
            else
                HIP_TRY(hipMemcpyAsync((double *)d_raw_ + 3 * lo, pin + 3 * lo, sizeof(double) * 3 * (size_t)(hi - lo),
                                       hipMemcpyHostToDevice, stream_));
            pieces.push_back({lo, hi, as32});
        }
        if (compute_centre) {
            c[0] = c[1] = c[2] = 0.0;
            for (int64_t ch = 0; ch < nch_all; ch++)
                for (int k = 0; k < 3; k++) c[k] += part[3 * ch + k];
            for (int k = 0; k < 3; k++) c[k] /= (double)nt;
        }
        for (const Piece &pc : pieces) {
            if (pc.f32)
                HIP_TRY(launch_expand_f32(reinterpret_cast<const float *>((const char *)d_raw_ + (size_t)pc.lo * 24), pc.hi - pc.lo,
                                          pc.lo, c, (float4 *)d_tgt_ + pc.lo, d_tgt64_ ? (Pt64 *)d_tgt64_ + pc.lo : nullptr, stream_));
            else
                HIP_TRY(launch_expand_f64((const double *)d_raw_ + 3 * pc.lo, pc.hi - pc.lo, c, (float4 *)d_tgt_ + pc.lo,
                                          d_tgt64_ ? (Pt64 *)d_tgt64_ + pc.lo : nullptr, stream_, pc.lo));
        }
        last_upload_f32_ = !pieces.empty() && pieces.back().f32;
        double lo3[3] = {INFINITY, INFINITY, INFINITY}, hi3[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int64_t ch = 0; ch < nch_all; ch++)
            for (int k = 0; k < 3; k++) {
                lo3[k] = std::min(lo3[k], lohi[6 * ch + k]);
                hi3[k] = std::max(hi3[k], lohi[6 * ch + 3 + k]);
            }
        for (int k = 0; k < 3; k++) { host_mn_[k] = (float)(lo3[k] - c[k]); host_mx_[k] = (float)(hi3[k] - c[k]); }
        host_box_valid_ = std::isfinite(host_mn_[0] + host_mn_[1] + host_mn_[2] + host_mx_[0] + host_mx_[1] + host_mx_[2]);
        if (trace_pieces) {
            const double enq = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_up0).count();
            (void)hipStreamSynchronize(stream_);
            const double all = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_up0).count();
            std::fprintf(stderr, "target upload trace: %lld points in %d pieces: staging %.3f ms of %.3f ms on the host, %.3f ms until the stream is idle\n",
                         (long long)nt, (int)pieces.size(), stage_ms, enq, all);
        }
    } else if (compute_centre) {
        c[0] = c[1] = c[2] = 0.0;
    }
    // (the copies and the expansion are queued; what follows -- the grid build, the passes -- is queued behind them on the
    //  same stream, and the source upload on its own stream needs none of it: nobody has to wait here)
    if (!upload_overlap_) HIP_TRY(hipStreamSynchronize(stream_));
    return VISMA_ICP_OK;
}

// open3d::VoxelDownSample(scene, voxel) (O3D/Core/Geometry/DownSample.cpp:179-220) + the target upload of
// RegistrationICP as ONE step (src/evaluation.cpp:258-271, src/annotation.cpp:112): the scene goes up once, is
// down-sampled on the device (voxel.hip: the reference's values bit for bit, voxels in ascending index order)
// and the result becomes the target where it lies -- the down-sampled cloud never crosses PCIe.  The centroid
// is summed on the device in the host's order (centroid_f64), so the registration that follows is the one a
// caller gets from the two separate calls, bit for bit.
int HipEngine::set_target_voxel_f64(const double *xyz, int64_t n, int stride, double voxel, double *c, bool compute_centre,
                         bool want64, int64_t *nt_out)
{
    HIP_TRY(hipSetDevice(device_));
    *nt_out = 0;
    if (n < 0 || n > 0x7fffffff - 4096) { err_ = "scene too large for 32-bit indices"; return VISMA_ICP_ERR_INVALID; }
    void *d_in = nullptr;
    if (n > 0) {
        int rc = pool_alloc(&d_in, (size_t)n * 24);
        if (rc) return rc;
        double *pin = reinterpret_cast<double *>(staging(2, (size_t)n * 6));
        const int64_t piece = 1 << 20;
        for (int64_t lo = 0; lo < n; lo += piece) {
            const int64_t hi = std::min(n, lo + piece);
            parallel_for((hi - lo + kHostChunk - 1) / kHostChunk, 1, [&](int64_t ch) {
                const int64_t a = lo + ch * kHostChunk, b = std::min(hi, a + kHostChunk);
                if (stride == 3) std::memcpy(pin + 3 * a, xyz + 3 * a, sizeof(double) * 3 * (size_t)(b - a));
                else
                    for (int64_t j = a; j < b; j++) {
                        const double *q = xyz + (size_t)j * stride;
                        pin[3 * j] = q[0]; pin[3 * j + 1] = q[1]; pin[3 * j + 2] = q[2];
                    }
            });
            HIP_TRY(hipMemcpyAsync((double *)d_in + 3 * lo, pin + 3 * lo, sizeof(double) * 3 * (size_t)(hi - lo),
                                   hipMemcpyHostToDevice, stream_));
        }
    }
    double *d_o = nullptr;
    int64_t nvox = 0;
    int too_fine = 0;
    hipError_t e = voxel_down_sample_core((const double *)d_in, nullptr, nullptr, n, voxel, &d_o, nullptr, nullptr, &nvox,
                                          &too_fine, stream_);
    free_dev(d_in);
    if (e != hipSuccess) { err_ = std::string("voxel_down_sample: ") + hipGetErrorString(e); (void)hipGetLastError(); return VISMA_ICP_ERR_HIP; }
    if (too_fine) { (void)hipFree(d_o); err_ = "voxel grid too fine to key in 62 bits"; return VISMA_ICP_ERR_INVALID; }
    if (d_vox_out_) (void)hipFree(d_vox_out_);
    d_vox_out_ = d_o;
    vox_out_n_ = nvox;
    int rc = ensure_target(nvox);
    if (rc) return rc;
    if (want64) { rc = pool_alloc(&d_tgt64_, sizeof(Pt64) * (size_t)std::max<int64_t>(nvox, 1)); if (rc) return rc; }
    if (nvox > 0) {
        if (compute_centre) {
            const int64_t nch = (nvox + kHostChunk - 1) / kHostChunk;
            double *d_part = nullptr;
            HIP_TRY(hipMalloc((void **)&d_part, sizeof(double) * (size_t)(3 * nch + 3)));
            hipError_t e2 = centroid_device(d_o, nvox, kHostChunk, d_part, d_part + 3 * nch, stream_);
            if (e2 == hipSuccess) e2 = hipMemcpyAsync(c, d_part + 3 * nch, sizeof(double) * 3, hipMemcpyDeviceToHost, stream_);
            if (e2 == hipSuccess) e2 = hipStreamSynchronize(stream_);
            (void)hipFree(d_part);
            if (e2 != hipSuccess) { err_ = std::string("centroid: ") + hipGetErrorString(e2); (void)hipGetLastError(); return VISMA_ICP_ERR_HIP; }
        }
        HIP_TRY(launch_expand_f64(d_o, nvox, c, (float4 *)d_tgt_, (Pt64 *)d_tgt64_, stream_));
    } else if (compute_centre) {
        c[0] = c[1] = c[2] = 0.0;
    }
    HIP_TRY(hipStreamSynchronize(stream_));
    *nt_out = nvox;
    return VISMA_ICP_OK;
}

// The radius of the coming registration is known (hint): the source's buffers are made first (the grid build
// sorts the target's f64 copy only when the source has one), then the grid is built on the stream -- 0.7 ms of GPU
// work at C4 that runs while the host stages the source instead of after it.  A wrong hint costs nothing but this
// build: the registration rebuilds for its own radius.
int HipEngine::prepare_search(int64_t ns, bool want64, double max_dist)
{
    HIP_TRY(hipSetDevice(device_));
    if (!(max_dist > 0.0) || nn_mode_ == VISMA_ICP_NN_BRUTE || nt_ <= 0 || ns <= 0 || tshard_) return VISMA_ICP_OK;
    std::vector<int32_t> unused;
    int rc = begin_raw_source(ns, want64, unused);
    if (rc) return rc;
    prepared_ns_ = ns;
    prepared_want64_ = want64;
    if (!(grid_valid_ && grid_radius_ == max_dist)) {
        rc = build_grid(max_dist);
        if (rc) return rc;
    }
    return VISMA_ICP_OK;
}

// buffers of a source of ns points that arrives as raw f64 triples in d_raw_
int HipEngine::begin_raw_source(int64_t ns, bool want64, std::vector<int32_t> &order)
{
    if (prepared_ns_ == ns && prepared_want64_ == want64 && ns > 0 && d_src_ && (!want64 || d_src64_)) {
        // prepare_search made these buffers (and built the grid against them) a moment ago
        prepared_ns_ = -1;
        order.resize((size_t)ns);
        return VISMA_ICP_OK;
    }
    prepared_ns_ = -1;
    int rc = ensure_source(ns);
    if (rc) return rc;
    free_dev(d_sorted64_); free_dev(d_nrm64_);
    grid_valid_ = false;
    order.resize((size_t)std::max<int64_t>(ns, 0));
    if (want64) {
        if (!d_tgt64_) { err_ = "set_source_f64 without an f64 target"; return VISMA_ICP_ERR_STATE; }
        rc = pool_alloc(&d_src64_, sizeof(Pt64) * (size_t)std::max<int64_t>(ns, 1));
        if (rc) return rc;
    }
    return VISMA_ICP_OK;
}

// d_raw_ holds ns points (caller order): Morton order on the device, fp32 + f64 copies, the permutation back
int HipEngine::finish_raw_source(int64_t ns, const double *c, std::vector<int32_t> &order, const void *raw, hipStream_t st)
{
    if (!raw) raw = d_raw_;
    if (!st) st = stream_;
    void *scratch = nullptr, *d_order = nullptr;
    const size_t sb = order_source_scratch_bytes(ns);
    int rc = pool_alloc(&scratch, sb);
    if (rc) return rc;
    rc = pool_alloc(&d_order, sizeof(int32_t) * (size_t)ns);
    if (rc) { free_dev(scratch); return rc; }
    hipError_t e = order_source_device((const double *)raw, ns, c, (float4 *)d_src_, (Pt64 *)d_src64_,
                                       (int32_t *)d_order, scratch, sb, st);
    if (e == hipSuccess)
        e = hipMemcpyAsync(order.data(), d_order, sizeof(int32_t) * (size_t)ns, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    free_dev(scratch); free_dev(d_order);
    if (e != hipSuccess) { err_ = std::string("source ordering: ") + hipGetErrorString(e); (void)hipGetLastError(); return VISMA_ICP_ERR_HIP; }
    return VISMA_ICP_OK;
}

int HipEngine::set_source_f64(const double *xyz, int64_t ns, int stride, const double *c, bool want64,
                   std::vector<int32_t> &order)
{
    HIP_TRY(hipSetDevice(device_));
    raw_source_points_ = 0;
    int rc = begin_raw_source(ns, want64, order);
    if (rc) return rc;
    if (ns > 0) {
        void *raw = nullptr;
        hipStream_t st = stream_;
        if (upload_overlap_) {
            // its own buffer, its own stream (hip_engine.hpp: stream_src_)
            if ((size_t)ns * 24 > raw_src_bytes_) {
                HIP_TRY(hipStreamSynchronize(stream_src_));
                if (d_raw_src_) (void)hipFree(d_raw_src_);
                d_raw_src_ = nullptr; raw_src_bytes_ = 0;
                HIP_TRY(hipMalloc(&d_raw_src_, (size_t)ns * 24 + (size_t)ns * 6));
                raw_src_bytes_ = (size_t)ns * 24 + (size_t)ns * 6;
            }
            raw = d_raw_src_;
            st = stream_src_;
        } else {
            rc = ensure_raw((size_t)ns);
            if (rc) return rc;
            raw = d_raw_;
        }
        double *pin = reinterpret_cast<double *>(staging(3, (size_t)ns * 6));
        parallel_for((ns + kHostChunk - 1) / kHostChunk, 1, [&](int64_t ch) {
            const int64_t a = ch * kHostChunk, b = std::min(ns, a + kHostChunk);
            if (stride == 3) std::memcpy(pin + 3 * a, xyz + 3 * a, sizeof(double) * 3 * (size_t)(b - a));
            else
                for (int64_t j = a; j < b; j++) {
                    const double *q = xyz + (size_t)j * stride;
                    pin[3 * j] = q[0]; pin[3 * j + 1] = q[1]; pin[3 * j + 2] = q[2];
                }
        });
        HIP_TRY(hipMemcpyAsync(raw, pin, sizeof(double) * 3 * (size_t)ns, hipMemcpyHostToDevice, st));
        rc = finish_raw_source(ns, c, order, raw, st);
        if (rc) return rc;
    }
    return VISMA_ICP_OK;
}

// The source of feh::ICPRefinement (src/evaluation.cpp:252-259) made where it is used: every mesh sampled on the
// device (mesh.hip), moved by its model_to_scene, the clouds concatenated in d_raw_ -- which then is what an
// uploaded source would be.  Mesh k draws from the stream seed + k.
int HipEngine::set_source_meshes_f64(const MeshSource *meshes, int n_meshes, int quirks, unsigned long long seed, const double *c,
                          bool want64, std::vector<int32_t> &order, int64_t *ns_out)
{
    HIP_TRY(hipSetDevice(device_));
    raw_source_points_ = 0;
    int64_t room = 0;
    for (int k = 0; k < n_meshes; k++) {
        if (meshes[k].samples < 0 || meshes[k].nv < 0 || meshes[k].nf < 0 ||
            (meshes[k].nf > 0 && (!meshes[k].V || !meshes[k].F))) { err_ = "bad mesh source"; return VISMA_ICP_ERR_INVALID; }
        if (meshes[k].nf > 0) room += meshes[k].samples;
    }
    if (room > 0x7fffffff - 4096) { err_ = "source too large for 32-bit indices"; return VISMA_ICP_ERR_INVALID; }
    int rc = ensure_raw((size_t)std::max<int64_t>(room, 1));
    if (rc) return rc;
    int64_t ns = 0;
    for (int k = 0; k < n_meshes; k++) {
        int64_t m = 0;
        hipError_t e = sample_mesh_transformed_device(meshes[k].V, meshes[k].nv, meshes[k].F, meshes[k].nf, meshes[k].samples,
                                                      quirks, seed + (unsigned long long)k,
                                                      meshes[k].has_transform ? meshes[k].T : nullptr,
                                                      (double *)d_raw_ + 3 * ns, room - ns, &m, stream_);
        if (e != hipSuccess) {
            err_ = std::string("mesh source: ") + (e == hipErrorInvalidValue ? "face index out of range" : hipGetErrorString(e));
            (void)hipGetLastError();
            return e == hipErrorInvalidValue ? VISMA_ICP_ERR_INVALID : VISMA_ICP_ERR_HIP;
        }
        ns += m;
    }
    rc = begin_raw_source(ns, want64, order);
    if (rc) return rc;
    if (ns > 0) {
        rc = finish_raw_source(ns, c, order);
        if (rc) return rc;
    }
    raw_source_points_ = ns;
    *ns_out = ns;
    return VISMA_ICP_OK;
}

int HipEngine::ensure_second(int64_t ns_pad, int splits)
{
    if (!d_pend_count_) {              // the list of queries the reduce kernel leaves to the rescan passes
        HIP_TRY(hipMalloc(&d_pend_count_, sizeof(int)));
        HIP_TRY(hipMalloc(&d_pend_q32_, sizeof(float4) * kBrutePendCap));
        HIP_TRY(hipMalloc(&d_pend_q64_, sizeof(Pt64) * kBrutePendCap));
        HIP_TRY(hipMalloc(&d_pend_best_, sizeof(unsigned long long) * kBrutePendCap));
        HIP_TRY(hipMalloc(&d_pend_idx_, sizeof(unsigned) * kBrutePendCap));
    }
    const size_t need = sizeof(float) * (size_t)ns_pad * (size_t)splits;
    if (need > second_bytes_) {
        free_dev(d_second_);
        HIP_TRY(hipMalloc(&d_second_, need));
        second_bytes_ = need;
    }
    return VISMA_ICP_OK;
}

int HipEngine::set_target_normals64(const Pt64 *n)
{
    HIP_TRY(hipSetDevice(device_));
    free_dev(d_nrm64_);
    if (!n || !d_tgt64_) return VISMA_ICP_OK;
    HIP_TRY(hipMalloc(&d_nrm64_, sizeof(Pt64) * std::max<int64_t>(nt_, 1)));
    if (nt_ > 0) HIP_TRY(hipMemcpy(d_nrm64_, n, sizeof(Pt64) * nt_, hipMemcpyHostToDevice));
    return VISMA_ICP_OK;
}

void HipEngine::get_timing(visma_icp_timing *t, bool reset)
{
    std::vector<unsigned long long> slots(3 * 4096, 0ull);
    (void)hipSetDevice(device_);
    (void)hipStreamSynchronize(stream_);
    (void)collect_timing();
    (void)hipMemcpy(slots.data(), d_cand_, slots.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double c[2] = {0.0, 0.0};
    for (size_t i = 0; i < 2 * 4096; i += 2) { c[0] += (double)slots[i]; c[1] += (double)slots[i + 1]; }
    timing_.grid_candidates = c[0];
    timing_.grid_candidates_27cell = c[1];
    timing_.grid_certified = 0.0;
    for (size_t i = 2 * 4096; i < 3 * 4096; i++) timing_.grid_certified += (double)slots[i];
    if (d_tstats_) {
        std::vector<unsigned long long> ts(24 * 512, 0ull);
        (void)hipMemcpy(ts.data(), d_tstats_, ts.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double v[24];
        for (int k = 0; k < 24; k++) v[k] = 0.0;
        for (size_t i = 0; i < ts.size(); i++) v[i % 24] += (double)ts[i];
        for (int k = 0; k < 6; k++) timing_.tile_phase_cycles[k] = v[8 + k];
        timing_.tile_phase_cycles[6] = 0.0;
        timing_.tile_parts = v[7];
        timing_.tile_workgroups = v[0];
        timing_.tile_fallback_workgroups = v[1];
        timing_.tile_points = v[2];
        timing_.f64_reranks = v[3];
        timing_.tile_rows = v[4];
        timing_.grid_candidates += v[5];
        timing_.grid_candidates_27cell += v[6];
    }
    *t = timing_;
    if (reset) {
        std::memset(&timing_, 0, sizeof(timing_));
        (void)hipMemset(d_cand_, 0, 3 * 4096 * sizeof(unsigned long long));
        if (d_tstats_) (void)hipMemset(d_tstats_, 0, 24 * 512 * sizeof(unsigned long long));
    }
}

int HipEngine::pool_alloc(void **p, size_t bytes)
{
    bytes = std::max<size_t>(bytes, 256);
    int best = -1;
    for (int i = 0; i < (int)pool_free_.size(); i++)
        if (pool_free_[i].second >= bytes && pool_free_[i].second <= 2 * bytes + (1u << 20) &&
            (best < 0 || pool_free_[i].second < pool_free_[best].second))
            best = i;
    if (best >= 0) {
        *p = pool_free_[best].first;
        pool_live_[*p] = pool_free_[best].second;
        pool_free_.erase(pool_free_.begin() + best);
        return VISMA_ICP_OK;
    }
    const size_t want = bytes + bytes / 8;                 // a little head-room: the next cloud is rarely the same size
    if (hipMalloc(p, want) != hipSuccess) {
        (void)hipGetLastError();
        pool_trim(0);                                      // give the recycled buffers back and try the exact size
        HIP_TRY(hipMalloc(p, bytes));
        pool_live_[*p] = bytes;
        return VISMA_ICP_OK;
    }
    pool_live_[*p] = want;
    return VISMA_ICP_OK;
}

int HipEngine::ensure_target(int64_t nt)
{
    if (nt < 0) { err_ = "negative point count"; return VISMA_ICP_ERR_INVALID; }
    if (nt > 0x7fffffff - 4096) { err_ = "target too large for 32-bit indices"; return VISMA_ICP_ERR_INVALID; }
    free_dev(d_tgt_); free_dev(d_nrm_); free_dev(d_tgt64_); free_dev(d_sorted64_); free_dev(d_nrm64_);
    has_normals_ = false;
    host_box_valid_ = false;
    // pad to a whole number of LDS chunks with +inf points (never accepted)
    nt_pad_ = ((nt + kTChunk - 1) / kTChunk) * kTChunk;
    if (nt_pad_ == 0) nt_pad_ = kTChunk;
    { int prc = pool_alloc(&d_tgt_, sizeof(float4) * (size_t)nt_pad_); if (prc) return prc; }
    HIP_TRY(launch_fill_inf((float4 *)d_tgt_ + nt, nt_pad_ - nt, stream_));
    nt_ = nt;
    grid_valid_ = false;
    ring_flips_ = 0;
    have_pass_ = false;
    return VISMA_ICP_OK;
}

int HipEngine::ensure_aux(int64_t ns_pad)
{
    if (ns_pad > aux_cap_) {
        free_dev(d_idx_); free_dev(d_d2_); free_dev(d_pos_); free_dev(d_ru_);
        HIP_TRY(hipMalloc(&d_idx_, sizeof(int32_t) * (ns_pad > 0 ? ns_pad : 1)));
        HIP_TRY(hipMalloc(&d_d2_, sizeof(float) * (ns_pad > 0 ? ns_pad : 1)));
        HIP_TRY(hipMalloc(&d_pos_, sizeof(Pt64) * (ns_pad > 0 ? ns_pad : 1)));
        if (runner_up_) HIP_TRY(hipMalloc(&d_ru_, sizeof(Pt64) * (ns_pad > 0 ? ns_pad : 1)));
        aux_cap_ = ns_pad;
        return invalidate_pos();
    }
    return VISMA_ICP_OK;
}

// Pick brute force or the grid for this (target, radius); build the grid if needed.
int HipEngine::choose_mode(double max_dist)
{
    if (nn_mode_ == VISMA_ICP_NN_BRUTE || nt_ == 0) { use_grid_ = false; return VISMA_ICP_OK; }
    {
        int rc = ensure_f64_views();
        if (rc) return rc;
    }
    if (grid_valid_ && grid_radius_ == max_dist && ring_replan_needed()) {
        // (between the two occupancy thresholds a single registration wants the ring search, a sweep radius-sized cells)
        ring_flips_++;
        grid_valid_ = false;
    }
    if (!(grid_valid_ && grid_radius_ == max_dist)) {
        int rc = build_grid(max_dist);
        if (rc) return rc;
    }
    if (nn_mode_ == VISMA_ICP_NN_GRID) { use_grid_ = true; return VISMA_ICP_OK; }
    // AUTO: the grid pays off when a 3x3x3 neighbourhood is a small part of the
    // target; a degenerate grid (few cells) would scan most of the cloud per
    // query without LDS tiling -- use the tiled brute-force kernel there.  (With a
    // proper grid the fused kernel wins at every size measured: 20-22 us per
    // iteration against 33-37 for 500 ... 4000 target points.)
    use_grid_ = grid_.ncell >= 512;
    if (shard_f64_protocol()) use_grid_ = true;              // sharded ranks: the exact search on every shard
    // Small f64 clouds keep the (f64) grid search even on a degenerate grid -- a radius that is
    // large against the cloud's extent -- so that their correspondences stay the reference's;
    // scanning most of a few-thousand-point target per query is cheap.
    if (!use_grid_ && d_src64_ && d_tgt64_ && (double)ns_ * (double)nt_ <= 2.0e8) use_grid_ = true;
    return VISMA_ICP_OK;
}

int HipEngine::build_grid(double max_dist)
{
    int e0 = -1;
    if (profiling_) { e0 = next_event_pair(); HIP_TRY(hipEventRecord(ev_[e0], stream_)); }
    float mn[3], mx[3];
    static const bool check_box = std::getenv("VISMA_ICP_CHECK_BOX") != nullptr;   // (tests: both ways, must agree)
    if (!host_box_valid_ || check_box) {
        if (!d_box_) HIP_TRY(hipMalloc(&d_box_, sizeof(unsigned) * 8));
        HIP_TRY(launch_grid_bbox((const float4 *)d_tgt_, nt_, (unsigned *)d_box_, stream_));
        unsigned box[6];
        HIP_TRY(hipMemcpyAsync(box, d_box_, sizeof(box), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        grid_decode_bbox(box, mn, mx);
        if (host_box_valid_ && (std::memcmp(mn, host_mn_, sizeof(mn)) != 0 || std::memcmp(mx, host_mx_, sizeof(mx)) != 0)) {
            err_ = "bounding box from the staging pass differs from the device's";
            return VISMA_ICP_ERR_ENGINE;
        }
    } else {
        // (known from the upload's staging pass: no kernel, no round trip)
        std::memcpy(mn, host_mn_, sizeof(mn));
        std::memcpy(mx, host_mx_, sizeof(mx));
    }
    grid_ = grid_plan(mn, mx, max_dist, kGridMaxCells, grid_sub_);
    grid_occupancy_ = 0.0;
    // cells, counts, scan, scatter (+ the f64 and the packed copies) for the plan in grid_
    auto build = [&]() -> int {
        if ((int64_t)nt_ > sorted_cap_) {
            free_dev(d_sorted_); free_dev(d_cell_of_);
            HIP_TRY(hipMalloc(&d_sorted_, sizeof(float4) * (nt_ + kSortedSlack)));   // batches read past a run's end
            HIP_TRY(hipMalloc(&d_cell_of_, 2 * sizeof(unsigned) * nt_));   // (cell, rank in the cell)
            sorted_cap_ = nt_;
        }
        if (grid_.ncell + 1 > cell_cap_) {
            free_dev(d_count_); free_dev(d_start_); free_dev(d_bsum_);
            HIP_TRY(hipMalloc(&d_count_, sizeof(unsigned) * (grid_.ncell + 1)));
            HIP_TRY(hipMalloc(&d_start_, sizeof(unsigned) * (grid_.ncell + 8)));    // (16-byte reads near the end)
            HIP_TRY(hipMemsetAsync(d_start_, 0, sizeof(unsigned) * (grid_.ncell + 8), stream_));
            HIP_TRY(hipMalloc(&d_bsum_, sizeof(unsigned) * (grid_scan_blocks(grid_.ncell) + 1)));
            cell_cap_ = grid_.ncell + 1;
        }
        free_dev(d_sorted64_);
        if (d_tgt64_ && d_src64_) { int prc = pool_alloc(&d_sorted64_, sizeof(Pt64) * (size_t)std::max<int64_t>(nt_, 1)); if (prc) return prc; }
        HIP_TRY(launch_grid_build((const float4 *)d_tgt_, nt_, grid_, (unsigned *)d_cell_of_,
                                  (unsigned *)d_count_, (unsigned *)d_bsum_, (unsigned *)d_start_,
                                  (float4 *)d_sorted_, stream_, d_sorted64_ ? (const Pt64 *)d_tgt64_ : nullptr,
                                  (Pt64 *)d_sorted64_));
        // the exact search ranks on a packed copy: 12 bytes per candidate (grid.hip: P12)
        free_dev(d_sorted12_);
        if (d_sorted64_) {
            int prc = pool_alloc(&d_sorted12_, sizeof(float) * 3 * (size_t)(nt_ + kSortedSlack));
            if (prc) return prc;
            HIP_TRY(launch_pack12((const float4 *)d_sorted_, (float *)d_sorted12_, nt_, stream_));
        }
        return VISMA_ICP_OK;
    };
    {
        int brc = build();
        if (brc) return brc;
    }
    // ---- a radius that is large against the point spacing?  (grid_ring.hip)  The radius-sized table tells: points per
    // occupied cell.  Counted only where it can matter -- the count is a pass over the table and a round trip: a surface
    // occupies a few n^2 of a table of n^3 cells, so its occupancy is at most ~n times the mean over ALL cells (C4:
    // 0.08 x 428 = 34, not counted; r = 0.15 m on the same target: 524 x 20, counted: 3,200).
    const bool ring_possible = ring_mode_ != 0 && d_tgt64_ && d_src64_ && nt_ > 0 && grid_sub_ <= 1 && grid_.sub == 1 &&
                               nn_mode_ != VISMA_ICP_NN_BRUTE;
    const double occ_guess = (double)nt_ / (double)std::max<int64_t>(grid_.ncell, 1) *
                             (double)std::max(grid_.dim[0], std::max(grid_.dim[1], grid_.dim[2]));
    ring_for_sweep_ = caller_sweep_;
    if (ring_possible && (ring_mode_ > 0 || occ_guess >= ring_occ_min_)) {
        if (!d_occ_) HIP_TRY(hipMalloc(&d_occ_, sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(d_occ_, 0, sizeof(unsigned long long), stream_));
        HIP_TRY(launch_count_occupied((const unsigned *)d_count_, grid_.ncell, (unsigned long long *)d_occ_, stream_));
        unsigned long long occupied = 0ull;
        HIP_TRY(hipMemcpyAsync(&occupied, d_occ_, sizeof(occupied), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        grid_occupancy_ = occupied ? (double)nt_ / (double)occupied : 0.0;
        if (grid_occupancy_ > ring_occ_target_ && (ring_mode_ > 0 || grid_occupancy_ >= ring_threshold())) {
            // surfaces: the occupancy falls with the square of the edge
            const double cell = (double)grid_.h * std::sqrt(ring_occ_target_ / grid_occupancy_);
            const GridParams fine = grid_plan_ring(mn, mx, max_dist, cell, kGridMaxCells);
            if (fine.ring > 0) {
                // the rows' visiting order around a query (the same for every query: a table per ring count)
                if (!d_ring_tab_ || ring_tab_rings_ != fine.ring) {
                    free_dev(d_ring_tab_);
                    ring_tab_rings_ = 0;
                    int nrows = 0;
                    HIP_TRY(build_ring_table(fine.ring, &d_ring_tab_, &nrows));
                    ring_tab_rings_ = fine.ring;
                }
                grid_ = fine;
                ring_tab_ = RingTable{(const RingRow *)d_ring_tab_, nrows_of_rings(fine.ring)};
                int brc = build();
                if (brc) return brc;
            }
        }
    }
    if (profiling_) { HIP_TRY(hipEventRecord(ev_[e0 + 1], stream_)); pending_.push_back({e0, 2}); }
    grid_valid_ = true;
    grid_radius_ = max_dist;
    return invalidate_pos();                                 // the slots of the old sorted order mean nothing now
}

int HipEngine::collect_timing()
{
    for (const auto &p : pending_) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev_[p.first], ev_[p.first + 1]));
        if (p.second == 0) { timing_.nn_ms += ms; timing_.nn_launches++; }
        else if (p.second == 1) { timing_.reduce_ms += ms; timing_.reduce_launches++; }
        else { timing_.aux_ms += ms; timing_.aux_launches++; }
    }
    pending_.clear();
    // persistent launches: the whole launch is timed (waits for the host included), its passes counted
    for (const auto &p : sess_pending_) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev_[p.first], ev_[p.first + 1]));
        timing_.nn_ms += ms;
        timing_.nn_launches += p.second;
        timing_.persist_ms += ms;
        timing_.persist_launches += 1.0;
        timing_.persist_passes += (double)p.second;
    }
    sess_pending_.clear();
    ev_used_ = 0;
    return VISMA_ICP_OK;
}

int HipEngine::ensure_tile_buffers(size_t rows, size_t ticket_words)
{
    if (rows > partial_rows_) {
        free_dev(d_partials_);
        HIP_TRY(hipMalloc(&d_partials_, sizeof(double) * kReduceAcc * rows));
        partial_rows_ = rows;
    }
    if (ticket_words > tickets_cap_) {
        free_dev(d_partials2_); free_dev(d_tickets_);
        HIP_TRY(hipMalloc(&d_partials2_, sizeof(double) * kReduceAcc * ticket_words));
        HIP_TRY(hipMalloc(&d_tickets_, sizeof(unsigned) * ticket_words));
        HIP_TRY(hipMemsetAsync(d_tickets_, 0, sizeof(unsigned) * ticket_words, stream_));
        tickets_cap_ = ticket_words;
    }
    if (!d_tstats_) {
        HIP_TRY(hipMalloc(&d_tstats_, sizeof(unsigned long long) * 24 * 512));
        HIP_TRY(hipMemsetAsync(d_tstats_, 0, sizeof(unsigned long long) * 24 * 512, stream_));
    }
    return VISMA_ICP_OK;
}

// f64 views of clouds that were uploaded as fp32 (the exact search needs them)
int HipEngine::ensure_f64_views()
{
    if (!exact_) return VISMA_ICP_OK;
    if (!d_src64_ && d_src_) {
        HIP_TRY(hipMalloc(&d_src64_, sizeof(Pt64) * std::max<int64_t>(ns_, 1)));
        HIP_TRY(launch_promote_pt64((const float4 *)d_src_, (Pt64 *)d_src64_, ns_, stream_));
    }
    if (!d_tgt64_ && d_tgt_) {
        HIP_TRY(hipMalloc(&d_tgt64_, sizeof(Pt64) * std::max<int64_t>(nt_, 1)));
        HIP_TRY(launch_promote_pt64((const float4 *)d_tgt_, (Pt64 *)d_tgt64_, nt_, stream_));
        free_dev(d_sorted64_);
        grid_valid_ = false;
    }
    if (!d_sorted64_) grid_valid_ = false;
    return VISMA_ICP_OK;
}

}  // namespace drv
}  // namespace visma
