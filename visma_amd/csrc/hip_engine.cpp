// hip_engine.cpp -- HipEngine: the product engine of the ICP driver (gfx950 kernels on one HIP stream):
// cloud upload and layout, the radius-cell grid, the per-pass launches (brute force / lane-serial grid /
// warm-started cooperative grid), the fused fold, the device-resident loops (single problem, sweeps, batches of
// problems with their own clouds), the transports of the source- and target-sharded modes (IPC mailboxes, RCCL).
#include "engine.hpp"

namespace visma {
namespace drv {

namespace {

class HipEngine : public Engine {
public:
    explicit HipEngine(int device) : device_(device) {}
    ~HipEngine() override
    {
        if (!inited_) { (void)hipGetLastError(); return; }   // never touched the device
        (void)hipSetDevice(device_);
        if (comm_) g_rccl.CommDestroy(comm_);
        for (int r = 0; r < ipc_n_; r++)
            if (r != ipc_rank_ && peers_.box[r]) (void)hipIpcCloseMemHandle(peers_.box[r]);
        free_dev(d_mbox_); free_dev(d_ipc_flag_); free_dev(d_raw_); free_dev(d_sorted12_); free_dev(bt_sorted12_);
        free_dev(d_pend_count_); free_dev(d_pend_q32_); free_dev(d_pend_q64_); free_dev(d_pend_best_); free_dev(d_pend_idx_);
        for (hipEvent_t e : ev_) (void)hipEventDestroy(e);
        free_dev(d_src_); free_dev(d_tgt_); free_dev(d_nrm_); free_dev(d_keys_); free_dev(d_gkeys_);
        free_dev(d_claim_); free_dev(d_d64_);
        free_dev(d_src64_); free_dev(d_tgt64_); free_dev(d_sorted64_); free_dev(d_nrm64_);
        for (int i = 0; i < 4; i++) if (pin_[i]) (void)hipHostFree(pin_[i]);
        free_dev(d_idx_); free_dev(d_d2_); free_dev(d_pos_); free_dev(d_partials_); free_dev(d_stats_);
        if (d_vox_out_) (void)hipFree(d_vox_out_);
        free_dev(d_box_); free_dev(d_sorted_); free_dev(d_cell_of_); free_dev(d_count_);
        free_dev(d_start_); free_dev(d_bsum_); free_dev(d_cand_); free_dev(d_state_);
        free_dev(d_partials2_); free_dev(d_tickets_); free_dev(d_tstats_); free_dev(d_second_);
        free_dev(bt_src_); free_dev(bt_idx_); free_dev(bt_d2_); free_dev(bt_pos_); free_dev(bt_tgt_); free_dev(bt_sorted_);
        free_dev(bt_nrm_); free_dev(bt_nrm64_); free_dev(bt_raw_);
        free_dev(bt_src64_); free_dev(bt_tgt64_); free_dev(bt_sorted64_);
        free_dev(bt_cell_of_); free_dev(bt_count_); free_dev(bt_start_); free_dev(bt_bsum_); free_dev(bt_descs_);
        if (h_state_) (void)hipHostFree(h_state_);
        if (h_stats_) (void)hipHostFree(h_stats_);
        pool_trim(0);
        if (stream_) (void)hipStreamDestroy(stream_);
    }

    int init()
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
        if (const char *e = std::getenv("VISMA_ICP_GRID_LANES")) {
            const int v = std::atoi(e);
            if (v > 0) grid_lanes_ = v;   // G + 100*U (lanes per query, loads in flight per lane)
        }
        if (const char *e = std::getenv("VISMA_ICP_TILE")) tile_enabled_ = std::atoi(e) != 0;
        if (const char *e = std::getenv("VISMA_ICP_TILE_CONFIG")) { const int v = std::atoi(e); if (v >= 0 && v <= 10) tile_config_ = v; }
        if (const char *e = std::getenv("VISMA_ICP_TILE_FALLBACK")) tile_fallback_ = std::atoi(e) != 0;
        if (const char *e = std::getenv("VISMA_ICP_TILE_FOLD")) tile_fused_fold_ = std::atoi(e) != 0;
        if (const char *e = std::getenv("VISMA_ICP_FUSED_FOLD")) fused_fold_ = std::atoi(e) != 0;
        HIP_TRY(hipHostMalloc(&h_stats_, sizeof(double) * 2 * kNStats,     // {value, tag} granules
                              hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(h_stats_, 0, sizeof(double) * 2 * kNStats);
        HIP_TRY(hipHostGetDevicePointer((void **)&h_stats_dev_, h_stats_, 0));
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
    int set_target_f64(const double *xyz, int64_t nt, int stride, double *c, bool compute_centre, bool want64) override
    {
        HIP_TRY(hipSetDevice(device_));
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
            const int64_t piece = 1 << 20;                       // (a parallel_for starts its threads anew)
            struct Piece { int64_t lo, hi; bool f32; };
            std::vector<Piece> pieces;
            bool try32 = true;
            for (int64_t lo = 0; lo < nt; lo += piece) {
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
        } else if (compute_centre) {
            c[0] = c[1] = c[2] = 0.0;
        }
        HIP_TRY(hipStreamSynchronize(stream_));
        return VISMA_ICP_OK;
    }
    bool last_upload_f32_ = false;   // (reported by VISMA_ICP_UPLOAD_TRACE)
    bool host_box_valid_ = false;    // the fp32 target's bounding box is known from the staging pass
    float host_mn_[3] = {0, 0, 0}, host_mx_[3] = {0, 0, 0};

    // open3d::VoxelDownSample(scene, voxel) (O3D/Core/Geometry/DownSample.cpp:179-220) + the target upload of
    // RegistrationICP as ONE step (src/evaluation.cpp:258-271, src/annotation.cpp:112): the scene goes up once, is
    // down-sampled on the device (voxel.hip: the reference's values bit for bit, voxels in ascending index order)
    // and the result becomes the target where it lies -- the down-sampled cloud never crosses PCIe.  The centroid
    // is summed on the device in the host's order (centroid_f64), so the registration that follows is the one a
    // caller gets from the two separate calls, bit for bit.
    int set_target_voxel_f64(const double *xyz, int64_t n, int stride, double voxel, double *c, bool compute_centre,
                             bool want64, int64_t *nt_out) override
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
    // the down-sampled cloud of the last set_target_voxel_f64 (kept on the device until the next one), for callers
    // that want the points as well
    int get_voxel_target(double *out, int64_t n) override
    {
        HIP_TRY(hipSetDevice(device_));
        if (n != vox_out_n_) { err_ = "no down-sampled target of that size"; return VISMA_ICP_ERR_STATE; }
        if (n > 0) HIP_TRY(hipMemcpy(out, d_vox_out_, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToHost));
        return VISMA_ICP_OK;
    }
    double *d_vox_out_ = nullptr;
    int64_t vox_out_n_ = -1;
    // The radius of the coming registration is known (hint): the source's buffers are made first (the grid build
    // sorts the target's f64 copy only when the source has one), then the grid is built on the stream -- 0.7 ms of GPU
    // work at C4 that runs while the host stages the source instead of after it.  A wrong hint costs nothing but this
    // build: the registration rebuilds for its own radius.
    int prepare_search(int64_t ns, bool want64, double max_dist) override
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
    int64_t prepared_ns_ = -1;
    bool prepared_want64_ = false;
    // buffers of a source of ns points that arrives as raw f64 triples in d_raw_
    int begin_raw_source(int64_t ns, bool want64, std::vector<int32_t> &order)
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
    int ensure_raw(size_t points)
    {
        if (points * 24 > raw_bytes_) {
            free_dev(d_raw_);
            int rc = pool_alloc(&d_raw_, points * 24);
            if (rc) return rc;
            raw_bytes_ = points * 24;
        }
        return VISMA_ICP_OK;
    }
    // d_raw_ holds ns points (caller order): Morton order on the device, fp32 + f64 copies, the permutation back
    int finish_raw_source(int64_t ns, const double *c, std::vector<int32_t> &order)
    {
        void *scratch = nullptr, *d_order = nullptr;
        const size_t sb = order_source_scratch_bytes(ns);
        int rc = pool_alloc(&scratch, sb);
        if (rc) return rc;
        rc = pool_alloc(&d_order, sizeof(int32_t) * (size_t)ns);
        if (rc) { free_dev(scratch); return rc; }
        hipError_t e = order_source_device((const double *)d_raw_, ns, c, (float4 *)d_src_, (Pt64 *)d_src64_,
                                           (int32_t *)d_order, scratch, sb, stream_);
        if (e == hipSuccess)
            e = hipMemcpyAsync(order.data(), d_order, sizeof(int32_t) * (size_t)ns, hipMemcpyDeviceToHost, stream_);
        if (e == hipSuccess) e = hipStreamSynchronize(stream_);
        free_dev(scratch); free_dev(d_order);
        if (e != hipSuccess) { err_ = std::string("source ordering: ") + hipGetErrorString(e); (void)hipGetLastError(); return VISMA_ICP_ERR_HIP; }
        return VISMA_ICP_OK;
    }
    int set_source_f64(const double *xyz, int64_t ns, int stride, const double *c, bool want64,
                       std::vector<int32_t> &order) override
    {
        HIP_TRY(hipSetDevice(device_));
        raw_source_points_ = 0;
        int rc = begin_raw_source(ns, want64, order);
        if (rc) return rc;
        if (ns > 0) {
            rc = ensure_raw((size_t)ns);
            if (rc) return rc;
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
            HIP_TRY(hipMemcpyAsync(d_raw_, pin, sizeof(double) * 3 * (size_t)ns, hipMemcpyHostToDevice, stream_));
            rc = finish_raw_source(ns, c, order);
            if (rc) return rc;
        }
        return VISMA_ICP_OK;
    }
    // The source of feh::ICPRefinement (src/evaluation.cpp:252-259) made where it is used: every mesh sampled on the
    // device (mesh.hip), moved by its model_to_scene, the clouds concatenated in d_raw_ -- which then is what an
    // uploaded source would be.  Mesh k draws from the stream seed + k.
    int set_source_meshes_f64(const MeshSource *meshes, int n_meshes, int quirks, unsigned long long seed, const double *c,
                              bool want64, std::vector<int32_t> &order, int64_t *ns_out) override
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
    int get_mesh_source(double *out, int64_t ns) override
    {
        HIP_TRY(hipSetDevice(device_));
        if (ns != raw_source_points_ || !d_raw_) { err_ = "no mesh-sampled source of that size on this context"; return VISMA_ICP_ERR_STATE; }
        if (ns > 0) HIP_TRY(hipMemcpy(out, d_raw_, sizeof(double) * 3 * (size_t)ns, hipMemcpyDeviceToHost));
        return VISMA_ICP_OK;
    }
    int64_t raw_source_points_ = 0;                        // > 0: d_raw_ holds the mesh-sampled source (caller order)
    int set_source64(const Pt64 *src) override
    {
        HIP_TRY(hipSetDevice(device_));
        free_dev(d_src64_); free_dev(d_sorted64_); free_dev(d_nrm64_);
        grid_valid_ = false;
        { int irc = invalidate_pos(); if (irc) return irc; }
        if (!src || !d_tgt64_) { err_ = "set_source64 without an f64 target"; return VISMA_ICP_ERR_STATE; }
        { int prc = pool_alloc(&d_src64_, sizeof(Pt64) * (size_t)std::max<int64_t>(ns_, 1)); if (prc) return prc; }
        if (ns_ > 0) HIP_TRY(hipMemcpyAsync(d_src64_, src, sizeof(Pt64) * ns_, hipMemcpyHostToDevice, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        return VISMA_ICP_OK;
    }
    void *d_raw_ = nullptr;
    size_t raw_bytes_ = 0;
    void *d_sorted12_ = nullptr;                           // packed (x,y,z) copy of d_sorted_ for the exact search
    const float4 *search_sorted() const                    // what launch_nn_grid_reduce gets as `sorted`
    {
        if (exact_ && d_src64_ && d_sorted64_ && d_sorted12_) return (const float4 *)d_sorted12_;
        return (const float4 *)d_sorted_;
    }
    int set_clouds64(const Pt64 *src, const Pt64 *tgt) override
    {
        HIP_TRY(hipSetDevice(device_));
        free_dev(d_src64_); free_dev(d_tgt64_); free_dev(d_sorted64_); free_dev(d_nrm64_);
        grid_valid_ = false;                                     // the sorted f64 copy is built with the grid
        { int irc = invalidate_pos(); if (irc) return irc; }
        if (!src || !tgt) return VISMA_ICP_OK;
        { int prc = pool_alloc(&d_src64_, sizeof(Pt64) * (size_t)std::max<int64_t>(ns_, 1)); if (prc) return prc; }
        { int prc = pool_alloc(&d_tgt64_, sizeof(Pt64) * (size_t)std::max<int64_t>(nt_, 1)); if (prc) return prc; }
        if (ns_ > 0) HIP_TRY(hipMemcpyAsync(d_src64_, src, sizeof(Pt64) * ns_, hipMemcpyHostToDevice, stream_));
        if (nt_ > 0) HIP_TRY(hipMemcpyAsync(d_tgt64_, tgt, sizeof(Pt64) * nt_, hipMemcpyHostToDevice, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        return VISMA_ICP_OK;
    }
    // arithmetic of the last pass / loop / batch: 0 fp32 ranking only, 1 exact (fp32 + f64 re-rank), 2 f64
    bool search_is_f64() const override { return last_mode_ == 2; }
    bool search_is_exact() const override { return last_mode_ != 0; }
    int grid_search_mode() const { return (use_grid_ && d_src64_ && d_sorted64_) ? (exact_ ? 1 : 2) : (brute_exact() ? 1 : 0); }
    // the brute-force kernels run their exact flavour when the f64 clouds are there (not on sharded ranks,
    // which exchange the fp32 keys of this path)
    bool brute_exact() const { return !use_grid_ && exact_ && d_src64_ && d_tgt64_ && !tshard_; }
    int ensure_second(int64_t ns_pad, int splits)
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
    void *d_pend_count_ = nullptr, *d_pend_q32_ = nullptr, *d_pend_q64_ = nullptr, *d_pend_best_ = nullptr,
         *d_pend_idx_ = nullptr;
    BruteExact bex_store_{};
    const BruteExact *bex_ptr()
    {
        if (!brute_exact()) return nullptr;
        bex_store_ = brute_ex();
        return &bex_store_;
    }
    BruteExact brute_ex() const
    {
        BruteExact e;
        e.src64 = (const Pt64 *)d_src64_;
        e.tgt64 = (const Pt64 *)d_tgt64_;
        e.nrm64 = (const Pt64 *)d_nrm64_;
        e.second = (const float *)d_second_;
        e.nt = nt_;
        e.pend = BrutePend{};
        if (d_pend_count_ && 2 * reduce_max_blocks() <= (int)partial_rows_) {
            e.pend.count = (int *)d_pend_count_;
            e.pend.q32 = (float4 *)d_pend_q32_;
            e.pend.q64 = (Pt64 *)d_pend_q64_;
            e.pend.best = (unsigned long long *)d_pend_best_;
            e.pend.best_idx = (unsigned *)d_pend_idx_;
        }
        return e;
    }
    void set_exact(bool on) override { exact_ = on; }
    int set_target_normals64(const Pt64 *n) override
    {
        HIP_TRY(hipSetDevice(device_));
        free_dev(d_nrm64_);
        if (!n || !d_tgt64_) return VISMA_ICP_OK;
        HIP_TRY(hipMalloc(&d_nrm64_, sizeof(Pt64) * std::max<int64_t>(nt_, 1)));
        if (nt_ > 0) HIP_TRY(hipMemcpy(d_nrm64_, n, sizeof(Pt64) * nt_, hipMemcpyHostToDevice));
        return VISMA_ICP_OK;
    }

    float *staging(int slot, size_t nfloats) override
    {
        slot &= 3;
        if (pin_cap_[slot] < nfloats) {
            if (pin_[slot]) (void)hipHostFree(pin_[slot]);
            pin_[slot] = nullptr;
            pin_cap_[slot] = 0;
            const size_t want = nfloats + nfloats / 4 + 1024;
            if (hipSetDevice(device_) != hipSuccess ||
                hipHostMalloc((void **)&pin_[slot], want * sizeof(float), hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                pin_[slot] = nullptr;
                return Engine::staging(slot, nfloats);     // pageable memory still works, only slower
            }
            pin_cap_[slot] = want;
        }
        return pin_[slot];
    }
    int set_source(const float *xyzw, int64_t ns) override
    {
        HIP_TRY(hipSetDevice(device_));
        int rc = ensure_source(ns);
        if (rc) return rc;
        if (ns > 0) HIP_TRY(hipMemcpyAsync(d_src_, xyzw, sizeof(float4) * ns, hipMemcpyHostToDevice, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        return VISMA_ICP_OK;
    }
    int set_source_device(const void *d, int64_t ns) override
    {
        HIP_TRY(hipSetDevice(device_));
        int rc = ensure_source(ns);
        if (rc) return rc;
        if (ns > 0) HIP_TRY(hipMemcpyAsync(d_src_, d, sizeof(float4) * ns, hipMemcpyDeviceToDevice, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        return VISMA_ICP_OK;
    }
    int set_target(const float *xyzw, int64_t nt) override
    {
        HIP_TRY(hipSetDevice(device_));
        int rc = ensure_target(nt);
        if (rc) return rc;
        if (nt > 0) HIP_TRY(hipMemcpyAsync(d_tgt_, xyzw, sizeof(float4) * nt, hipMemcpyHostToDevice, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        return VISMA_ICP_OK;
    }
    int set_target_device(const void *d, int64_t nt) override
    {
        HIP_TRY(hipSetDevice(device_));
        int rc = ensure_target(nt);
        if (rc) return rc;
        if (nt > 0) HIP_TRY(hipMemcpyAsync(d_tgt_, d, sizeof(float4) * nt, hipMemcpyDeviceToDevice, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        return VISMA_ICP_OK;
    }
    int set_target_normals(const float *nxyzw, int64_t nt) override
    {
        HIP_TRY(hipSetDevice(device_));
        if (nt != nt_) { err_ = "normals count != target count"; return VISMA_ICP_ERR_INVALID; }
        free_dev(d_nrm_);
        HIP_TRY(hipMalloc(&d_nrm_, sizeof(float4) * (nt > 0 ? nt : 1)));
        if (nt > 0) HIP_TRY(hipMemcpy(d_nrm_, nxyzw, sizeof(float4) * nt, hipMemcpyHostToDevice));
        has_normals_ = true;
        return VISMA_ICP_OK;
    }

    int nn_pass(const Mat4 &Tc, double max_dist) override
    {
        HIP_TRY(hipSetDevice(device_));
        if (!d_src_ || !d_tgt_) { err_ = "clouds not set"; return VISMA_ICP_ERR_STATE; }
        const int64_t ns_min_pad = ((ns_ + kBlock - 1) / kBlock) * kBlock;
        for (int i = 0; i < 12; i++) { T32_.m[i] = (float)Tc.m[i]; T64_last_.m[i] = Tc.m[i]; }
        r2f_ = (float)(max_dist * max_dist);
        r2d_ = (double)r2f_;                                     // (double)(float)(r*r): KDTreeFlann.cpp:184-185
        int rc = choose_mode(max_dist);
        if (rc) return rc;
        view_offset_ = 0;
        last_mode_ = grid_search_mode();
        if (use_grid_) {
            // the grid search is fused with the reduction: it runs in reduce()
            // (or in get_correspondences() if no reduction is asked for)
            rc = ensure_aux(ns_min_pad);
            if (rc) return rc;
            grid_pending_ = true;
            have_pass_ = true;
            return VISMA_ICP_OK;
        }
        plan_ = nn_plan(ns_, nt_pad_);
        const int64_t ns_pad = (int64_t)plan_.src_tiles * kBlock * plan_.spt;
        const size_t need = sizeof(unsigned long long) * (size_t)ns_pad * plan_.tgt_splits;
        if (need > keys_bytes_) {
            free_dev(d_keys_);
            HIP_TRY(hipMalloc(&d_keys_, need));
            keys_bytes_ = need;
        }
        rc = ensure_aux(ns_pad);
        if (rc) return rc;
        ns_pad_ = ns_pad;
        const bool bex = brute_exact();
        if (bex) { rc = ensure_second(ns_pad_, plan_.tgt_splits); if (rc) return rc; }
        last_mode_ = grid_search_mode();
        int e0 = -1;
        if (profiling_) { e0 = next_event_pair(); HIP_TRY(hipEventRecord(ev_[e0], stream_)); }
        HIP_TRY(launch_nn_brute((const float4 *)d_src_, ns_, (const float4 *)d_tgt_, nt_pad_, T32_,
                                r2f_, (unsigned long long *)d_keys_, ns_pad_, plan_, nullptr, stream_,
                                bex ? (const Pt64 *)d_src64_ : nullptr, &T64_last_, bex ? (float *)d_second_ : nullptr));
        if (profiling_) { HIP_TRY(hipEventRecord(ev_[e0 + 1], stream_)); pending_.push_back({e0, 0}); }
        grid_pending_ = false;
        brute_reduced_ = false;
        have_pass_ = true;
        return VISMA_ICP_OK;
    }

    int reduce(const Mat4 &Tc, bool plane, const double offset[3], double *stats) override
    {
        HIP_TRY(hipSetDevice(device_));
        if (!have_pass_) { err_ = "reduce before nn_pass"; return VISMA_ICP_ERR_STATE; }
        if (plane && !d_nrm_) { err_ = "point-to-plane needs target normals"; return VISMA_ICP_ERR_STATE; }
        Xform64 T64;
        for (int i = 0; i < 12; i++) T64.m[i] = Tc.m[i];
        int e0 = -1;
        // profiling level n > 1: time (and count candidates on) every n-th pass only --
        // four event records per iteration cost ~14 us of the ~75 they measure
        const bool prof = profiling_ > 0 && (++prof_tick_ % profiling_) == 0;
        // without RCCL the fold kernel publishes to mapped host memory itself
        const unsigned long long seq = ++pub_seq_;
        const bool ipc = ipc_n_ > 1;
        bool ipc_done = false;                           // the exchange ran inside the search launch
        double *pub = (comm_ || ipc) ? nullptr : h_stats_dev_;
#ifdef VISMA_WITH_TILE
        if (use_tile()) {
            // ONE launch: streamed search + exact re-rank + moments + fused fold + publication
            const int cfg = tile_config(ns_);
            const int nblocks = tile_blocks(ns_, cfg);
            const size_t tstride = 1 + (size_t)(nblocks + 31) / 32;
            int rc = ensure_tile_buffers((size_t)nblocks, tstride);
            if (rc) return rc;
            TileArgs ta = tile_args(T64, offset, prof);
            ta.bpp = nblocks;
            ta.tickets = tile_fused_fold_ ? (unsigned *)d_tickets_ : nullptr;
            ta.ticket_stride = (int)tstride;
            ta.stats_out = (double *)d_stats_;
            ta.stats_stride = 0;
            ta.host_out = pub;
            ta.seq = seq;
            if (prof) { e0 = next_event_pair(); HIP_TRY(hipEventRecord(ev_[e0], stream_)); }
            HIP_TRY(launch_nn_tile_reduce(ta, plane ? 1 : 0, cfg, nblocks, stream_));
            if (prof) { HIP_TRY(hipEventRecord(ev_[e0 + 1], stream_)); pending_.push_back({e0, 0}); }
            if (!tile_fused_fold_) {
                if (prof) { e0 = next_event_pair(); HIP_TRY(hipEventRecord(ev_[e0], stream_)); }
                HIP_TRY(launch_finalize((const double *)d_partials_, nblocks, plane ? 1 : 0,
                                        (double *)d_stats_, stream_, pub, seq));
                if (prof) { HIP_TRY(hipEventRecord(ev_[e0 + 1], stream_)); pending_.push_back({e0, 1}); }
            }
            grid_pending_ = false;
        } else
#endif
        if (use_grid_) {
            int nblocks = 1;
            // the fold of the partial rows runs inside the search launch (no second kernel)
            const bool fused = fused_fold_ && !tshard_;
            const int lanes = pass_lanes();
            FoldArgs fa{};
            if (fused) {
                // (peer-to-peer mailboxes: the folding workgroup exchanges with the peers and publishes itself)
                int rc = make_fold(grid_launch_blocks(ns_, lanes, grid_blocks()), 1, (double *)d_stats_, 0,
                                   ipc ? h_stats_dev_ : pub, seq, &fa);
                if (rc) return rc;
                if (ipc) { add_ipc(&fa); ipc_done = true; }
            }
            if (prof) { e0 = next_event_pair(); HIP_TRY(hipEventRecord(ev_[e0], stream_)); }
            HIP_TRY(launch_nn_grid_reduce((const float4 *)d_src_, ns_, search_sorted(),
                                          (const unsigned *)d_start_, grid_, (const float4 *)d_nrm_,
                                          T32_, T64, offset, r2f_, plane ? 1 : 0, (int32_t *)d_idx_,
                                          (float *)d_d2_, (double *)d_partials_, grid_blocks(),
                                          &nblocks, lanes,
                                          prof ? (unsigned long long *)d_cand_ : nullptr, nullptr,
                                          1, 0, stream_, f64_src(), f64_sorted(), r2d_, (const Pt64 *)d_nrm64_,
                                          exact_ ? 1 : 0, fused ? &fa : nullptr, shard_d64(), (Pt64 *)d_pos_, 1, cert_prev()));
            last_kernel_ = pass_kernel(lanes);
            pos_fresh_ = d_pos_ != nullptr;
            note_state_pass(T64);
            if (prof) { HIP_TRY(hipEventRecord(ev_[e0 + 1], stream_)); pending_.push_back({e0, 0}); }
            if (!tshard_ && !fused) {
                if (prof) { e0 = next_event_pair(); HIP_TRY(hipEventRecord(ev_[e0], stream_)); }
                HIP_TRY(launch_finalize((const double *)d_partials_, nblocks, plane ? 1 : 0,
                                        (double *)d_stats_, stream_, pub, seq));
                if (prof) { HIP_TRY(hipEventRecord(ev_[e0 + 1], stream_)); pending_.push_back({e0, 1}); }
            }
            grid_pending_ = false;
        } else {
            if (prof) { e0 = next_event_pair(); HIP_TRY(hipEventRecord(ev_[e0], stream_)); }
            HIP_TRY(launch_reduce((const float4 *)d_src_, ns_, (const float4 *)d_tgt_,
                                  (const float4 *)d_nrm_, (const unsigned long long *)d_keys_,
                                  plan_.tgt_splits, ns_pad_, T32_, T64, offset, r2f_, plane ? 1 : 0,
                                  (int32_t *)d_idx_, (float *)d_d2_, (double *)d_partials_,
                                  reduce_max_blocks(), (double *)d_stats_, nullptr, nullptr, stream_,
                                  tshard_ ? nullptr : pub, seq, bex_ptr()));
            if (prof) { HIP_TRY(hipEventRecord(ev_[e0 + 1], stream_)); pending_.push_back({e0, 1}); }
            brute_reduced_ = true;
        }
        if (tshard_) {
            // The local pass above found this shard's winner of every source point.  The
            // global winner is the smallest (d2, global index) key over the ranks; its owner
            // accumulates the pair, so each correspondence is counted exactly once.
            int rc = shard_exchange(T64, plane, offset, pub, seq);
            if (rc) return rc;
        }
        if (ipc && !ipc_done) {
            // ONE exchange of the 38 f64 accumulators per ICP iteration: remote stores into the peers'
            // mailboxes over xGMI, rank-ordered sum, publication to the host -- one tiny launch
            HIP_TRY(launch_ipc_allreduce((const double *)d_stats_, (double *)d_stats_, peers_, ipc_rank_, ipc_n_,
                                         ipc_seq_dev(), h_stats_dev_, seq, (int *)d_ipc_flag_, stream_));
        } else if (comm_) {
            // ONE all-reduce of the 38 f64 accumulators per ICP iteration
            int rc = g_rccl.AllReduce(d_stats_, d_stats_, kNStats, kNcclFloat64, kNcclSum, comm_, stream_);
            if (rc != 0) {
                err_ = std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
                return VISMA_ICP_ERR_RCCL;
            }
        }
        // publish to mapped host memory and spin on the sequence word (no DMA
        // packet, no interrupt wake-up: ~10 us less per iteration than memcpy+sync)
        if (comm_ && !ipc) HIP_TRY(launch_publish_stats((const double *)d_stats_, h_stats_dev_, seq, stream_));
        // every granule carries the sequence number it was written for
        volatile unsigned long long *g = reinterpret_cast<volatile unsigned long long *>(h_stats_);
        auto all_tagged = [&]() {
            for (int i = kNStats - 1; i >= 0; --i)
                if (g[2 * i + 1] != seq) return false;
            return true;
        };
        bool seen = false;
        for (long long spin = 0; spin < 400000000ll; ++spin) {
            if (all_tagged()) { seen = true; break; }
            if ((spin & 0xFFFFF) == 0xFFFFF && hipStreamQuery(stream_) != hipErrorNotReady) {
                seen = all_tagged();
                break;
            }
        }
        if (!seen) {
            HIP_TRY(hipStreamSynchronize(stream_));   // surfaces a kernel fault, if any
            if (!all_tagged()) {
                int flag = 0;
                if (d_ipc_flag_) (void)hipMemcpy(&flag, d_ipc_flag_, sizeof(int), hipMemcpyDeviceToHost);
                err_ = flag ? "all-reduce: rank " + std::to_string(flag - 1) + " never delivered its statistics"
                            : std::string("statistics were not published");
                return VISMA_ICP_ERR_HIP;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        for (int i = 0; i < kNStats; i++) {
            const unsigned long long v = g[2 * i];
            std::memcpy(&stats[i], &v, sizeof(double));
        }
        return maybe_collect_timing();
    }

    int get_correspondences(int32_t *idx, float *d2) override
    {
        HIP_TRY(hipSetDevice(device_));
        if (!have_pass_) { err_ = "no nn_pass yet"; return VISMA_ICP_ERR_STATE; }
#ifdef VISMA_WITH_TILE
        if (use_grid_ && grid_pending_ && use_tile()) {
            // nn_pass without a reduction: run the fused kernel for its index output
            const int cfg = tile_config(ns_);
            const int nblocks = tile_blocks(ns_, cfg);
            int rc = ensure_tile_buffers((size_t)nblocks, 1);
            if (rc) return rc;
            TileArgs ta = tile_args(T64_last_, nullptr, false);
            ta.bpp = nblocks;
            HIP_TRY(launch_nn_tile_reduce(ta, 0, cfg, nblocks, stream_));
            grid_pending_ = false;
        } else
#endif
        if (use_grid_ && grid_pending_) {
            // nn_pass without a reduction: run the fused kernel for its index output
            const Xform64 T64 = T64_last_;
            int nblocks = 1;
            HIP_TRY(launch_nn_grid_reduce((const float4 *)d_src_, ns_, search_sorted(),
                                          (const unsigned *)d_start_, grid_, (const float4 *)d_nrm_,
                                          T32_, T64, nullptr, r2f_, 0, (int32_t *)d_idx_,
                                          (float *)d_d2_, (double *)d_partials_, reduce_max_blocks(),
                                          &nblocks, (last_kernel_ = pass_kernel(pass_lanes()), pass_lanes()), nullptr, nullptr, 1, 0, stream_,
                                          f64_src(), f64_sorted(), r2d_, (const Pt64 *)d_nrm64_, exact_ ? 1 : 0,
                                          nullptr, nullptr, (Pt64 *)d_pos_, 1, cert_prev()));
            pos_fresh_ = d_pos_ != nullptr;
            note_state_pass(T64);
            grid_pending_ = false;
        } else if (!use_grid_ && !brute_reduced_) {
            // brute-force pass without a reduction yet: the index is recovered by
            // the reduction kernel, run it for its index output
            const Xform64 T64 = T64_last_;
            HIP_TRY(launch_reduce((const float4 *)d_src_, ns_, (const float4 *)d_tgt_,
                                  (const float4 *)d_nrm_, (const unsigned long long *)d_keys_,
                                  plan_.tgt_splits, ns_pad_, T32_, T64, nullptr, r2f_, 0,
                                  (int32_t *)d_idx_, (float *)d_d2_, (double *)d_partials_,
                                  reduce_max_blocks(), (double *)d_stats_, nullptr, nullptr, stream_, nullptr, 0, bex_ptr()));
            brute_reduced_ = true;
        }
        HIP_TRY(hipStreamSynchronize(stream_));
        if (ns_ > 0) {
            HIP_TRY(hipMemcpy(idx, (int32_t *)d_idx_ + view_offset_, sizeof(int32_t) * ns_, hipMemcpyDeviceToHost));
            if (d2) HIP_TRY(hipMemcpy(d2, (float *)d_d2_ + view_offset_, sizeof(float) * ns_, hipMemcpyDeviceToHost));
        }
        return VISMA_ICP_OK;
    }

    bool supports_device_loop() const override { return true; }

    void select_problem(int b) override { view_offset_ = (int64_t)b * loop_out_stride_; }

    int run_loop(const LoopParams &lp, const Mat4 *Tc0s, int nprob, LoopResult *out) override
    {
        HIP_TRY(hipSetDevice(device_));
        if (nprob < 1) { err_ = "nprob < 1"; return VISMA_ICP_ERR_INVALID; }
        if (!d_src_ || !d_tgt_) { err_ = "clouds not set"; return VISMA_ICP_ERR_STATE; }
        if (lp.plane && !d_nrm_) { err_ = "point-to-plane needs target normals"; return VISMA_ICP_ERR_STATE; }
        int rc = choose_mode(lp.max_dist);
        if (rc) return rc;
        // Many problems advancing together fill the chip whatever the cloud size: AUTO then
        // takes the grid even for a target too small to pay off for ONE problem (a sweep over
        // a 3 k-point target fell back to 24 sequential brute-force loops: 25 ms instead of 2).
        if (nprob > 1 && !use_grid_ && nn_mode_ == VISMA_ICP_NN_AUTO && grid_valid_ && nt_ > 0) use_grid_ = true;
        r2f_ = (float)(lp.max_dist * lp.max_dist);
        r2d_ = (double)r2f_;
        for (int i = 0; i < 12; i++) T32_.m[i] = (float)lp.Tc0.m[i];
        if (nprob > 1 && (!use_grid_ || comm_ || ipc_n_ > 1)) {
            err_ = "batched loop needs the grid search on a single GPU";
            return VISMA_ICP_ERR_STATE;
        }
        if (tshard_ && !(shard_loop_on_device() && use_grid_)) {
            err_ = "the device loop of a target shard needs the library's RCCL communicator, f64 clouds and the grid search";
            return VISMA_ICP_ERR_STATE;
        }
        const int64_t ns_rounded = ((ns_ + kBlock - 1) / kBlock) * kBlock;
        view_offset_ = 0;
        loop_out_stride_ = ns_rounded;
        last_mode_ = grid_search_mode();
        if (use_grid_) {
            rc = ensure_aux(ns_rounded * nprob);
            if (rc) return rc;
            const size_t rows = (size_t)reduce_max_blocks() * nprob;
            if (rows > partial_rows_) {
                free_dev(d_partials_);
                HIP_TRY(hipMalloc(&d_partials_, sizeof(double) * kReduceAcc * rows));
                partial_rows_ = rows;
            }
        } else {
            plan_ = nn_plan(ns_, nt_pad_);
            ns_pad_ = (int64_t)plan_.src_tiles * kBlock * plan_.spt;
            const size_t need = sizeof(unsigned long long) * (size_t)ns_pad_ * plan_.tgt_splits;
            if (need > keys_bytes_) {
                free_dev(d_keys_);
                HIP_TRY(hipMalloc(&d_keys_, need));
                keys_bytes_ = need;
            }
            rc = ensure_aux(ns_pad_);
            if (rc) return rc;
            if (brute_exact()) { rc = ensure_second(ns_pad_, plan_.tgt_splits); if (rc) return rc; }
        }
        if (nprob > state_cap_) {
            free_dev(d_state_);
            if (h_state_) { (void)hipHostFree(h_state_); h_state_ = nullptr; }
            HIP_TRY(hipMalloc(&d_state_, sizeof(DevIcpState) * nprob));
            HIP_TRY(hipHostMalloc((void **)&h_state_, sizeof(DevIcpState) * nprob, hipHostMallocDefault));
            state_cap_ = nprob;
        }
        for (int b = 0; b < nprob; b++) {
            DevIcpState &h = h_state_[b];
            std::memset(&h, 0, sizeof(h));
            const Mat4 &T0 = Tc0s ? Tc0s[b] : lp.Tc0;
            for (int i = 0; i < 12; i++) h.Tc[i] = T0.m[i];
            for (int a = 0; a < 3; a++) h.centre[a] = lp.centre[a];
            h.rel_fit = lp.rel_fit; h.rel_rmse = lp.rel_rmse;
            h.ns_total = lp.ns_total > 0 ? lp.ns_total : ns_;
            h.active = 1;
            h.max_iter = lp.max_iter; h.solver = lp.solver; h.scaling = lp.scaling ? 1 : 0;
            h.plane = lp.plane ? 1 : 0; h.world_frame = lp.world ? 1 : 0;
            h.check_stop = lp.check_stop ? 1 : 0;
            h.r2f = r2f_;
        }
        HIP_TRY(hipMemcpyAsync(d_state_, h_state_, sizeof(DevIcpState) * nprob, hipMemcpyHostToDevice, stream_));
        DevIcpState *st = (DevIcpState *)d_state_;
        const Xform64 T64{};   // ignored: the kernels read the transform from the state
        const int plane = lp.plane ? 1 : 0;
        // with a stop test the host looks at the state every `chunk` passes; launches
        // after convergence are no-ops (the kernels return on !active)
        const int chunk = lp.check_stop ? 8 : lp.passes;
        int done = 0;
        while (done < lp.passes) {
            const int n = std::min(chunk, lp.passes - done);
            for (int j = 0; j < n; j++) {
                int nblocks = 1, e0 = -1;
                if (profiling_) { e0 = next_event_pair(); HIP_TRY(hipEventRecord(ev_[e0], stream_)); }
                bool fused = false;
                if (use_grid_) {
                    // fold inside the search launch: the statistics land in the problems' device state
                    FoldArgs fa{};
                    fused = fused_fold_ != 0 && !tshard_;       // (target shards fold after their exchange)
                    const int lanes = pass_lanes(nprob);
                    if (fused) {
                        rc = make_fold(grid_launch_blocks(ns_, lanes, reduce_max_blocks()), nprob,
                                       st->stats, (long long)(sizeof(DevIcpState) / sizeof(double)), nullptr, 0, &fa);
                        if (rc) return rc;
                        if (ipc_n_ > 1) add_ipc(&fa);      // (one problem per rank: ipc needs nprob == 1)
                    }
                    HIP_TRY(launch_nn_grid_reduce((const float4 *)d_src_, ns_, search_sorted(),
                                                  (const unsigned *)d_start_, grid_, (const float4 *)d_nrm_,
                                                  T32_, T64, nullptr, r2f_, plane, (int32_t *)d_idx_,
                                                  (float *)d_d2_, (double *)d_partials_,
                                                  reduce_max_blocks(), &nblocks, lanes,
                                                  profiling_ ? (unsigned long long *)d_cand_ : nullptr, st,
                                                  nprob, loop_out_stride_, stream_, f64_src(), f64_sorted(), r2d_, (const Pt64 *)d_nrm64_,
                                                  exact_ ? 1 : 0, fused ? &fa : nullptr, tshard_ ? shard_d64() : nullptr,
                                                  (Pt64 *)d_pos_, cert_enabled_ ? 1 : (1 | 8)));
                    last_kernel_ = pass_kernel(lanes);
                    pos_fresh_ = d_pos_ != nullptr;
                    prev_T_valid_ = false;                   // (the state's pose now lives in the device loop's state)
                    if (tshard_) {
                        // the shards' winners compared on the stream (two MIN all-reduces), the owners' moments
                        // into the partial rows: everything stream-ordered, the host is not involved
                        rc = shard_exchange_on_stream(st, plane, &nblocks);
                        if (rc) return rc;
                    }
                } else {
                    HIP_TRY(launch_nn_brute((const float4 *)d_src_, ns_, (const float4 *)d_tgt_, nt_pad_,
                                            T32_, r2f_, (unsigned long long *)d_keys_, ns_pad_, plan_, st,
                                            stream_, brute_exact() ? (const Pt64 *)d_src64_ : nullptr, nullptr,
                                            brute_exact() ? (float *)d_second_ : nullptr));
                }
                if (profiling_) { HIP_TRY(hipEventRecord(ev_[e0 + 1], stream_)); pending_.push_back({e0, 0}); }
                if (profiling_) { e0 = next_event_pair(); HIP_TRY(hipEventRecord(ev_[e0], stream_)); }
                if (!use_grid_) {
                    HIP_TRY(launch_reduce((const float4 *)d_src_, ns_, (const float4 *)d_tgt_,
                                          (const float4 *)d_nrm_, (const unsigned long long *)d_keys_,
                                          plan_.tgt_splits, ns_pad_, T32_, T64, nullptr, r2f_, plane,
                                          (int32_t *)d_idx_, (float *)d_d2_, (double *)d_partials_,
                                          reduce_max_blocks(), nullptr, st, &nblocks, stream_, nullptr, 0, bex_ptr()));
                }
                if (ipc_n_ > 1) {
                    if (!fused) HIP_TRY(launch_finalize_state((const double *)d_partials_, nblocks, st, plane, stream_));
                    if (!(fused && use_grid_))               // (fused: the folding workgroup exchanged already)
                        HIP_TRY(launch_ipc_allreduce(st->stats, st->stats, peers_, ipc_rank_, ipc_n_, ipc_seq_dev(), nullptr, 0,
                                                     (int *)d_ipc_flag_, stream_));
                    HIP_TRY(launch_solve_state(st, 1, stream_));
                } else if (comm_) {
                    if (!fused) HIP_TRY(launch_finalize_state((const double *)d_partials_, nblocks, st, plane, stream_));
                    // ONE all-reduce of the 38 f64 accumulators per ICP iteration
                    int nrc = g_rccl.AllReduce(st->stats, st->stats, kNStats, kNcclFloat64, kNcclSum, comm_, stream_);
                    if (nrc != 0) {
                        err_ = std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(nrc) : "error");
                        return VISMA_ICP_ERR_RCCL;
                    }
                    HIP_TRY(launch_solve_state(st, 1, stream_));
                } else if (fused) {
                    HIP_TRY(launch_solve_state(st, nprob, stream_));
                } else {
                    HIP_TRY(launch_finalize_solve((const double *)d_partials_, nblocks, st, plane, nprob, stream_));
                }
                if (profiling_) { HIP_TRY(hipEventRecord(ev_[e0 + 1], stream_)); pending_.push_back({e0, 1}); }
            }
            done += n;
            HIP_TRY(hipMemcpyAsync(h_state_, d_state_, sizeof(DevIcpState) * nprob, hipMemcpyDeviceToHost, stream_));
            int ipc_flag = 0;
            if (ipc_n_ > 1) HIP_TRY(hipMemcpyAsync(&ipc_flag, d_ipc_flag_, sizeof(int), hipMemcpyDeviceToHost, stream_));
            HIP_TRY(hipStreamSynchronize(stream_));
            if (ipc_flag) {
                err_ = "all-reduce: rank " + std::to_string(ipc_flag - 1) + " never delivered its statistics";
                return VISMA_ICP_ERR_HIP;
            }
            rc = maybe_collect_timing();
            if (rc) return rc;
            bool any = false;
            for (int b = 0; b < nprob; b++) any = any || h_state_[b].active;
            if (!any) break;
        }
        for (int b = 0; b < nprob; b++) {
            const DevIcpState &h = h_state_[b];
            out[b].Tc = Mat4::identity();
            for (int i = 0; i < 12; i++) out[b].Tc.m[i] = h.Tc[i];
            out[b].fit = h.fit; out[b].rmse = h.rmse;
            out[b].k = (int64_t)std::llround(h.K);
            out[b].iters = h.iter; out[b].passes = h.passes;
        }
        for (int i = 0; i < 12; i++) T32_.m[i] = (float)h_state_[0].Tc[i];
        have_pass_ = true;
        grid_pending_ = false;
        brute_reduced_ = true;
        return VISMA_ICP_OK;
    }

    int run_loop_batch(const LoopParams &lp, const std::vector<BatchProblem> &pb, LoopResult *out) override
    {
        HIP_TRY(hipSetDevice(device_));
        const int B = (int)pb.size();
        if (B < 1) return VISMA_ICP_OK;
        if (comm_) { err_ = "batched loop is single-GPU"; return VISMA_ICP_ERR_STATE; }
        StageTrace tr("batch/engine");
        // ---- layout of the concatenated arrays
        std::vector<ProbDesc> descs((size_t)B);
        int64_t src_tot = 0, tgt_tot = 0, cell_tot = 0, max_ncell = 0, out_tot = 0;
        int total_blocks = 0;
        // lanes per query / loads in flight (G + 100 U); VISMA_ICP_BATCH_LANES overrides
        // (measured on config 3, 288 problems / 5.2 M queries per pass: G=1,U=8 21.6 ms, G=4,U=8 34 ms;
        // few small problems need the lanes of G=4 to fill the chip)
        int64_t queries = 0;
        for (int b = 0; b < B; b++) queries += pb[b].ns;
        int lanes = queries >= 200000 ? 801 : 804;
        {
            bool all64 = B > 0;
            for (int b = 0; b < B; b++) all64 = all64 && (pb[b].src64 || pb[b].src_share >= 0 || pb[b].ns == 0);
            // the f64 search gathers 32-byte candidates (measured on the yaw sweeps); the exact search ranks
            // 16-byte ones (config 3: 801 524 k it/s, 1201 458 k, 402 418 k, 802 351 k, 804 216 k)
            if (all64 && !exact_) lanes = queries >= 200000 ? 402 : 804;
        }
        if (const char *e = std::getenv("VISMA_ICP_BATCH_LANES")) { const int v = std::atoi(e); if (v > 0) lanes = v; }
        const int G = lanes % 100;
        if (G < 1 || G > 64 || (G & (G - 1))) { err_ = "bad VISMA_ICP_BATCH_LANES"; return VISMA_ICP_ERR_INVALID; }
        bool one_per_lane = true;
        for (int b = 0; b < B; b++) {
            const BatchProblem &q = pb[b];
            if (q.ns < 0 || q.nt < 0 || !(q.max_dist > 0.0)) { err_ = "bad batch problem"; return VISMA_ICP_ERR_INVALID; }
            ProbDesc &d = descs[b];
            std::memset(&d, 0, sizeof(d));
            // a small cloud does not get a huge cell table: cap the grid, h grows (still exact)
            const int64_t cap = std::min<int64_t>(kGridMaxCells, std::max<int64_t>(4096, 8 * q.nt));
            float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
            if (q.nt > 0) for (int a = 0; a < 3; a++) { mn[a] = q.bb_min[a]; mx[a] = q.bb_max[a]; }
            if ((q.src_share >= 0 && (q.src_share >= b || pb[q.src_share].ns != q.ns || pb[q.src_share].src_share >= 0)) ||
                (q.grid_share >= 0 && (q.grid_share >= b || pb[q.grid_share].nt != q.nt || pb[q.grid_share].grid_share >= 0))) {
                err_ = "bad cloud sharing in the batch";
                return VISMA_ICP_ERR_INVALID;
            }
            if (q.grid_share >= 0) {
                d.g = descs[q.grid_share].g;
                d.sorted_off = descs[q.grid_share].sorted_off;
                d.start_off = descs[q.grid_share].start_off;
            } else {
                d.g = grid_plan(mn, mx, q.max_dist, cap);
                d.sorted_off = tgt_tot;
                d.start_off = cell_tot;
                tgt_tot += q.nt;
                cell_tot += d.g.ncell + 1;
                max_ncell = std::max(max_ncell, d.g.ncell);
            }
            if (q.src_share >= 0) {
                d.src_off = descs[q.src_share].src_off;
            } else {
                d.src_off = src_tot;
                src_tot += q.ns;
            }
            d.out_off = out_tot;
            out_tot += q.ns;
            d.ns = (int)q.ns;
            d.first_block = total_blocks;
            // one query per lane (the kernel's ONE variant) up to 262,144 source points per problem
            int64_t nb = (q.ns + kBlock - 1) / kBlock;
            if (nb < 1) nb = 1;
            if (nb > 1024) { nb = 1024; one_per_lane = false; }
            d.nblocks = (int)nb;
            total_blocks += d.nblocks;
        }
        // ---- device buffers
        if (src_tot > bt_src_cap_) {
            free_dev(bt_src_);
            HIP_TRY(hipMalloc(&bt_src_, sizeof(float4) * std::max<int64_t>(src_tot, 1)));
            bt_src_cap_ = src_tot;
        }
        bool f64 = B > 0;
        for (int b = 0; b < B; b++) {
            const BatchProblem &q = pb[b];
            const BatchProblem &sq = q.src_share >= 0 ? pb[q.src_share] : q, &tq = q.grid_share >= 0 ? pb[q.grid_share] : q;
            f64 = f64 && (q.ns == 0 || sq.src64) && (q.nt == 0 || tq.tgt64 || (tq.tgt_raw && tq.tgt_f64));
        }
        if (f64 && (src_tot > bt_src64_cap_ || tgt_tot > bt_tgt64_cap_)) {
            free_dev(bt_src64_); free_dev(bt_tgt64_); free_dev(bt_sorted64_);
            HIP_TRY(hipMalloc(&bt_src64_, sizeof(Pt64) * std::max<int64_t>(src_tot, 1)));
            HIP_TRY(hipMalloc(&bt_tgt64_, sizeof(Pt64) * std::max<int64_t>(tgt_tot, 1)));
            HIP_TRY(hipMalloc(&bt_sorted64_, sizeof(Pt64) * std::max<int64_t>(tgt_tot, 1)));
            bt_src64_cap_ = src_tot; bt_tgt64_cap_ = tgt_tot;
        }
        if (lp.plane) {
            for (int b = 0; b < B; b++) {
                const BatchProblem &tq = pb[b].grid_share >= 0 ? pb[pb[b].grid_share] : pb[b];
                if (pb[b].nt > 0 && !(f64 ? (const void *)tq.nrm64 : (const void *)tq.nrm_xyzw)) {
                    err_ = "point-to-plane batch without target normals";
                    return VISMA_ICP_ERR_STATE;
                }
            }
            if (f64 && !exact_) { err_ = "point-to-plane batches run the exact or the fp32 search"; return VISMA_ICP_ERR_STATE; }
            if (f64 && tgt_tot > bt_nrm64_cap_) {
                free_dev(bt_nrm64_);
                HIP_TRY(hipMalloc(&bt_nrm64_, sizeof(Pt64) * std::max<int64_t>(tgt_tot, 1)));
                bt_nrm64_cap_ = tgt_tot;
            }
            if (!f64 && tgt_tot > bt_nrm_cap_) {
                free_dev(bt_nrm_);
                HIP_TRY(hipMalloc(&bt_nrm_, sizeof(float4) * std::max<int64_t>(tgt_tot, 1)));
                bt_nrm_cap_ = tgt_tot;
            }
        }
        if (out_tot > bt_out_cap_) {
            free_dev(bt_idx_); free_dev(bt_d2_); free_dev(bt_pos_);
            HIP_TRY(hipMalloc(&bt_idx_, sizeof(int32_t) * std::max<int64_t>(out_tot, 1)));
            HIP_TRY(hipMalloc(&bt_d2_, sizeof(float) * std::max<int64_t>(out_tot, 1)));
            HIP_TRY(hipMalloc(&bt_pos_, sizeof(Pt64) * std::max<int64_t>(out_tot, 1)));
            bt_out_cap_ = out_tot;
        }
        tr.mark("layout, buffers");
        // (new problems: no previous winners)
        HIP_TRY(hipMemsetAsync(bt_pos_, 0xFF, sizeof(Pt64) * (size_t)std::max<int64_t>(out_tot, 1), stream_));
        {
            bool any_raw = false;
            for (int b = 0; b < B; b++) any_raw = any_raw || pb[b].tgt_raw != nullptr;
            if (any_raw && (size_t)tgt_tot * 24 > bt_raw_bytes_) {
                free_dev(bt_raw_);
                HIP_TRY(hipMalloc(&bt_raw_, (size_t)std::max<int64_t>(tgt_tot, 1) * 24));
                bt_raw_bytes_ = (size_t)tgt_tot * 24;
            }
        }
        if (tgt_tot > bt_tgt_cap_) {
            free_dev(bt_tgt_); free_dev(bt_sorted_); free_dev(bt_cell_of_);
            HIP_TRY(hipMalloc(&bt_tgt_, sizeof(float4) * std::max<int64_t>(tgt_tot, 1)));
            HIP_TRY(hipMalloc(&bt_sorted_, sizeof(float4) * (std::max<int64_t>(tgt_tot, 1) + kSortedSlack)));
            HIP_TRY(hipMalloc(&bt_cell_of_, 2 * sizeof(unsigned) * std::max<int64_t>(tgt_tot, 1)));   // (cell, rank)
            bt_tgt_cap_ = tgt_tot;
        }
        if (cell_tot > bt_cell_cap_) {
            free_dev(bt_count_); free_dev(bt_start_);
            HIP_TRY(hipMalloc(&bt_count_, sizeof(unsigned) * cell_tot));
            HIP_TRY(hipMalloc(&bt_start_, sizeof(unsigned) * (cell_tot + 8)));      // (16-byte reads near the end)
            HIP_TRY(hipMemsetAsync(bt_start_, 0, sizeof(unsigned) * (cell_tot + 8), stream_));
            bt_cell_cap_ = cell_tot;
        }
        if (grid_scan_blocks(max_ncell) + 1 > bt_bsum_cap_) {
            free_dev(bt_bsum_);
            bt_bsum_cap_ = grid_scan_blocks(max_ncell) + 1;
            HIP_TRY(hipMalloc(&bt_bsum_, sizeof(unsigned) * bt_bsum_cap_));
        }
        // descriptors, followed by the workgroup -> problem map (one word per workgroup: the warm kernel reads its
        // problem with one load instead of a binary search over the descriptors -- ~9 dependent scalar loads per wave)
        const size_t desc_bytes = sizeof(ProbDesc) * (size_t)B + sizeof(int) * (size_t)std::max(total_blocks, 1);
        if (desc_bytes > bt_desc_cap_) {
            free_dev(bt_descs_);
            HIP_TRY(hipMalloc(&bt_descs_, desc_bytes));
            bt_desc_cap_ = desc_bytes;
        }
        if ((size_t)total_blocks > partial_rows_) {
            free_dev(d_partials_);
            HIP_TRY(hipMalloc(&d_partials_, sizeof(double) * kReduceAcc * (size_t)total_blocks));
            partial_rows_ = (size_t)total_blocks;
        }
        if (B > state_cap_) {
            free_dev(d_state_);
            if (h_state_) { (void)hipHostFree(h_state_); h_state_ = nullptr; }
            HIP_TRY(hipMalloc(&d_state_, sizeof(DevIcpState) * B));
            HIP_TRY(hipHostMalloc((void **)&h_state_, sizeof(DevIcpState) * B, hipHostMallocDefault));
            state_cap_ = B;
        }
        last_mode_ = f64 ? (exact_ ? 1 : 2) : 0;
        // ---- uploads + per-problem grid builds (stream ordered, no host sync)
        int e0 = -1;
        if (profiling_) { e0 = next_event_pair(); HIP_TRY(hipEventRecord(ev_[e0], stream_)); }
        for (int b = 0; b < B; b++) {
            const BatchProblem &q = pb[b];
            const ProbDesc &d = descs[b];
            if (q.ns > 0 && q.src_share < 0) HIP_TRY(hipMemcpyAsync((float4 *)bt_src_ + d.src_off, q.src_xyzw, sizeof(float4) * q.ns, hipMemcpyHostToDevice, stream_));
            if (f64 && q.ns > 0 && q.src_share < 0)
                HIP_TRY(hipMemcpyAsync((Pt64 *)bt_src64_ + d.src_off, q.src64, sizeof(Pt64) * q.ns, hipMemcpyHostToDevice, stream_));
            if (q.grid_share >= 0) continue;
            if (q.nt > 0 && q.tgt_raw) {
                double *raw = (double *)bt_raw_ + 3 * d.sorted_off;
                HIP_TRY(hipMemcpyAsync(raw, q.tgt_raw, sizeof(double) * 3 * q.nt, hipMemcpyHostToDevice, stream_));
                HIP_TRY(launch_expand_f64(raw, q.nt, q.centre, (float4 *)bt_tgt_ + d.sorted_off,
                                          f64 ? (Pt64 *)bt_tgt64_ + d.sorted_off : nullptr, stream_));
            } else if (q.nt > 0) {
                HIP_TRY(hipMemcpyAsync((float4 *)bt_tgt_ + d.sorted_off, q.tgt_xyzw, sizeof(float4) * q.nt, hipMemcpyHostToDevice, stream_));
                if (f64)
                    HIP_TRY(hipMemcpyAsync((Pt64 *)bt_tgt64_ + d.sorted_off, q.tgt64, sizeof(Pt64) * q.nt, hipMemcpyHostToDevice, stream_));
            }
            if (lp.plane && q.nt > 0) {
                if (f64) HIP_TRY(hipMemcpyAsync((Pt64 *)bt_nrm64_ + d.sorted_off, q.nrm64, sizeof(Pt64) * q.nt, hipMemcpyHostToDevice, stream_));
                else HIP_TRY(hipMemcpyAsync((float4 *)bt_nrm_ + d.sorted_off, q.nrm_xyzw, sizeof(float4) * q.nt, hipMemcpyHostToDevice, stream_));
            }
            HIP_TRY(launch_grid_build((const float4 *)bt_tgt_ + d.sorted_off, q.nt, d.g,
                                      (unsigned *)bt_cell_of_ + 2 * d.sorted_off, (unsigned *)bt_count_ + d.start_off,
                                      (unsigned *)bt_bsum_, (unsigned *)bt_start_ + d.start_off,
                                      (float4 *)bt_sorted_ + d.sorted_off, stream_,
                                      f64 ? (const Pt64 *)bt_tgt64_ + d.sorted_off : nullptr,
                                      f64 ? (Pt64 *)bt_sorted64_ + d.sorted_off : nullptr));
        }
        const bool packed = f64 && exact_;                   // the exact search ranks on packed (x,y,z) triples
        if (packed) {
            if (tgt_tot > bt_sorted12_cap_) {
                free_dev(bt_sorted12_);
                HIP_TRY(hipMalloc(&bt_sorted12_, sizeof(float) * 3 * (size_t)(std::max<int64_t>(tgt_tot, 1) + kSortedSlack)));
                bt_sorted12_cap_ = tgt_tot;
            }
            HIP_TRY(launch_pack12((const float4 *)bt_sorted_, (float *)bt_sorted12_, tgt_tot, stream_));
        }
        bt_desc_host_.resize(desc_bytes);
        std::memcpy(bt_desc_host_.data(), descs.data(), sizeof(ProbDesc) * (size_t)B);
        {
            int *map = reinterpret_cast<int *>(bt_desc_host_.data() + sizeof(ProbDesc) * (size_t)B);
            for (int b = 0; b < B; b++)
                for (int k = 0; k < descs[b].nblocks; k++) map[descs[b].first_block + k] = b;
        }
        HIP_TRY(hipMemcpyAsync(bt_descs_, bt_desc_host_.data(), desc_bytes, hipMemcpyHostToDevice, stream_));
        for (int b = 0; b < B; b++) {
            DevIcpState &h = h_state_[b];
            std::memset(&h, 0, sizeof(h));
            for (int i = 0; i < 12; i++) h.Tc[i] = pb[b].Tc0.m[i];
            for (int a = 0; a < 3; a++) h.centre[a] = pb[b].centre[a];
            h.rel_fit = lp.rel_fit; h.rel_rmse = lp.rel_rmse;
            h.ns_total = pb[b].ns;
            h.active = 1;
            h.max_iter = lp.max_iter; h.solver = lp.solver; h.scaling = lp.scaling ? 1 : 0;
            h.plane = lp.plane ? 1 : 0; h.world_frame = lp.world ? 1 : 0; h.check_stop = lp.check_stop ? 1 : 0;
            h.r2f = (float)(pb[b].max_dist * pb[b].max_dist);
        }
        HIP_TRY(hipMemcpyAsync(d_state_, h_state_, sizeof(DevIcpState) * B, hipMemcpyHostToDevice, stream_));
        // the staging memory of the caller must stay valid until the copies are done
        tr.mark("uploads, grids enqueued");
        HIP_TRY(hipStreamSynchronize(stream_));
        tr.mark("... and finished");
        if (profiling_) { HIP_TRY(hipEventRecord(ev_[e0 + 1], stream_)); pending_.push_back({e0, 2}); }
        // ---- the loop: one NN launch (search + fold) + one solve launch per pass for ALL problems
        DevIcpState *st = (DevIcpState *)d_state_;
        FoldArgs bfa{};
        if (fused_fold_) {
            int max_nb = 1;
            for (int b = 0; b < B; b++) max_nb = std::max(max_nb, descs[b].nblocks);
            const size_t tstride = 1 + (size_t)(max_nb + 31) / 32;
            int rc2 = ensure_tile_buffers((size_t)total_blocks, tstride * B);
            if (rc2) return rc2;
            bfa.tickets = (unsigned *)d_tickets_;
            bfa.partials2 = (double *)d_partials2_;
            bfa.ticket_stride = (int)tstride;
            bfa.stats_out = st->stats;
            bfa.stats_stride = (long long)(sizeof(DevIcpState) / sizeof(double));
        }
        const int chunk = lp.check_stop ? 8 : lp.passes;
        int done = 0;
        // the first pass prunes progressively (lane-serial kernel), the later ones start from its winners
        const bool coop = coop_enabled_ && packed && std::getenv("VISMA_ICP_BATCH_LANES") == nullptr &&
                          (tgt_tot + kSortedSlack) * 12 < (1ll << 32);
        bool fresh = false;
        while (done < lp.passes) {
            const int n = std::min(chunk, lp.passes - done);
            for (int j = 0; j < n; j++) {
                if (profiling_) { e0 = next_event_pair(); HIP_TRY(hipEventRecord(ev_[e0], stream_)); }
                HIP_TRY(launch_nn_grid_reduce_batch((const float4 *)bt_src_, packed ? (const float4 *)bt_sorted12_ : (const float4 *)bt_sorted_,
                                                    (const unsigned *)bt_start_, (const ProbDesc *)bt_descs_, B,
                                                    total_blocks, (int32_t *)bt_idx_, (float *)bt_d2_,
                                                    (double *)d_partials_, (coop && fresh) ? kCoopLanes : lanes, one_per_lane ? 1 : 0, st, stream_,
                                                    f64 ? (const Pt64 *)bt_src64_ : nullptr,
                                                    f64 ? (const Pt64 *)bt_sorted64_ : nullptr, exact_ ? 1 : 0,
                                                    fused_fold_ ? &bfa : nullptr,
                                                    profiling_ ? (unsigned long long *)d_cand_ : nullptr,
                                                    (lp.plane && !f64) ? (const float4 *)bt_nrm_ : nullptr,
                                                    (lp.plane && f64) ? (const Pt64 *)bt_nrm64_ : nullptr,
                                                    (Pt64 *)bt_pos_, cert_enabled_ ? (1 | 2) : (1 | 2 | 8)));   // warm | workgroup map behind the descriptors
                last_kernel_ = (coop && fresh) ? 2 : 1;
                fresh = true;
                if (profiling_) { HIP_TRY(hipEventRecord(ev_[e0 + 1], stream_)); pending_.push_back({e0, 0}); }
                if (profiling_) { e0 = next_event_pair(); HIP_TRY(hipEventRecord(ev_[e0], stream_)); }
                if (fused_fold_) HIP_TRY(launch_solve_state(st, B, stream_));
                else HIP_TRY(launch_finalize_solve_batch((const double *)d_partials_, (const ProbDesc *)bt_descs_, st, B, stream_, lp.plane ? 1 : 0));
                if (profiling_) { HIP_TRY(hipEventRecord(ev_[e0 + 1], stream_)); pending_.push_back({e0, 1}); }
            }
            done += n;
            HIP_TRY(hipMemcpyAsync(h_state_, d_state_, sizeof(DevIcpState) * B, hipMemcpyDeviceToHost, stream_));
            HIP_TRY(hipStreamSynchronize(stream_));
            int rc = maybe_collect_timing();
            if (rc) return rc;
            bool any = false;
            for (int b = 0; b < B; b++) any = any || h_state_[b].active;
            if (!any) break;
        }
        tr.mark("passes");
        for (int b = 0; b < B; b++) {
            const DevIcpState &h = h_state_[b];
            out[b].Tc = Mat4::identity();
            for (int i = 0; i < 12; i++) out[b].Tc.m[i] = h.Tc[i];
            out[b].fit = h.fit; out[b].rmse = h.rmse;
            out[b].k = (int64_t)std::llround(h.K);
            out[b].iters = h.iter; out[b].passes = h.passes;
        }
        return VISMA_ICP_OK;
    }

    int set_nn_mode(int mode) override
    {
        if (mode != VISMA_ICP_NN_AUTO && mode != VISMA_ICP_NN_BRUTE && mode != VISMA_ICP_NN_GRID) {
            err_ = "unknown nn mode";
            return VISMA_ICP_ERR_INVALID;
        }
        nn_mode_ = mode;
        return VISMA_ICP_OK;
    }
    int nn_mode_used() const override { return use_grid_ ? VISMA_ICP_NN_GRID : VISMA_ICP_NN_BRUTE; }
    int search_kernel_used() const override { return use_grid_ ? last_kernel_ : 0; }
    int forget_winners() override { HIP_TRY(hipSetDevice(device_)); return invalidate_pos(); }

    int set_target_shard(int64_t offset, int64_t global_nt) override
    {
        if (offset < 0 || global_nt < 0 || global_nt > 0x7fffffffll) {
            err_ = "bad target shard (global indices must fit 31 bits)";
            return VISMA_ICP_ERR_INVALID;
        }
        tshard_ = global_nt > 0;
        tgt_offset_ = offset;
        tgt_global_ = global_nt;
        return VISMA_ICP_OK;
    }
    void set_minreduce(visma_icp_minreduce_fn fn, void *user) override { minreduce_ = fn; minreduce_user_ = user; }

    int shard_exchange(const Xform64 &T64, bool plane, const double offset[3], double *pub, unsigned long long seq)
    {
        if (tgt_offset_ + nt_ > tgt_global_) { err_ = "target shard exceeds the global target"; return VISMA_ICP_ERR_INVALID; }
        if (!comm_ && !minreduce_) { err_ = "target-sharded mode needs visma_icp_comm_init or visma_icp_set_minreduce"; return VISMA_ICP_ERR_STATE; }
        if (ns_ > gkeys_cap_) {
            free_dev(d_gkeys_); free_dev(d_claim_);
            HIP_TRY(hipMalloc(&d_gkeys_, sizeof(unsigned long long) * std::max<int64_t>(ns_, 1)));
            HIP_TRY(hipMalloc(&d_claim_, sizeof(unsigned long long) * std::max<int64_t>(ns_, 1)));
            gkeys_cap_ = ns_;
        }
        // MIN over the ranks of `keys` (RCCL on the stream, or the host-supplied exchange)
        auto min_reduce = [&](void *keys) -> int {
            if (comm_) {
                int rc = g_rccl.AllReduce(keys, keys, (size_t)ns_, kNcclUint64, kNcclMin, comm_, stream_);
                if (rc != 0) {
                    err_ = std::string("ncclAllReduce(min): ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
                    return VISMA_ICP_ERR_RCCL;
                }
                return VISMA_ICP_OK;
            }
            h_gkeys_.resize((size_t)ns_);
            HIP_TRY(hipMemcpyAsync(h_gkeys_.data(), keys, sizeof(unsigned long long) * ns_, hipMemcpyDeviceToHost, stream_));
            HIP_TRY(hipStreamSynchronize(stream_));
            if (ns_ > 0 && minreduce_(minreduce_user_, (uint64_t *)h_gkeys_.data(), ns_) != 0) {
                err_ = "min-reduce callback failed";
                return VISMA_ICP_ERR_ENGINE;
            }
            HIP_TRY(hipMemcpyAsync(keys, h_gkeys_.data(), sizeof(unsigned long long) * ns_, hipMemcpyHostToDevice, stream_));
            return VISMA_ICP_OK;
        };
        if (shard_d64()) {
            // Every shard ran the EXACT search: compare the shards in f64.  (1) MIN of the f64 d2 bits =
            // the global nearest distance; (2) MIN of the global index over the shards that hold it =
            // lowest index on exact ties, like one GPU; the owner accumulates from the f64 coordinates.
            HIP_TRY(launch_shard_keys64((const int32_t *)d_idx_, (const double *)d_d64_, ns_, (unsigned long long *)d_gkeys_, stream_));
            int rc = min_reduce(d_gkeys_);
            if (rc) return rc;
            HIP_TRY(launch_shard_claim64((const int32_t *)d_idx_, (const double *)d_d64_, (const unsigned long long *)d_gkeys_,
                                         ns_, (unsigned)tgt_offset_, (unsigned long long *)d_claim_, stream_));
            rc = min_reduce(d_claim_);
            if (rc) return rc;
            int nb = 1;
            HIP_TRY(launch_shard_accumulate64((const Pt64 *)d_src64_, ns_, (const unsigned long long *)d_gkeys_,
                                              (const unsigned long long *)d_claim_, (const Pt64 *)d_tgt64_, nt_,
                                              (unsigned)tgt_offset_, (const float4 *)d_nrm_, (const Pt64 *)d_nrm64_, T64,
                                              offset, r2d_, plane ? 1 : 0, (int32_t *)d_idx_, (float *)d_d2_,
                                              (double *)d_partials_, reduce_max_blocks(), &nb, stream_));
            HIP_TRY(launch_finalize((const double *)d_partials_, nb, plane ? 1 : 0, (double *)d_stats_, stream_, pub, seq));
            return VISMA_ICP_OK;
        }
        HIP_TRY(launch_shard_keys((const int32_t *)d_idx_, (const float *)d_d2_, ns_, (unsigned)tgt_offset_,
                                  (unsigned long long *)d_gkeys_, stream_));
        {
            int rc = min_reduce(d_gkeys_);
            if (rc) return rc;
        }
        int nblocks = 1;
        HIP_TRY(launch_shard_accumulate((const float4 *)d_src_, ns_, (const unsigned long long *)d_gkeys_,
                                        (const float4 *)d_tgt_, nt_, (unsigned)tgt_offset_, (const float4 *)d_nrm_,
                                        T64, offset, r2f_, plane ? 1 : 0, (int32_t *)d_idx_, (float *)d_d2_,
                                        (double *)d_partials_, reduce_max_blocks(), &nblocks, stream_));
        HIP_TRY(launch_finalize((const double *)d_partials_, nblocks, plane ? 1 : 0, (double *)d_stats_, stream_, pub, seq));
        return VISMA_ICP_OK;
    }

    // Target shards in the device loop: the f64 protocol of shard_exchange with RCCL's stream-ordered
    // all-reduces; the kernels read transform / frame / radius from the state.
    bool shard_loop_on_device() const override { return tshard_ && comm_ != nullptr && shard_f64_protocol(); }
    int shard_exchange_on_stream(const DevIcpState *st, int plane, int *nblocks)
    {
        if (tgt_offset_ + nt_ > tgt_global_) { err_ = "target shard exceeds the global target"; return VISMA_ICP_ERR_INVALID; }
        if (ns_ > gkeys_cap_) {
            free_dev(d_gkeys_); free_dev(d_claim_);
            HIP_TRY(hipMalloc(&d_gkeys_, sizeof(unsigned long long) * std::max<int64_t>(ns_, 1)));
            HIP_TRY(hipMalloc(&d_claim_, sizeof(unsigned long long) * std::max<int64_t>(ns_, 1)));
            gkeys_cap_ = ns_;
        }
        auto min_reduce = [&](void *keys) -> int {
            int rc = g_rccl.AllReduce(keys, keys, (size_t)ns_, kNcclUint64, kNcclMin, comm_, stream_);
            if (rc != 0) {
                err_ = std::string("ncclAllReduce(min): ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
                return VISMA_ICP_ERR_RCCL;
            }
            return VISMA_ICP_OK;
        };
        HIP_TRY(launch_shard_keys64((const int32_t *)d_idx_, (const double *)d_d64_, ns_, (unsigned long long *)d_gkeys_, stream_));
        int rc = min_reduce(d_gkeys_);
        if (rc) return rc;
        HIP_TRY(launch_shard_claim64((const int32_t *)d_idx_, (const double *)d_d64_, (const unsigned long long *)d_gkeys_,
                                     ns_, (unsigned)tgt_offset_, (unsigned long long *)d_claim_, stream_));
        rc = min_reduce(d_claim_);
        if (rc) return rc;
        const Xform64 T64{};
        HIP_TRY(launch_shard_accumulate64((const Pt64 *)d_src64_, ns_, (const unsigned long long *)d_gkeys_,
                                          (const unsigned long long *)d_claim_, (const Pt64 *)d_tgt64_, nt_,
                                          (unsigned)tgt_offset_, (const float4 *)d_nrm_, (const Pt64 *)d_nrm64_, T64,
                                          nullptr, r2d_, plane, (int32_t *)d_idx_, (float *)d_d2_,
                                          (double *)d_partials_, reduce_max_blocks(), nblocks, stream_, st));
        return VISMA_ICP_OK;
    }

    int comm_init(int rank, int nranks, const void *id) override
    {
        HIP_TRY(hipSetDevice(device_));
        if (!g_rccl.load()) { err_ = g_rccl.error; return VISMA_ICP_ERR_RCCL; }
        NcclId nid;
        std::memcpy(&nid, id, sizeof(nid));
        int rc = g_rccl.CommInitRank(&comm_, nranks, nid, rank);
        if (rc != 0) {
            err_ = std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
            comm_ = nullptr;
            return VISMA_ICP_ERR_RCCL;
        }
        return VISMA_ICP_OK;
    }
    bool has_device_allreduce() const override { return comm_ != nullptr || ipc_n_ > 1; }
    int bind_device() override { HIP_TRY(hipSetDevice(device_)); return VISMA_ICP_OK; }
    hipStream_t aux_stream() override { return stream_; }

    // ---- one-shot all-reduce through IPC-mapped mailboxes (kernels.hip: ipc_allreduce_kernel) ----
    int ensure_mailbox()
    {
        if (d_mbox_) return VISMA_ICP_OK;
        const size_t bytes = sizeof(double) * 2 * kNStats * kIpcMaxRanks * 2;   // two halves, see ipc_allreduce_kernel
        // uncached device memory: remote stores and local polls both go to memory
        if (hipExtMallocWithFlags(&d_mbox_, bytes, hipDeviceMallocUncached) != hipSuccess) {
            (void)hipGetLastError();
            HIP_TRY(hipExtMallocWithFlags(&d_mbox_, bytes, hipDeviceMallocFinegrained));
        }
        HIP_TRY(hipMemset(d_mbox_, 0, bytes));
        HIP_TRY(hipMalloc(&d_ipc_flag_, 16));                    // {int timeout flag, pad, u64 exchange counter}
        HIP_TRY(hipMemset(d_ipc_flag_, 0, 16));
        return VISMA_ICP_OK;
    }
    int ipc_export(void *out) override
    {
        HIP_TRY(hipSetDevice(device_));
        int rc = ensure_mailbox();
        if (rc) return rc;
        // Exporting starts a NEW session: peers only learn the handle after this call, so nothing can be on its
        // way into the mailbox yet -- drop the mappings of an earlier session, clear the granules it left (a
        // retry after a failed handshake must not read them as this session's) and restart the count.
        HIP_TRY(hipStreamSynchronize(stream_));
        for (int r = 0; r < kIpcMaxRanks; r++) {
            if (r != ipc_rank_ && peers_.box[r]) (void)hipIpcCloseMemHandle(peers_.box[r]);
            peers_.box[r] = nullptr;
        }
        ipc_n_ = 0;
        HIP_TRY(hipMemset(d_mbox_, 0, sizeof(double) * 2 * kNStats * kIpcMaxRanks * 2));
        HIP_TRY(hipMemset(d_ipc_flag_, 0, 16));
        hipIpcMemHandle_t h;
        HIP_TRY(hipIpcGetMemHandle(&h, d_mbox_));
        static_assert(sizeof(h) <= VISMA_ICP_IPC_HANDLE_BYTES, "handle size");
        std::memset(out, 0, VISMA_ICP_IPC_HANDLE_BYTES);
        std::memcpy(out, &h, sizeof(h));
        return VISMA_ICP_OK;
    }
    int ipc_init(int rank, int nranks, const void *handles) override
    {
        HIP_TRY(hipSetDevice(device_));
        if (nranks < 1 || nranks > kIpcMaxRanks || rank < 0 || rank >= nranks) { err_ = "bad rank arguments"; return VISMA_ICP_ERR_INVALID; }
        int rc = ensure_mailbox();
        if (rc) return rc;
        for (int r = 0; r < nranks; r++) {
            if (r == rank) { peers_.box[r] = d_mbox_; continue; }
            hipIpcMemHandle_t h;
            std::memcpy(&h, (const char *)handles + (size_t)r * VISMA_ICP_IPC_HANDLE_BYTES, sizeof(h));
            void *p = nullptr;
            hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                for (int q = 0; q < r; q++)
                    if (q != rank && peers_.box[q]) { (void)hipIpcCloseMemHandle(peers_.box[q]); peers_.box[q] = nullptr; }
                err_ = std::string("hipIpcOpenMemHandle(rank ") + std::to_string(r) + "): " + hipGetErrorString(e);
                return VISMA_ICP_ERR_HIP;
            }
            peers_.box[r] = p;
        }
        ipc_rank_ = rank;
        ipc_n_ = nranks;
        // Handshake (the call is collective): one all-reduce of known values proves that every peer's
        // stores arrive in this rank's mailbox and the other way round -- a mapping that opens but does
        // not carry traffic (no peer access between two devices) must fail HERE, not in the first iteration.
        if (nranks > 1) {
            double *d_hs = nullptr;
            HIP_TRY(hipMalloc((void **)&d_hs, sizeof(double) * kNStats));
            std::vector<double> hs((size_t)kNStats);
            for (int a = 0; a < kNStats; a++) hs[(size_t)a] = (double)((rank + 1) * (a + 1));
            hipError_t e = hipMemcpyAsync(d_hs, hs.data(), sizeof(double) * kNStats, hipMemcpyHostToDevice, stream_);
            if (e == hipSuccess) e = hipMemsetAsync(d_ipc_flag_, 0, sizeof(int), stream_);
            if (e == hipSuccess)
                e = launch_ipc_allreduce(d_hs, d_hs, peers_, ipc_rank_, ipc_n_, ipc_seq_dev(), nullptr, 0, (int *)d_ipc_flag_,
                                         stream_, kIpcHandshakeSpins);
            int flag = 0;
            if (e == hipSuccess) e = hipMemcpyAsync(hs.data(), d_hs, sizeof(double) * kNStats, hipMemcpyDeviceToHost, stream_);
            if (e == hipSuccess) e = hipMemcpyAsync(&flag, d_ipc_flag_, sizeof(int), hipMemcpyDeviceToHost, stream_);
            if (e == hipSuccess) e = hipStreamSynchronize(stream_);
            (void)hipFree(d_hs);
            bool good = e == hipSuccess && flag == 0;
            const double tri = 0.5 * (double)nranks * (double)(nranks + 1);
            for (int a = 0; good && a < kNStats; a++) good = hs[(size_t)a] == tri * (double)(a + 1);
            if (!good) {
                (void)hipGetLastError();
                (void)hipMemset(d_ipc_flag_, 0, sizeof(int));
                for (int q = 0; q < nranks; q++)
                    if (q != rank && peers_.box[q]) { (void)hipIpcCloseMemHandle(peers_.box[q]); peers_.box[q] = nullptr; }
                ipc_n_ = 0;
                err_ = e != hipSuccess ? std::string("peer-to-peer handshake: ") + hipGetErrorString(e)
                     : flag ? "peer-to-peer handshake: rank " + std::to_string(flag - 1) + " did not answer"
                            : std::string("peer-to-peer handshake: wrong sum");
                return VISMA_ICP_ERR_HIP;
            }
        }
        return VISMA_ICP_OK;
    }

    void set_profiling(int level) override { profiling_ = level < 0 ? 0 : level; prof_tick_ = 0; }
    void get_timing(visma_icp_timing *t, bool reset) override
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
    void launch_config(int *tiles, int *splits) override { *tiles = plan_.src_tiles; *splits = plan_.tgt_splits; }

private:
    // Cloud-sized device buffers are recycled: a registration after another of about the same size
    // (every caller's loop) re-uses them instead of paying hipFree + hipMalloc (a device sync and ~0.5 ms
    // per 100 MB).  pool_alloc'ed pointers are returned by the ordinary free_dev.
    std::unordered_map<void *, size_t> pool_live_;
    std::vector<std::pair<void *, size_t>> pool_free_;
    int pool_alloc(void **p, size_t bytes)
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
    void pool_trim(size_t keep)
    {
        while (pool_free_.size() > keep) {
            (void)hipFree(pool_free_.front().first);
            pool_free_.erase(pool_free_.begin());
        }
    }
    void free_dev(void *&p)
    {
        if (!p) return;
        auto it = pool_live_.find(p);
        if (it == pool_live_.end()) {
            (void)hipFree(p);
        } else {
            pool_free_.push_back({p, it->second});
            pool_live_.erase(it);
            pool_trim(10);
        }
        p = nullptr;
    }
    int ensure_source(int64_t ns)
    {
        if (ns < 0) { err_ = "negative point count"; return VISMA_ICP_ERR_INVALID; }
        if (ns > 0x7fffffff - 4096) { err_ = "source too large for 32-bit indices"; return VISMA_ICP_ERR_INVALID; }
        free_dev(d_src_);
        { int prc = pool_alloc(&d_src_, sizeof(float4) * (size_t)(ns > 0 ? ns : 1)); if (prc) return prc; }
        ns_ = ns;
        have_pass_ = false;
        free_dev(d_src64_);                                      // belongs to the previous source
        return invalidate_pos();
    }
    int ensure_target(int64_t nt)
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
        have_pass_ = false;
        return VISMA_ICP_OK;
    }
    int ensure_aux(int64_t ns_pad)
    {
        if (ns_pad > aux_cap_) {
            free_dev(d_idx_); free_dev(d_d2_); free_dev(d_pos_);
            HIP_TRY(hipMalloc(&d_idx_, sizeof(int32_t) * (ns_pad > 0 ? ns_pad : 1)));
            HIP_TRY(hipMalloc(&d_d2_, sizeof(float) * (ns_pad > 0 ? ns_pad : 1)));
            HIP_TRY(hipMalloc(&d_pos_, sizeof(Pt64) * (ns_pad > 0 ? ns_pad : 1)));
            aux_cap_ = ns_pad;
            return invalidate_pos();
        }
        return VISMA_ICP_OK;
    }
    // Pick brute force or the grid for this (target, radius); build the grid if needed.
    int choose_mode(double max_dist)
    {
        if (nn_mode_ == VISMA_ICP_NN_BRUTE || nt_ == 0) { use_grid_ = false; return VISMA_ICP_OK; }
        {
            int rc = ensure_f64_views();
            if (rc) return rc;
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
    int build_grid(double max_dist)
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
        if (profiling_) { HIP_TRY(hipEventRecord(ev_[e0 + 1], stream_)); pending_.push_back({e0, 2}); }
        grid_valid_ = true;
        grid_radius_ = max_dist;
        return invalidate_pos();                                 // the slots of the old sorted order mean nothing now
    }
    int next_event_pair()
    {
        if (ev_used_ + 2 > (int)ev_.size()) {
            for (int i = 0; i < 2; i++) {
                hipEvent_t e;
                if (hipEventCreate(&e) != hipSuccess) return 0;
                ev_.push_back(e);
            }
        }
        int r = ev_used_;
        ev_used_ += 2;
        return r;
    }
    // event pairs are only read back in bulk (get_timing, or when many are pending)
    int maybe_collect_timing()
    {
        if (pending_.size() < 2048) return VISMA_ICP_OK;
        HIP_TRY(hipStreamSynchronize(stream_));
        return collect_timing();
    }
    int collect_timing()
    {
        for (const auto &p : pending_) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, ev_[p.first], ev_[p.first + 1]));
            if (p.second == 0) { timing_.nn_ms += ms; timing_.nn_launches++; }
            else if (p.second == 1) { timing_.reduce_ms += ms; timing_.reduce_launches++; }
            else { timing_.aux_ms += ms; timing_.aux_launches++; }
        }
        pending_.clear();
        ev_used_ = 0;
        return VISMA_ICP_OK;
    }

    int device_;
    hipStream_t stream_ = nullptr;
    void *d_src_ = nullptr, *d_tgt_ = nullptr, *d_nrm_ = nullptr, *d_keys_ = nullptr;
    void *d_idx_ = nullptr, *d_d2_ = nullptr, *d_partials_ = nullptr, *d_stats_ = nullptr;
    double *h_stats_ = nullptr, *h_stats_dev_ = nullptr;
    unsigned long long pub_seq_ = 0;
    bool inited_ = false;
    int64_t nt_pad_ = 0, ns_pad_ = 0, aux_cap_ = 0;
    size_t keys_bytes_ = 0;
    NNLaunch plan_{0, 0, 0};
    Xform32 T32_{};
    float r2f_ = 0.f;
    bool have_pass_ = false;
    int profiling_ = 0;        // 0 off, 1 every launch, n every n-th reduce pass
    unsigned prof_tick_ = 0;
    std::vector<hipEvent_t> ev_;
    int ev_used_ = 0;
    std::vector<std::pair<int, int>> pending_;
    visma_icp_timing timing_{};
    void *d_src64_ = nullptr, *d_tgt64_ = nullptr, *d_sorted64_ = nullptr;   // double-precision search
    void *d_nrm64_ = nullptr;
    float *pin_[4] = {nullptr, nullptr, nullptr, nullptr};   // pinned staging (see staging())
    size_t pin_cap_[4] = {0, 0, 0, 0};
    NcclComm comm_ = nullptr;
    void *d_mbox_ = nullptr, *d_ipc_flag_ = nullptr;      // peer-to-peer all-reduce: own mailbox, timeout flag
    IpcPeers peers_{};
    int ipc_rank_ = 0, ipc_n_ = 0;
    // (the exchange counter lives next to the timeout flag in device memory: d_ipc_flag_ + 8 bytes)
    unsigned long long *ipc_seq_dev() const { return reinterpret_cast<unsigned long long *>((char *)d_ipc_flag_ + 8); }
    bool tshard_ = false;                 // target-sharded rank (else: source-sharded / single)
    int64_t tgt_offset_ = 0, tgt_global_ = 0, gkeys_cap_ = 0;
    void *d_gkeys_ = nullptr, *d_claim_ = nullptr, *d_d64_ = nullptr;   // shard exchange: keys, index claims, local f64 d2
    int64_t d64_cap_ = 0;
    // target-sharded rank running the exact / f64 grid search: the buffer its f64 distances go to
    // Which exchange the sharded ranks run must not depend on what a rank happens to hold (an empty
    // shard, a degenerate grid): every rank with f64 clouds and without a forced brute-force search
    // compares in f64.
    bool shard_f64_protocol() const { return tshard_ && d_src64_ && d_tgt64_ && nn_mode_ != VISMA_ICP_NN_BRUTE; }
    double *shard_d64()
    {
        if (!shard_f64_protocol()) return nullptr;
        if (ns_ > d64_cap_) {
            free_dev(d_d64_);
            if (hipMalloc(&d_d64_, sizeof(double) * std::max<int64_t>(ns_, 1)) != hipSuccess) { (void)hipGetLastError(); d_d64_ = nullptr; d64_cap_ = 0; return nullptr; }
            d64_cap_ = ns_;
        }
        return (double *)d_d64_;
    }
    std::vector<unsigned long long> h_gkeys_;
    visma_icp_minreduce_fn minreduce_ = nullptr;
    void *minreduce_user_ = nullptr;
    // radius-cell grid (valid for one target + one radius)
    int nn_mode_ = VISMA_ICP_NN_AUTO;
    bool use_grid_ = false, grid_valid_ = false, grid_pending_ = false, brute_reduced_ = false;
    double grid_radius_ = 0.0;
    GridParams grid_{};
    void *d_box_ = nullptr, *d_sorted_ = nullptr, *d_cell_of_ = nullptr, *d_count_ = nullptr;
    void *d_start_ = nullptr, *d_bsum_ = nullptr, *d_cand_ = nullptr;
    void *d_state_ = nullptr;
    DevIcpState *h_state_ = nullptr;
    int state_cap_ = 0;
    size_t partial_rows_ = 0;
    // batch of problems with their own clouds (concatenated arrays)
    void *bt_src_ = nullptr, *bt_idx_ = nullptr, *bt_d2_ = nullptr, *bt_pos_ = nullptr, *bt_tgt_ = nullptr, *bt_sorted_ = nullptr;
    void *bt_src64_ = nullptr, *bt_tgt64_ = nullptr, *bt_sorted64_ = nullptr;
    int64_t bt_src64_cap_ = 0, bt_tgt64_cap_ = 0;
    void *bt_sorted12_ = nullptr;                          // packed copy of bt_sorted_ (exact search)
    int64_t bt_sorted12_cap_ = 0;
    void *bt_raw_ = nullptr;                               // targets as uploaded (caller's f64 values)
    size_t bt_raw_bytes_ = 0;
    void *bt_nrm_ = nullptr, *bt_nrm64_ = nullptr;         // point-to-plane batches: target normals
    int64_t bt_nrm_cap_ = 0, bt_nrm64_cap_ = 0;
    void *bt_cell_of_ = nullptr, *bt_count_ = nullptr, *bt_start_ = nullptr, *bt_bsum_ = nullptr, *bt_descs_ = nullptr;
    int64_t bt_src_cap_ = 0, bt_tgt_cap_ = 0, bt_cell_cap_ = 0, bt_out_cap_ = 0;
    int bt_bsum_cap_ = 0;
    size_t bt_desc_cap_ = 0;
    std::vector<char> bt_desc_host_;                       // (kept: the copy is asynchronous)
    int64_t view_offset_ = 0, loop_out_stride_ = 0;
    static constexpr int kGridMaxBlocks = 32768;   // (8 M queries at one per lane: the warm kernel keeps 4 waves per SIMD only there)
    double r2d_ = 0.0;
    const Pt64 *f64_src() const { return d_sorted64_ ? (const Pt64 *)d_src64_ : nullptr; }
    const Pt64 *f64_sorted() const { return d_src64_ ? (const Pt64 *)d_sorted64_ : nullptr; }
    int grid_sub_ = 1;         // row refinement the planner may use (VISMA_ICP_GRID_SUB=2: 25 half-pitch rows --
                               // 42 % fewer candidates at C4 but slower, 59 vs 51 us: more rows, 4x the table)
    int grid_blocks_env_ = 0;  // VISMA_ICP_GRID_BLOCKS override of the workgroup cap below
    int grid_blocks() const
    {
        // workgroup cap of the single-problem grid launch.  Up to 262,144 sources 1024
        // workgroups give one query per lane group (the kernel's ONE variant); beyond that
        // more workgroups keep it that way -- the fold of their partial rows costs less
        // than running the multi-round variant (1M sources: 0.12 vs 0.17 ms per iteration)
        if (grid_blocks_env_ > 0) return grid_blocks_env_;
        if (ns_ <= 262144) return 1024;
        return (int)std::min<int64_t>(kGridMaxBlocks, (ns_ + kBlock - 1) / kBlock);
    }
    // ---- warm start (grid_coop.hip): every query's winner as the candidate array holds it (fp32 point),
    // written by every exact grid search.  The array is kept CONSISTENT with the current source order and
    // target -- every entry is NaN (all bits set) or a point of the current target (reset whenever either
    // changes) -- so any pass may read it; pos_fresh_ only says that some pass has filled it since (policy:
    // the first pass of a registration runs the lane-serial kernel, which prunes progressively; the later ones
    // the warm-started kernel).
    void *d_pos_ = nullptr;
    bool pos_fresh_ = false;
    // the certificate of grid_coop.hip: the transform of the pass that left the state (host-driven passes over ONE
    // problem; device loops carry it in their DevIcpState and leave prev_T_valid_ false behind them)
    Xform64 prev_T_{};
    bool prev_T_valid_ = false;
    int cert_enabled_ = 1;       // VISMA_ICP_CERT=0: every query searched every pass (A/B timing)
    const Xform64 *cert_prev() const { return (cert_enabled_ && prev_T_valid_ && pos_fresh_) ? &prev_T_ : nullptr; }
    void note_state_pass(const Xform64 &T) { prev_T_ = T; prev_T_valid_ = true; }
    int last_kernel_ = 0;        // what the last pass ran: 0 brute force, 1 lane-serial grid, 2 warm-started cooperative grid
    int coop_enabled_ = 1;       // VISMA_ICP_COOP=0: every pass on the lane-serial kernel
    int invalidate_pos()
    {
        pos_fresh_ = false;
        prev_T_valid_ = false;
        if (d_pos_ && aux_cap_ > 0) HIP_TRY(hipMemsetAsync(d_pos_, 0xFF, sizeof(Pt64) * (size_t)aux_cap_, stream_));
        return VISMA_ICP_OK;
    }
    bool coop_ok() const
    {
        // (the kernel addresses the candidate array with 32-bit byte offsets: 12 bytes per slot)
        return coop_enabled_ && exact_ && d_src64_ && d_sorted64_ && d_sorted12_ && d_pos_ && grid_.sub == 1 &&
               (nt_ + kSortedSlack) * 12 < (1ll << 32);
    }
    // which kernel a lanes code selects (see launch_nn_grid_reduce)
    int pass_kernel(int lanes) const { return (lanes == kCoopLanes && coop_ok()) ? 2 : 1; }
    // lanes code of the next grid pass over `nprob` problems sharing the clouds
    int pass_lanes(int nprob = 1) const
    {
        if (grid_lanes_ > 0) return grid_lanes_;
        if (coop_ok() && pos_fresh_) return kCoopLanes;
        return grid_lanes(nprob);
    }
    int grid_lanes_ = 0;   // lanes cooperating on one query; 0 = by source size (VISMA_ICP_GRID_LANES overrides)
    int grid_lanes(int nprob = 1) const
    {
        if (grid_lanes_ > 0) return grid_lanes_;
        const int64_t q = ns_ * (int64_t)nprob;                  // queries of one launch
        // measured on MI355X: small clouds need the extra parallelism, large ones the locality
        // (lanes per query, loads in flight per lane), encoded G + 100*U
        if (f64_src() && exact_)   // exact search (re-measured with the branch-free insertion: tools/lanes_probe.py,
                                   // bench.py --workload c5: sweeps of ~200 k queries 801 309 k it/s, 402 295 k, 802 274 k)
            // single problems (also the source shards of 2 / 4 / 8 ranks against a 4 M-point target, tools/lanes_probe.py
            // 32768 / 65536 / 131072 x 4194304: 804 31.6 us vs 408 34.0; 802 34.0; 1201 41.5 vs 801 43.0, 802 47.6)
            return nprob > 1 ? (q <= 32768 ? 408 : (q <= 98304 ? 402 : 801))
                             : (q <= 32768 ? 804 : (q <= 98304 ? 802 : (q <= 196608 ? 1201 : 801)));
        if (f64_src())   // 32-byte candidates: fewer in flight per lane (measured 5k: 408 15.8 us vs 804 18.1)
            return q <= 32768 ? 408 : (nprob > 1 ? 402 : (q <= 131072 ? 802 : 801));
        if (nprob > 1) return q <= 32768 ? 804 : 402;          // sweeps: many queries per launch
        return ns_ <= 32768 ? 804 : (ns_ <= 98304 ? 802 : 1201);
    }
    int64_t sorted_cap_ = 0, cell_cap_ = 0;

    // ---- streamed search with exact tie-breaks + fused fold (tile.hip) --------------------
    bool exact_ = true;          // search precision "exact" (default): fp32 ranking, f64 re-rank of near-ties
    int tile_enabled_ = 0;       // VISMA_ICP_TILE=1: the LDS-streamed kernel (tile.hip, experimental)
    int fused_fold_ = 1;         // VISMA_ICP_FUSED_FOLD=0: fold the partial rows in a second launch
    int tile_config_ = -1;       // VISMA_ICP_TILE_CONFIG: LDS tile geometry (see launch_nn_tile_reduce)
    int tile_fallback_ = 0;      // VISMA_ICP_TILE_FALLBACK=1: every workgroup searches from global memory (tests)
    int tile_fused_fold_ = 1;    // VISMA_ICP_TILE_FOLD=0: fold the partial rows in a second launch (experiments)
    void *d_partials2_ = nullptr, *d_tickets_ = nullptr, *d_tstats_ = nullptr;
    size_t tickets_cap_ = 0;     // words in d_tickets_ (= rows in d_partials2_)
    Xform64 T64_last_{};         // transform of the last nn_pass, f64
    void *d_second_ = nullptr;   // brute force, exact flavour: runner-up distances [splits][ns_pad]
    size_t second_bytes_ = 0;
    int last_mode_ = 0;          // see search_is_f64()
    // (tile.hip is an experiment that lost -- 2.4-3x slower than the gather kernels, DESIGN.md 4.1b -- and is
    //  only compiled into the side build -DVISMA_WITH_TILE that tests/test_tile_kernel.py loads)
    bool use_tile() const
    {
#ifdef VISMA_WITH_TILE
        return tile_enabled_ && use_grid_ && exact_ && d_src64_ && d_sorted64_ && grid_.sub == 1 && !tshard_;
#else
        return false;
#endif
    }
    int tile_config(int64_t queries) const
    {
        if (tile_config_ >= 0) return tile_config_;
        (void)queries;
        return 0;
    }
#ifdef VISMA_WITH_TILE
    // workgroups of ONE problem with ns queries
    static int tile_blocks(int64_t ns, int config)
    {
        const int nth = tile_threads(config);
        const int64_t nb = (ns + nth - 1) / nth;
        return (int)(nb < 1 ? 1 : nb);
    }
#endif
    int ensure_tile_buffers(size_t rows, size_t ticket_words)
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
#ifdef VISMA_WITH_TILE
    // the arguments every tile launch of the resident clouds shares
    TileArgs tile_args(const Xform64 &T64, const double offset[3], bool prof) const
    {
        TileArgs a;
        std::memset(&a, 0, sizeof(a));
        a.src64 = (const Pt64 *)d_src64_;
        a.ns = (int)ns_;
        a.sorted = (const float4 *)d_sorted_;
        a.sorted64 = (const Pt64 *)d_sorted64_;
        a.start = (const unsigned *)d_start_;
        a.g = grid_;
        a.nrm = (const float4 *)d_nrm_;
        a.nrm64 = (const Pt64 *)d_nrm64_;
        a.T64 = T64;
        for (int k = 0; k < 3; k++) a.off.v[k] = offset ? offset[k] : 0.0;
        a.r2f = r2f_;
        a.idx_out = (int *)d_idx_;
        a.d2_out = (float *)d_d2_;
        a.partials = (double *)d_partials_;
        a.partials2 = (double *)d_partials2_;
        a.stats = prof ? (unsigned long long *)d_tstats_ : nullptr;
        a.nprob = 1;
        a.force_fallback = tile_fallback_;
        return a;
    }
#endif
    // workgroups per problem of launch_nn_grid_reduce (same arithmetic as the launcher)
    static int grid_launch_blocks(int64_t ns, int lanes, int max_blocks)
    {
        const int G = lanes % 100;
        int64_t want = (ns * G + kBlock - 1) / kBlock;
        int nb = (int)(want > max_blocks ? max_blocks : want);
        return nb < 1 ? 1 : nb;
    }
    // fold arguments for `nprob` problems of `bpp` workgroups each (buffers grown as needed)
    void add_ipc(FoldArgs *fa)
    {
        fa->peers = peers_;
        fa->ipc_rank = ipc_rank_;
        fa->ipc_n = ipc_n_;
        fa->ipc_seq_dev = ipc_seq_dev();
        fa->ipc_flag = (int *)d_ipc_flag_;
        fa->ipc_spins = kIpcSpinLimit;
    }
    int make_fold(int bpp, int nprob, double *stats_out, long long stats_stride, double *host_out,
                  unsigned long long seq, FoldArgs *out)
    {
        const size_t tstride = 1 + (size_t)(bpp + 31) / 32;
        int rc = ensure_tile_buffers((size_t)bpp * nprob, tstride * nprob);
        if (rc) return rc;
        out->tickets = (unsigned *)d_tickets_;
        out->partials2 = (double *)d_partials2_;
        out->ticket_stride = (int)tstride;
        out->stats_out = stats_out;
        out->stats_stride = stats_stride;
        out->host_out = host_out;
        out->seq = seq;
        return VISMA_ICP_OK;
    }
    // f64 views of clouds that were uploaded as fp32 (the exact search needs them)
    int ensure_f64_views()
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
};


}  // namespace

Engine *new_hip_engine(int device, int *rc, std::string *err)
{
    std::unique_ptr<HipEngine> e(new HipEngine(device));
    const int r = e->init();
    if (rc) *rc = r;
    if (r) { if (err) *err = e->error(); return nullptr; }
    return e.release();
}

}  // namespace drv
}  // namespace visma
