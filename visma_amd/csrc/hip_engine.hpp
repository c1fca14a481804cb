// hip_engine.hpp -- class HipEngine: the product engine of the ICP driver (gfx950 kernels on one HIP stream).
// Members, small inline methods and the declarations of the large ones, which live in
//   hip_engine_clouds.cpp   uploads and layout, the radius-cell grid, buffers and pools, timing
//   hip_engine_passes.cpp   the per-pass launches, the fused fold, the device-resident loops (single, sweeps, batches)
//   hip_engine_comm.cpp     the transports of the source- and target-sharded modes (IPC mailboxes, RCCL, exchanges)
//   hip_engine.cpp          the factory
#pragma once
#include "engine.hpp"

namespace visma {
namespace drv {

class HipEngine : public Engine {
public:
    explicit HipEngine(int device) : device_(device) {}
    ~HipEngine() override
    {
        if (!inited_) { (void)hipGetLastError(); return; }   // never touched the device
        (void)hipSetDevice(device_);
        // first of all: a persistent launch that still polls for a command reads d_sorted12_, the pending-query buffers
        // and (ranks) stores into the peers' mailboxes -- it must have ended before any of them is freed or unmapped
        if (sess_live_) (void)end_session();
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
        free_dev(d_idx_); free_dev(d_d2_); free_dev(d_pos_); free_dev(d_ru_); free_dev(d_partials_); free_dev(d_stats_);
        if (d_vox_out_) (void)hipFree(d_vox_out_);
        free_dev(d_box_); free_dev(d_sorted_); free_dev(d_cell_of_); free_dev(d_count_); free_dev(d_occ_); free_dev(d_ring_tab_);
        free_dev(d_start_); free_dev(d_bsum_); free_dev(d_cand_); free_dev(d_state_);
        free_dev(d_partials2_); free_dev(d_tickets_); free_dev(d_tstats_); free_dev(d_second_);
        free_dev(bt_src_); free_dev(bt_idx_); free_dev(bt_d2_); free_dev(bt_pos_); free_dev(bt_tgt_); free_dev(bt_sorted_);
        free_dev(bt_nrm_); free_dev(bt_nrm64_); free_dev(bt_raw_);
        free_dev(bt_src64_); free_dev(bt_tgt64_); free_dev(bt_sorted64_);
        free_dev(bt_cell_of_); free_dev(bt_count_); free_dev(bt_start_); free_dev(bt_bsum_); free_dev(bt_descs_);
        if (h_state_) (void)hipHostFree(h_state_);
        if (h_stats_) (void)hipHostFree(h_stats_);
        if (h_cmd_ && !cmd_direct_) (void)hipHostFree(h_cmd_);
        if (d_cmd_block_) (void)hipFree(d_cmd_block_);
        if (h_flag_) (void)hipHostFree(h_flag_);
        free_dev(d_relay_); free_dev(d_timeline_); free_dev(d_peer_table_); free_dev(d_fold_tag_);
        if (d_sweep_relay_) (void)hipFree(d_sweep_relay_);
        pool_trim(0);
        if (d_raw_src_) (void)hipFree(d_raw_src_);
        if (stream_src_) (void)hipStreamDestroy(stream_src_);
        if (stream_) (void)hipStreamDestroy(stream_);
    }

    int init();

    int set_target_f64(const double *xyz, int64_t nt, int stride, double *c, bool compute_centre, bool want64) override;
    bool last_upload_f32_ = false;   // (reported by VISMA_ICP_UPLOAD_TRACE)
    bool host_box_valid_ = false;    // the fp32 target's bounding box is known from the staging pass
    float host_mn_[3] = {0, 0, 0}, host_mx_[3] = {0, 0, 0};

    int set_target_voxel_f64(const double *xyz, int64_t n, int stride, double voxel, double *c, bool compute_centre,
                             bool want64, int64_t *nt_out) override;
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
    int prepare_search(int64_t ns, bool want64, double max_dist) override;
    int64_t prepared_ns_ = -1;
    bool prepared_want64_ = false;
    int begin_raw_source(int64_t ns, bool want64, std::vector<int32_t> &order);
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
    int finish_raw_source(int64_t ns, const double *c, std::vector<int32_t> &order, const void *raw = nullptr, hipStream_t st = nullptr);
    int set_source_f64(const double *xyz, int64_t ns, int stride, const double *c, bool want64,
                       std::vector<int32_t> &order) override;
    int set_source_meshes_f64(const MeshSource *meshes, int n_meshes, int quirks, unsigned long long seed, const double *c,
                              bool want64, std::vector<int32_t> &order, int64_t *ns_out) override;
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
    int ensure_second(int64_t ns_pad, int splits);
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
    int set_target_normals64(const Pt64 *n) override;

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

    int nn_pass(const Mat4 &Tc, double max_dist) override;

    int reduce(const Mat4 &Tc, bool plane, const double offset[3], double *stats) override;

    int get_correspondences(int32_t *idx, float *d2) override;

    // ---- the persistent launch of a host loop (kernels.h: PersistArgs; grid_coop.hip: nn_coop_kernel_persist)
    void set_persistent(int enabled, double timeout_ms) override
    {
        persist_enabled_ = enabled != 0;
        persist_cooldown_ = 0;
        if (timeout_ms >= 0.5 && timeout_ms <= 5000.0) persist_timeout_ms_ = timeout_ms;
    }
    void set_ring_search(int mode) override
    {
        const int m = mode < 0 ? -1 : (mode > 0 ? 1 : 0);
        if (m == ring_mode_) return;
        if (sess_live_) (void)end_session();                 // (a launch that is alive searches the table that is about to go)
        ring_mode_ = m;
        grid_valid_ = false;                                 // the next pass plans its grid again
    }
    void get_ring_search(int *rings, double *cell, double *occupancy) const override
    {
        if (rings) *rings = grid_valid_ ? grid_.ring : 0;
        if (cell) *cell = grid_valid_ ? (double)grid_.h : 0.0;
        if (occupancy) *occupancy = grid_valid_ ? grid_occupancy_ : 0.0;
    }
    void get_sweep_info(double *launches, double *aborts) const override
    {
        if (launches) *launches = sweep_launches_total_;
        if (aborts) *aborts = sweep_aborts_total_;
    }
    void get_persistent_info(visma_icp_persistent_info *out) const override
    {
        out->enabled = (persist_enabled_ && persist_cooldown_ == 0) ? 1 : 0;
        out->last_loop_persistent = last_loop_persist_passes_ > 0 ? 1 : 0;
        out->last_loop_passes = last_loop_persist_passes_;
        out->launches = timing_persist_launches_total_;
        out->passes = timing_persist_passes_total_;
        out->aborts = persist_aborts_total_;
        out->timeout_ms = persist_timeout_ms_;
        out->cu_share = persist_cu_share();
        out->device_slots = persist_slots_seen_;
    }
    void stall_command(int nth, double ms) override { stall_nth_ = nth; stall_ms_ = ms; }
    void loop_begin(int max_passes) override
    {
        loop_scope_ = true;
        loop_budget_ = max_passes;
        loop_persist_passes_ = 0;
        if (persist_cooldown_ > 0) persist_cooldown_--;
        if (ring_lanes_auto_) ring_lanes_ = 8;               // (a new loop knows nothing about how far its start is from the end)
    }
    bool loop_across_ranks_ok() const override { return persist_ranks_ok(); }
    int loop_end() override
    {
        loop_scope_ = false;
        loop_budget_ = 0;
        const int rc = sess_live_ ? end_session() : VISMA_ICP_OK;
        last_loop_persist_passes_ = loop_persist_passes_;
        return rc;
    }

    bool supports_device_loop() const override { return true; }

    void select_problem(int b) override { view_offset_ = (int64_t)b * loop_out_stride_; }

    int run_loop(const LoopParams &lp, const Mat4 *Tc0s, int nprob, LoopResult *out) override;

    int run_loop_batch(const LoopParams &lp, const std::vector<BatchProblem> &pb, LoopResult *out) override;

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
    // (a batch runs the grid search whatever the context's own clouds last used)
    int search_kernel_used() const override { return (use_grid_ || last_was_batch_) ? last_kernel_ : 0; }
    bool last_was_batch_ = false;
    int forget_winners() override
    {
        HIP_TRY(hipSetDevice(device_));
        if (sess_live_) { int rc = end_session(); if (rc) return rc; }
        return invalidate_pos();
    }

    int set_target_shard(int64_t offset, int64_t global_nt) override;
    void set_minreduce(visma_icp_minreduce_fn fn, void *user) override { minreduce_ = fn; minreduce_user_ = user; }

    int shard_exchange(const Xform64 &T64, bool plane, const double offset[3], double *pub, unsigned long long seq);

    // Target shards in the device loop: the f64 protocol of shard_exchange with RCCL's stream-ordered
    // all-reduces; the kernels read transform / frame / radius from the state.
    bool shard_loop_on_device() const override { return tshard_ && comm_ != nullptr && shard_f64_protocol(); }
    int shard_exchange_on_stream(const DevIcpState *st, int plane, int *nblocks);

    int comm_init(int rank, int nranks, const void *id) override;
    bool has_device_allreduce() const override { return comm_ != nullptr || ipc_n_ > 1; }
    int bind_device() override { HIP_TRY(hipSetDevice(device_)); return VISMA_ICP_OK; }
    hipStream_t aux_stream() override { return stream_; }

    int ensure_mailbox();
    int ipc_export(void *out) override;
    int ipc_init(int rank, int nranks, const void *handles) override;

    void set_profiling(int level) override { profiling_ = level < 0 ? 0 : level; prof_tick_ = 0; }
    void get_timing(visma_icp_timing *t, bool reset) override;
    void launch_config(int *tiles, int *splits) override { *tiles = plan_.src_tiles; *splits = plan_.tgt_splits; }

private:
    // Cloud-sized device buffers are recycled: a registration after another of about the same size
    // (every caller's loop) re-uses them instead of paying hipFree + hipMalloc (a device sync and ~0.5 ms
    // per 100 MB).  pool_alloc'ed pointers are returned by the ordinary free_dev.
    std::unordered_map<void *, size_t> pool_live_;
    std::vector<std::pair<void *, size_t>> pool_free_;
    int pool_alloc(void **p, size_t bytes);
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
    int ensure_target(int64_t nt);
    int ensure_aux(int64_t ns_pad);
    int choose_mode(double max_dist);
    int build_grid(double max_dist);
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
    int collect_timing();

    int device_;
    hipStream_t stream_ = nullptr;
    // (round 6) the raw SOURCE upload -- copy, Morton order on the device, the permutation back -- runs on a stream of its
    // own, from a buffer of its own: it needs the target's centroid (known when the host has staged the last piece) and
    // nothing else of the target, so it overlaps the tail of the target's copies and the grid build that prepare_search
    // queued behind them (set_target_f64 no longer drains the stream on its way out).  VISMA_ICP_UPLOAD_OVERLAP=0: as before.
    hipStream_t stream_src_ = nullptr;
    void *d_raw_src_ = nullptr;
    size_t raw_src_bytes_ = 0;
    int upload_overlap_ = 1;
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
        // (the ring search works eight lanes per query and hides its dependent trips behind other waves: one query per
        //  octet up to a million queries)
        if (grid_.ring > 0) return (int)std::min<int64_t>(kGridMaxBlocks, std::max<int64_t>(1024, (ns_ * (ring_lanes() - 200) + kBlock - 1) / kBlock));
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
    // the runner-up half of that state (round 5b, grid_coop.hip: kCoopRu): per query the runner-up's f64 point, its
    // index | LB3 << 32; all bits set = none; same size, same resets as d_pos_.  The kernels are built WITHOUT that code
    // by default (-DVISMA_COOP_RU=1 builds it: measured slower), so the buffer exists only when asked for:
    // VISMA_ICP_RUNNER_UP=1 (read when the context is created) with such a build
    void *d_ru_ = nullptr;
    int runner_up_ = 0;
    Pt64 *ru_state() const { return runner_up_ ? (Pt64 *)d_ru_ : nullptr; }
    // the certificate of grid_coop.hip: the transform of the pass that left the state (host-driven passes over ONE
    // problem; device loops carry it in their DevIcpState and leave prev_T_valid_ false behind them)
    Xform64 prev_T_{};
    bool prev_T_valid_ = false;
    int cert_enabled_ = 1;       // VISMA_ICP_CERT=0: every query searched every pass (A/B timing)
    const Xform64 *cert_prev() const { return (cert_enabled_ && prev_T_valid_ && pos_fresh_) ? &prev_T_ : nullptr; }
    void note_state_pass(const Xform64 &T) { prev_T_ = T; prev_T_valid_ = true; }
    int last_kernel_ = 0;        // what the last pass ran: 0 brute force, 1 lane-serial grid, 2 warm-started cooperative grid,
                                 // 3 ring search over cells smaller than the radius (grid_ring.hip)
    int coop_enabled_ = 1;       // VISMA_ICP_COOP=0: every pass on the lane-serial kernel
    int invalidate_pos()
    {
        pos_fresh_ = false;
        prev_T_valid_ = false;
        if (d_pos_ && aux_cap_ > 0) HIP_TRY(hipMemsetAsync(d_pos_, 0xFF, sizeof(Pt64) * (size_t)aux_cap_, stream_));
        if (d_ru_ && aux_cap_ > 0) HIP_TRY(hipMemsetAsync(d_ru_, 0xFF, sizeof(Pt64) * (size_t)aux_cap_, stream_));
        return VISMA_ICP_OK;
    }
    bool coop_ok() const
    {
        // (the kernel addresses the candidate array with 32-bit byte offsets: 12 bytes per slot)
        return coop_enabled_ && exact_ && d_src64_ && d_sorted64_ && d_sorted12_ && d_pos_ && grid_.sub == 1 && grid_.ring == 0 &&
               (nt_ + kSortedSlack) * 12 < (1ll << 32);
    }
    // which kernel a lanes code selects (see launch_nn_grid_reduce)
    int pass_kernel(int lanes) const { return grid_.ring > 0 ? 3 : ((lanes == kCoopLanes && coop_ok()) ? 2 : 1); }
    // lanes code of the next grid pass over `nprob` problems sharing the clouds
    int pass_lanes(int nprob = 1) const
    {
        if (grid_.ring > 0) return ring_lanes();             // cells smaller than the radius: the ring search, whatever is forced
        if (grid_lanes_ > 0) return grid_lanes_;
        if (coop_ok() && pos_fresh_) return kCoopLanes;
        // the FIRST pass of a large registration inside the persistent launch of its host loop (round 5, OPT-IN:
        // VISMA_ICP_COLD_IN_LAUNCH=1): the certificate kernel started cold (no winner known: every query searches against
        // the radius) saves a launch, a drain and a host round trip, but its pass costs 82 us where the lane-serial
        // kernel's costs 62 + the launch: measured on one box, us per iteration over 1..20 from the identity,
        // 262,144 -> 4 M 32.9-33.3 inside vs 33.0 separate, 131,072 28.0-28.2 vs 27.5, partial overlap (half of the
        // queries without a partner: every one of their 27 cells listed) 41.1-41.6 vs 39.3 -- no gain, so not the default
        if (nprob == 1 && cold_in_launch_ && ns_ >= cold_in_launch_min_ns_ && loop_scope_ && loop_budget_ >= 2 && coop_ok() &&
            persist_static_ok())
            return kCoopLanes;
        return grid_lanes(nprob);
    }
    int cold_in_launch_ = 0;                 // VISMA_ICP_COLD_IN_LAUNCH=1: the first pass of large registrations inside the launch
    int64_t cold_in_launch_min_ns_ = 131072; // VISMA_ICP_COLD_IN_LAUNCH_MIN_NS (below: lane-serial 20 us vs 29 at 5 k)
    int grid_lanes_ = 0;   // lanes cooperating on one query; 0 = by source size (VISMA_ICP_GRID_LANES overrides)
    // ---- radii that are large against the point spacing (grid_ring.hip): build_grid looks at the occupancy of the
    // radius-sized table and, above ring_occ_min_ points per occupied cell, builds a finer one (ring_occ_target_ points per
    // occupied cell, surfaces assumed: occupancy ~ edge^2) that the ring kernel searches.  VISMA_ICP_RING = 0 never,
    // 1 whenever the f64 views exist and a finer table fits, unset: by occupancy.
    // lanes per query of a ring pass.  Many rows to visit (first passes: the nearest neighbour is several cells away) want
    // eight lanes per query, a converged registration (a handful of rows around the previous winner) four or two: the host
    // loop picks from the last pass's rmse in cells (measured at C4's sizes, literal motion: pass 20 119 / 99 / 88 us with
    // 8 / 4 / 2 lanes, pass 1 369 / 502 / 787).  VISMA_ICP_RING_LANES = 1, 2, 4, 8 fixes it.  Device-resident loops, whose
    // statistics never reach the host between passes, keep eight.
    int ring_lanes_ = 8;
    bool ring_lanes_auto_ = true;
    double ring_lanes_t84_ = 1.6, ring_lanes_t42_ = 0.8;    // rmse / cell edge below which 4 / 2 lanes take over (VISMA_ICP_RING_T84 / _T42:
                                                            // where the per-pass times of the fixed settings cross, profiles/r06_ring_search_probe.txt)
    int ring_lanes() const { return 200 + ((ring_lanes_auto_ && !pos_fresh_) ? 8 : ring_lanes_); }   // (a lanes code: G + 100 U, U unused)
    void note_ring_stats(const double *stats)
    {
        if (!ring_lanes_auto_ || grid_.ring <= 0) return;
        ring_lanes_ = 8;
        if (stats[0] > 0.0 && stats[1] >= 0.0 && loop_scope_) {
            const double cells = std::sqrt(stats[1] / stats[0]) / (double)grid_.h;
            ring_lanes_ = cells < ring_lanes_t42_ ? 2 : (cells < ring_lanes_t84_ ? 4 : 8);
            // (fewer lanes per query are fewer waves: only where they still fill the device -- 8,192 waves)
            while (ring_lanes_ < 8 && ns_ * ring_lanes_ < 8192 * 64) ring_lanes_ *= 2;
        }
    }
    int ring_mode_ = -1;
    // (48: measured, tools/ring_policy_probe.py -- single registrations gain from ~20 points per occupied cell on (1.5x at 21,
    //  3.8x at 52, 7.6x at 84), yaw sweeps, whose far-off starts leave most queries without a partner -- the ring walk's
    //  worst case: every row of the radius ball -- lose up to 3x below ~50 and gain 1.8-2x from there)
    double ring_occ_min_ = 48.0, ring_occ_target_ = 8.0;
    // ... and ONE registration at a time (host loop, device loop, the per-pass API) from 20 on: between the two thresholds the
    // table depends on who asks.  caller_sweep_: the pass about to run belongs to a sweep over several starts;
    // ring_for_sweep_: what the current table was planned for.  A context whose callers alternate re-plans at most three
    // times per target (1 ms each at 4 M points): the third time for good, to radius-sized cells.
    double ring_occ_min_single_ = 20.0;
    bool caller_sweep_ = false, ring_for_sweep_ = false;
    int ring_flips_ = 0;
    double ring_threshold() const { return (caller_sweep_ || ring_flips_ >= 3) ? ring_occ_min_ : std::min(ring_occ_min_single_, ring_occ_min_); }
    bool ring_replan_needed() const
    {
        if (ring_mode_ != -1 || !(grid_occupancy_ > 0.0) || ring_flips_ >= 3) return false;   // fixed / never counted / settled
        if (!(grid_occupancy_ >= ring_occ_min_single_ && grid_occupancy_ < ring_occ_min_)) return false;   // both callers agree
        return ring_for_sweep_ != caller_sweep_;
    }
    void *d_occ_ = nullptr, *d_ring_tab_ = nullptr;
    int ring_tab_rings_ = 0;
    RingTable ring_tab_{nullptr, 0};         // what a ring pass gets beside grid_ (launch_nn_grid_reduce: `ring`)
    static int nrows_of_rings(int rings) { return (2 * rings + 1) * (2 * rings + 1); }
    const RingTable *ring_table() const { return grid_.ring > 0 ? &ring_tab_ : nullptr; }
    double grid_occupancy_ = 0.0;            // points per occupied cell of the radius-sized table (0 = not measured)
    int grid_lanes(int nprob = 1) const
    {
        if (grid_.ring > 0) return ring_lanes();
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
        return tile_enabled_ && use_grid_ && exact_ && d_src64_ && d_sorted64_ && grid_.sub == 1 && grid_.ring == 0 && !tshard_;
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
    int ensure_tile_buffers(size_t rows, size_t ticket_words);
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
        fa->peer_table = (void *const *)d_peer_table_;
        fa->ipc_rank = ipc_rank_;
        fa->ipc_n = ipc_n_;
        fa->ipc_seq_dev = ipc_seq_dev();
        fa->ipc_flag = (int *)d_ipc_flag_;
        fa->ipc_spins = kIpcSpinLimit;
    }
    int make_fold(int bpp, int nprob, double *stats_out, long long stats_stride, double *host_out,
                  unsigned long long seq, FoldArgs *out);
    // device-resident loops: the problem's state advances in the fold epilogue of the search launch (FoldArgs::solve) --
    // the closed-form update on one GPU; Gauss-Newton / point-to-plane loops and ranks keep solve_state_kernel
    int solve_in_fold_ = 0;          // VISMA_ICP_SOLVE_IN_FOLD=1: see DESIGN.md 4.4 -- measured SLOWER than the launch of its own (A/B knob)
    // (`lanes`: the launch's lanes code -- the certificate kernels of grid_coop.hip do not carry the epilogue)
    bool solve_in_fold(const LoopParams &lp, int lanes = 0, int nprob = 2) const
    {
        static const bool forced = std::getenv("VISMA_ICP_COOP_KERNEL") != nullptr;
        if (!kSolveInFold || forced || (lanes == kCoopLanes && nprob == 1)) return false;
        return solve_in_fold_ && fused_fold_ && !lp.plane && lp.solver == VISMA_ICP_SOLVER_KABSCH && !tshard_ && !comm_ && ipc_n_ <= 1;
    }
    int ensure_f64_views();

    // ---- persistent sessions: ONE launch of the certificate kernel runs the remaining passes of the host loop that
    // announced itself (loop_begin): reduce() starts it, later reduce() calls only post the next transform to the
    // command block in mapped host memory and wait for the statistics as always; loop_end() / anything that does not
    // fit (another radius, plane, frame) posts STOP and waits for the launch to end.  The launch gives up by itself
    // when no command arrives in time (the host then finds the stream idle and goes on with ordinary launches).
    int persist_enabled_ = 1;        // VISMA_ICP_PERSIST=0: one launch per pass
    // After a launch that gave up (its workgroups could not all become resident -- another stream held compute units --, or
    // the host came back too late) the context's next loops launch once per pass; persistent launches are tried again after
    // kPersistCooldownLoops loops (until round 5: never again unless visma_icp_set_persistent asked), or at once when asked.
    static constexpr int kPersistCooldownLoops = 8;
    int persist_cooldown_ = 0;
    int persist_prio_ = 0;           // VISMA_ICP_PERSIST_PRIO: 0 none, 1 / 2 the wave-priority experiments of grid_coop.hip
    double persist_start_ms_ = 5.0;  // VISMA_ICP_PERSIST_START_MS: how long a launch waits for all of its workgroups to begin
    // what visma_icp_get_persistent_info reports (never reset)
    int loop_persist_passes_ = 0, last_loop_persist_passes_ = 0;
    double timing_persist_launches_total_ = 0.0, timing_persist_passes_total_ = 0.0, persist_aborts_total_ = 0.0;
    mutable int persist_slots_seen_ = 0;
    // the process-wide cap on the part of a device's workgroup slots a persistent launch may hold
    // (visma_icp_set_persistent_cu_share; VISMA_ICP_PERSIST_CU_SHARE)
    static double persist_cu_share() { return persist_cu_share_ref().load(); }
    double persist_timeout_ms_ = 200.0;   // VISMA_ICP_PERSIST_TIMEOUT_MS: the poller's patience
    bool loop_scope_ = false;
    int loop_budget_ = 0;            // passes the announced loop may still run
    bool sess_live_ = false;         // a persistent launch is in flight
    // (round 6) ... queued behind the registration's COLD pass while the host still waited for that pass's statistics: its
    // first pass too is posted as a command (PersistArgs::wait_first).  Dispatch, ramp and the host's turn-around overlap:
    // the 20-30 us between the end of the cold pass and the begin of the launch shrink to the turn-around.
    // VISMA_ICP_PERSIST_EARLY=0: the launch starts with the first warm pass, as in rounds 4-5.
    bool sess_early_ = false;
    int persist_early_ = 1;
    // (round 6) the persistent SWEEP launch of device-resident loops over shared clouds (grid_wave.hip: nn_wave_kernel_sweep):
    // after the first (cold) pass, ONE launch runs the remaining passes of every problem -- search, fold, closed-form update,
    // stop test inside it.  Closed-form point-to-point loops on one GPU whose workgroups all fit the device at once;
    // VISMA_ICP_SWEEP_PERSIST=0: one search launch + one solve launch per pass, as before.
    int sweep_persist_ = 1;
    void *d_sweep_relay_ = nullptr;      // 32 words per problem + the "dead" word behind them
    int sweep_relay_cap_ = 0;
    unsigned sweep_tag_ = 0;
    double sweep_launches_total_ = 0.0, sweep_aborts_total_ = 0.0;
    int sess_pass_ = 0, sess_max_ = 0;
    unsigned sess_tag0_ = 0, cmd_tag_ = 0;
    unsigned long long sess_seq0_ = 0;
    bool sess_plane_ = false, sess_prof_ = false;
    double sess_off_[3] = {0, 0, 0};
    float sess_r2f_ = 0.f;
    int sess_dev_slot_ = -1;         // this session holds the device's one persistent slot
    int sess_e0_ = -1;               // its event pair (profiling)
    std::vector<std::pair<int, int>> sess_pending_;   // (event pair, passes run) of finished sessions
    unsigned long long *h_cmd_ = nullptr, *h_cmd_dev_ = nullptr;   // kPersistWords command words; word 32: the kernel's flag
    void *d_relay_ = nullptr;
    void *d_fold_tag_ = nullptr;     // the polled fold's granule rows (2,048 rows + 256 group rows of 32 granules)
    void *d_peer_table_ = nullptr;   // peers_ in device memory (the persistent kernel's fold reads the mailboxes from there)
    bool peers_share_device_ = true; // some peer of the IPC ring sits on THIS device (tests): no persistent launch then --
                                     // the ranks' launches would each hold a part of the compute units and wait for the rest
    // Source-sharded ranks keep their launches alive across passes only when ASKED (VISMA_ICP_PERSIST_RANKS=1, read when
    // the context is created): until one run on two real devices has passed, the default for ranks is the path two and
    // three processes on one GPU have exercised -- one launch per pass, the exchange through the mailboxes inside it.
    int persist_ranks_ = 0;
    bool persist_ranks_ok() const
    {
        static const bool shared_ok = std::getenv("VISMA_ICP_PERSIST_SHARED_GPU") != nullptr;   // (tests with small clouds)
        return persist_ranks_ && ipc_n_ > 1 && d_peer_table_ && !comm_ && (!peers_share_device_ || shared_ok);
    }
    bool cmd_direct_ = false;        // h_cmd_ is fine-grained DEVICE memory, stored to through the PCIe BAR (large-BAR systems)
    void *d_cmd_block_ = nullptr;
    unsigned *h_flag_ = nullptr, *h_flag_dev_ = nullptr;   // the launch's "gave up" word, always in mapped host memory
    std::chrono::steady_clock::time_point t_stats_seen_{}, t_posted_{};   // when the host saw a pass's statistics / posted a command
    static bool trace_persist() { static const bool t = std::getenv("VISMA_ICP_PERSIST_TRACE") != nullptr; return t; }
    double host_gap_us_ = 0.0, wait_us_ = 0.0;
    int host_gaps_ = 0;
    void *d_timeline_ = nullptr;     // VISMA_ICP_PERSIST_TIMELINE=<file>: per pass and workgroup clocks of every session, appended
    int timeline_passes_ = 0, timeline_blocks_ = 0;
    std::string timeline_path_;
    int stall_nth_ = 0;              // (tests) sleep stall_ms_ before the stall_nth_-th command from now
    double stall_ms_ = 0.0;
    bool persist_possible(int lanes, int nblocks, bool fused, bool plane) const;
    // what a persistent launch needs of the context, whatever the pass (persist_possible adds the launch's shape)
    bool persist_static_ok() const
    {
        static const bool forced = std::getenv("VISMA_ICP_COOP_KERNEL") != nullptr;       // (A/B runs of the two one-pass kernels)
        return persist_enabled_ && persist_cooldown_ == 0 && fused_fold_ && !tshard_ && !comm_ && !minreduce_ && (ipc_n_ <= 1 || persist_ranks_ok()) &&
               !(grid_lanes_ > 0 && grid_lanes_ != kCoopLanes) && !forced && ns_ <= (int64_t)grid_blocks() * kBlock;
    }
    int start_session(const Xform64 &T64, bool plane, const double offset[3], unsigned long long seq, int nblocks, bool prof);
    void post_command(const Xform64 &T64, unsigned cmd);
    int end_session();
    void finish_session();
    void fill_persist_args(PersistArgs *pa);
    int launch_grid_pass(const Xform64 &T64, bool plane, const double offset[3], unsigned long long seq, bool prof,
                         const PersistArgs *persist, int *nblocks_out, bool *ipc_done, bool early = false);
};

}  // namespace drv
}  // namespace visma
