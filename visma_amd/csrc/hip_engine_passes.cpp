// hip_engine_passes.cpp -- HipEngine: the per-pass launches (brute force / lane-serial grid / warm-started grid), the fused fold, the device-resident loops (single problem, sweeps, batches of problems with their own clouds).
#include "hip_engine.hpp"

#if defined(__x86_64__) || defined(_M_X64)
#include <immintrin.h>
#define VISMA_STORE_FENCE() _mm_sfence()
#else
// (no write-combining BAR stores elsewhere: the command block then lies in host memory, hip_engine_clouds.cpp)
#define VISMA_STORE_FENCE() std::atomic_thread_fence(std::memory_order_seq_cst)
#endif

namespace visma {
namespace drv {

int HipEngine::nn_pass(const Mat4 &Tc, double max_dist)
{
    HIP_TRY(hipSetDevice(device_));
    if (!d_src_ || !d_tgt_) { err_ = "clouds not set"; return VISMA_ICP_ERR_STATE; }
    last_was_batch_ = false;
    if (sess_live_ && (float)(max_dist * max_dist) != sess_r2f_) {
        // (another radius: the grid may be rebuilt below -- not behind a launch that waits for this thread)
        int rc = end_session();
        if (rc) return rc;
    }
    const int64_t ns_min_pad = ((ns_ + kBlock - 1) / kBlock) * kBlock;
    for (int i = 0; i < 12; i++) { T32_.m[i] = (float)Tc.m[i]; T64_last_.m[i] = Tc.m[i]; }
    r2f_ = (float)(max_dist * max_dist);
    r2d_ = (double)r2f_;                                     // (double)(float)(r*r): KDTreeFlann.cpp:184-185
    caller_sweep_ = false;
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

int HipEngine::reduce(const Mat4 &Tc, bool plane, const double offset[3], double *stats)
{
    HIP_TRY(hipSetDevice(device_));
    if (!have_pass_) { err_ = "reduce before nn_pass"; return VISMA_ICP_ERR_STATE; }
    if (plane && !d_nrm_) { err_ = "point-to-plane needs target normals"; return VISMA_ICP_ERR_STATE; }
    Xform64 T64;
    for (int i = 0; i < 12; i++) T64.m[i] = Tc.m[i];
    int e0 = -1;
    if (sess_live_ && !use_grid_) { int src = end_session(); if (src) return src; }   // (the search changed under a live launch)
    // profiling level n > 1: time (and count candidates on) every n-th pass only --
    // four event records per iteration cost ~14 us of the ~75 they measure
    const bool prof = profiling_ > 0 && (++prof_tick_ % profiling_) == 0;
    // without RCCL the fold kernel publishes to mapped host memory itself
    const unsigned long long seq = ++pub_seq_;
    const bool ipc = ipc_n_ > 1;
    bool ipc_done = false;                           // the exchange ran inside the search launch
    bool in_session = false;                         // this pass runs inside a persistent launch (see hip_engine.hpp)
    bool posted = false;                             // ... as a command to a launch that was already running
    double *pub = (comm_ || ipc) ? nullptr : h_stats_dev_;
#ifdef VISMA_WITH_TILE
    if (use_tile()) {
        // ONE launch: streamed search + exact re-rank + moments + fused fold + publication
        const int cfg = tile_config(ns_);
        const int nblocks = tile_blocks(ns_, cfg);
        const size_t tstride = 1 + (size_t)(nblocks + kFoldGroup - 1) / kFoldGroup;
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
        const bool fused = fused_fold_ && !tshard_;
        const int lanes = pass_lanes();
        if (sess_live_) {
            // the persistent launch is waiting for exactly this: the next transform
            const double *o = offset;
            const bool same = plane == sess_plane_ && r2f_ == sess_r2f_ && o[0] == sess_off_[0] && o[1] == sess_off_[1] &&
                              o[2] == sess_off_[2] && sess_pass_ < sess_max_ && seq == sess_seq0_ + (unsigned long long)sess_pass_ &&
                              lanes == kCoopLanes;
            bool patient = true;
            if (same) {
                if (stall_nth_ > 0 && --stall_nth_ == 0)   // (tests: a host thread that does not come back in time)
                    std::this_thread::sleep_for(std::chrono::duration<double, std::milli>(stall_ms_));
                // The host keeps the patience: whoever comes back this late posts STOP, not the transform -- the launch
                // waits four times as long before it gives up by itself, so a GO never meets a half-empty launch.
                const double gap_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_stats_seen_).count();
                patient = gap_us <= persist_timeout_ms_ * 1e3;
                if (!patient) {
                    timing_.persist_aborts += 1.0; persist_aborts_total_ += 1.0;
                    persist_cooldown_ = kPersistCooldownLoops;   // (for the next loops, or until visma_icp_set_persistent asks again)
                } else if (trace_persist()) {
                    host_gap_us_ += gap_us;
                    host_gaps_++;
                }
            }
            if (same && patient) {
                post_command(T64, kPersistGo);
                if (trace_persist()) t_posted_ = std::chrono::steady_clock::now();
                sess_pass_++;
                loop_persist_passes_++;
                timing_persist_passes_total_ += 1.0;
                in_session = true;
                posted = true;                               // (its transform becomes "the previous one" once the pass has run)
                ipc_done = true;                             // (ranks: the folding workgroup exchanges inside the launch)
                last_kernel_ = 2;
            } else {
                int rc = end_session();
                if (rc) return rc;
            }
        }
        if (!in_session) {
            const int nb = grid_launch_blocks(ns_, lanes, grid_blocks());
            PersistArgs pa{};
            const PersistArgs *pp = nullptr;
            if (loop_scope_ && loop_budget_ >= 2 && persist_possible(lanes, nb, fused, plane)) {
                int rc = start_session(T64, plane, offset, seq, nb, profiling_ > 0);
                if (rc == VISMA_ICP_OK && sess_live_) {
                    fill_persist_args(&pa);
                    HIP_TRY(hipMemsetAsync((unsigned long long *)d_relay_ + kPersistDead, 0, 2 * sizeof(unsigned long long), stream_));   // (dead, started)
                    if (!timeline_path_.empty()) {
                        // (measurement: clocks of up to 64 passes of this launch, read back when it has ended)
                        if (!d_timeline_) HIP_TRY(hipMalloc(&d_timeline_, sizeof(unsigned long long) * 2 * 64 * 1024));
                        if (nb <= 1024) {
                            HIP_TRY(hipMemsetAsync(d_timeline_, 0, sizeof(unsigned long long) * 2 * 64 * 1024, stream_));
                            pa.timeline = (unsigned long long *)d_timeline_;
                            pa.timeline_passes = std::min(64, sess_max_);
                            timeline_passes_ = pa.timeline_passes;
                            timeline_blocks_ = nb;
                        }
                    }
                    pp = &pa;
                }
            }
            // (a cold pass that a persistent launch will follow: the launch's "dead / started" words are cleared BEFORE the
            //  cold pass, not between the two)
            const bool early_hope = !pp && persist_early_ && loop_scope_ && loop_budget_ >= 3 && lanes != kCoopLanes && !ipc &&
                                    !comm_ && !tshard_ && timeline_path_.empty() && d_relay_;
            if (early_hope)
                HIP_TRY(hipMemsetAsync((unsigned long long *)d_relay_ + kPersistDead, 0, 2 * sizeof(unsigned long long), stream_));
            int rc = launch_grid_pass(T64, plane, offset, seq, pp ? sess_prof_ : prof, pp, &nblocks, &ipc_done);
            if (rc) { if (pp) finish_session(); return rc; }
            if (pp) {
                sess_pass_ = 1; in_session = true;
                loop_persist_passes_++;
                timing_persist_launches_total_ += 1.0;
                timing_persist_passes_total_ += 1.0;
            } else if (early_hope && pass_lanes() == kCoopLanes) {
                // ---- the EARLY persistent launch (hip_engine.hpp: sess_early_): this was the registration's cold pass, the
                // passes after it run the certificate kernel -- queue their launch now, behind the cold pass, and let it wait
                // for its first transform like for every later one.  (pass_lanes() speaks about the NEXT pass now: the cold
                // launch left the per-query state behind.)
                const int nb2 = grid_launch_blocks(ns_, kCoopLanes, grid_blocks());
                if (persist_possible(kCoopLanes, nb2, fused, plane)) {
                    const int budget = loop_budget_;
                    loop_budget_ = budget - 1;                       // (what the launch may run: the passes after this one)
                    int src = start_session(T64, plane, offset, seq + 1ull, nb2, profiling_ > 0);
                    loop_budget_ = budget;
                    if (src == VISMA_ICP_OK && sess_live_) {
                        PersistArgs pe{};
                        fill_persist_args(&pe);
                        pe.wait_first = 1;
                        int nb_unused = 1;
                        bool ipc_unused = false;
                        // (T64: the COLD pass's transform -- the launch takes it as "the transform before" its first pass,
                        //  what that pass's certificates measure their motion from; its own arrives with the first command)
                        src = launch_grid_pass(T64, plane, offset, seq + 1ull, sess_prof_, &pe, &nb_unused, &ipc_unused, true);
                        if (src) { finish_session(); return src; }
                        sess_early_ = true;
                        sess_pass_ = 0;
                        timing_persist_launches_total_ += 1.0;
                    }
                }
            }
        }
        if (loop_budget_ > 0) loop_budget_--;
        (void)nblocks;
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
    auto wait_published = [&]() {
        // (inside a persistent launch: for as long as the launch lives -- a pass of a pathological configuration may take
        //  seconds, and a host that stopped looking would leave the launch waiting for a command until its patience ends)
        for (long long spin = 0; in_session || spin < 400000000ll; ++spin) {
            if (all_tagged()) return true;
            const long long every = in_session ? 0xFFFFll : 0xFFFFFll;
            if ((spin & every) == every && hipStreamQuery(stream_) != hipErrorNotReady) return all_tagged();
        }
        return false;
    };
    seen = wait_published();
    if (!seen && in_session) {
        // The stream is idle and the pass this thread asked for has not been published: the persistent launch ended by
        // itself (its workgroups were not all resident in time -- another process's launch held the rest of the compute
        // units --, or no command reached it within four times the host's patience).
        HIP_TRY(hipStreamSynchronize(stream_));
        seen = all_tagged();
        if (!seen) {
            timing_.persist_aborts += 1.0; persist_aborts_total_ += 1.0;
            if (trace_persist())
                std::fprintf(stderr, "[visma_icp] persistent launch gave up: pass %d of %d, its flag %u, tag %u (first %u)\n", sess_pass_, sess_max_,
                             *reinterpret_cast<volatile unsigned *>(h_flag_), cmd_tag_, sess_tag0_);
            finish_session();
            persist_cooldown_ = kPersistCooldownLoops;
            in_session = false;
            // Whatever made it leave (the host thread away for longer than four times its patience, a command that
            // arrived torn over more than that): some workgroups may have begun the pass and others not.  Nothing of a
            // half-run pass is kept: the fold's tickets are re-armed and the winners forgotten -- the pass runs cold.
            if (d_tickets_ && tickets_cap_ > 0) HIP_TRY(hipMemsetAsync(d_tickets_, 0, sizeof(unsigned) * tickets_cap_, stream_));
            { int irc = invalidate_pos(); if (irc) return irc; }
            int nblocks = 1;
            int rc = launch_grid_pass(T64, plane, offset, seq, false, nullptr, &nblocks, &ipc_done);
            if (rc) return rc;
            seen = wait_published();
        }
    }
    if (in_session && seen && posted) note_state_pass(T64);
    if (!in_session && seen && sess_live_ && sess_early_) t_stats_seen_ = std::chrono::steady_clock::now();   // (the early launch waits from here)
    if (in_session && seen) {
        const auto now = std::chrono::steady_clock::now();
        if (posted && trace_persist()) wait_us_ += std::chrono::duration<double, std::micro>(now - t_posted_).count();
        t_stats_seen_ = now;                                 // (the host's patience with itself counts from here)
    }
    if (in_session && seen && sess_live_ && sess_pass_ >= sess_max_) finish_session();   // (its last pass: the launch ends by itself)
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
    note_ring_stats(stats);                                  // (ring passes: lanes per query of the next one)
    return sess_live_ ? VISMA_ICP_OK : maybe_collect_timing();   // (no stream synchronisation while a session waits for the host)
}

// One search launch over the resident clouds (lane-serial, certificate, or -- persist != NULL -- the persistent form
// of the certificate kernel), with the fold inside the launch where that applies.
int HipEngine::launch_grid_pass(const Xform64 &T64, bool plane, const double offset[3], unsigned long long seq, bool prof,
                                const PersistArgs *persist, int *nblocks_out, bool *ipc_done, bool early)
{
    const bool ipc = ipc_n_ > 1;
    double *pub = (comm_ || ipc) ? nullptr : h_stats_dev_;
    // the fold of the partial rows runs inside the search launch (no second kernel)
    const bool fused = fused_fold_ && !tshard_;
    const int lanes = pass_lanes();
    int nblocks = 1, e0 = -1;
    FoldArgs fa{};
    if (fused) {
        // (peer-to-peer mailboxes: the folding workgroup exchanges with the peers and publishes itself)
        int rc = make_fold(grid_launch_blocks(ns_, lanes, grid_blocks()), 1, (double *)d_stats_, 0,
                           ipc ? h_stats_dev_ : pub, seq, &fa);
        if (rc) return rc;
        if (ipc) { add_ipc(&fa); *ipc_done = true; }
        if (persist) {
            // the persistent launch folds by polling: rows and group rows as tagged granules (no tickets)
            constexpr size_t kRows = 2048, kGroups = (kRows + kFoldGroup - 1) / kFoldGroup, kRowBytes = 32 * 16;
            static_assert(kFoldGroup >= 1 && kGroups <= kRows, "group rows of the polled fold");
            if (!d_fold_tag_) {
                HIP_TRY(hipMalloc(&d_fold_tag_, (kRows + kGroups) * kRowBytes));
                HIP_TRY(hipMemsetAsync(d_fold_tag_, 0, (kRows + kGroups) * kRowBytes, stream_));
            }
            if ((size_t)grid_launch_blocks(ns_, lanes, grid_blocks()) > kRows) { err_ = "persistent launch larger than its fold rows"; return VISMA_ICP_ERR_STATE; }
            // (the rows validate themselves with fold_row_tag(sequence number): when a session's numbers run through the
            //  tag's wrap, the buffer is cleared first -- a row of 2^32 - 1 passes ago must not validate)
            if (seq % kFoldTagPeriod + (unsigned long long)persist->max_passes >= kFoldTagPeriod)
                HIP_TRY(hipMemsetAsync(d_fold_tag_, 0, (kRows + kGroups) * kRowBytes, stream_));
            fa.rows_tagged = d_fold_tag_;
            fa.rows2_tagged = (char *)d_fold_tag_ + kRows * kRowBytes;
            fa.dead_flag = (unsigned long long *)d_relay_ + kPersistDead;
            fa.poll_ticks = persist->hard_ticks;
        }
    }
    if (prof) { e0 = next_event_pair(); HIP_TRY(hipEventRecord(ev_[e0], stream_)); }
    HIP_TRY(launch_nn_grid_reduce((const float4 *)d_src_, ns_, search_sorted(),
                                  (const unsigned *)d_start_, grid_, (const float4 *)d_nrm_,
                                  T32_, T64, offset, r2f_, plane ? 1 : 0, (int32_t *)d_idx_,
                                  (float *)d_d2_, (double *)d_partials_, grid_blocks(),
                                  &nblocks, lanes,
                                  prof ? (unsigned long long *)d_cand_ : nullptr, nullptr,
                                  1, 0, stream_, f64_src(), f64_sorted(), r2d_, (const Pt64 *)d_nrm64_,
                                  exact_ ? 1 : 0, fused ? &fa : nullptr, shard_d64(), (Pt64 *)d_pos_, 1 | (persist ? persist_prio_ << 5 : 0), cert_prev(), persist, ru_state(),
                                  ring_table()));
    if (!early) {
        // (an early persistent launch has run nothing yet: what its first pass leaves is noted when that pass has been seen)
        last_kernel_ = pass_kernel(lanes);
        pos_fresh_ = d_pos_ != nullptr;
        note_state_pass(T64);
    }
    if (prof) {
        HIP_TRY(hipEventRecord(ev_[e0 + 1], stream_));
        if (persist) sess_e0_ = e0;                          // (accounted when the session ends: its passes are known then)
        else pending_.push_back({e0, 0});
    }
    if (!tshard_ && !fused) {
        if (prof) { e0 = next_event_pair(); HIP_TRY(hipEventRecord(ev_[e0], stream_)); }
        HIP_TRY(launch_finalize((const double *)d_partials_, nblocks, plane ? 1 : 0,
                                (double *)d_stats_, stream_, pub, seq));
        if (prof) { HIP_TRY(hipEventRecord(ev_[e0 + 1], stream_)); pending_.push_back({e0, 1}); }
    }
    *nblocks_out = nblocks;
    return VISMA_ICP_OK;
}

// ---- persistent sessions -----------------------------------------------------------------------------------------
namespace {
std::atomic<int> g_persist_slot[64];   // one persistent launch per device and process at a time: two of them would
                                       // each hold a part of the compute units and wait for the rest
}

bool HipEngine::persist_possible(int lanes, int nblocks, bool fused, bool plane) const
{
    static const bool trace = std::getenv("VISMA_ICP_PERSIST_TRACE") != nullptr;
    auto no = [&](const char *why) {
        if (trace) std::fprintf(stderr, "[visma_icp] no persistent launch: %s\n", why);
        return false;
    };
    if (!persist_enabled_) return no("switched off");
    if (persist_cooldown_ > 0) return no("cooling down after a launch that gave up");
    if (!fused || tshard_ || comm_ || minreduce_ || (ipc_n_ > 1 && !persist_ranks_ok())) return no("sharded ranks / fold in a second launch");
    if (lanes != kCoopLanes || !coop_ok()) return no("not a pass of the certificate kernel");   // (pass_lanes: warm, or cold in a loop)
    if (grid_lanes_ > 0 && grid_lanes_ != kCoopLanes) return no("lanes forced");
    if (std::getenv("VISMA_ICP_COOP_KERNEL")) return no("kernel forced");       // (A/B runs of the two one-pass kernels)
    // one query per lane, every workgroup resident at once
    if ((int64_t)nblocks * kBlock < ns_) return no("several queries per lane");
    const int cap = coop_persist_capacity(plane ? 1 : 0);
    persist_slots_seen_ = cap;
    // (the process's share of the device: a launch that needs more runs one launch per pass, which the kernels of other
    //  streams interleave with -- a persistent launch holds its slots, spinning, for the length of the loop)
    const int allowed = (int)((double)cap * persist_cu_share());
    if (trace) std::fprintf(stderr, "[visma_icp] persistent launch: %d workgroups, device holds %d, this process may hold %d\n", nblocks, cap, allowed);
    return nblocks <= allowed;
}

void HipEngine::fill_persist_args(PersistArgs *pa)
{
    pa->host_cmd = h_cmd_dev_;
    pa->relay = (unsigned long long *)d_relay_;
    pa->direct = cmd_direct_ ? 1 : 0;
    pa->host_flag = h_flag_dev_;
    pa->max_passes = sess_max_;
    pa->tag0 = sess_tag0_;
    pa->wait_ticks = (long long)(4.0 * persist_timeout_ms_ * 1e5) + 1000000ll;   // (100 MHz; host patience x 4 + 10 ms)
    pa->hard_ticks = 60ll * 100000000ll;
    pa->start_ticks = (long long)(persist_start_ms_ * 1e5);
}

int HipEngine::start_session(const Xform64 &, bool plane, const double offset[3], unsigned long long seq, int, bool prof)
{
    if (device_ < 0 || device_ >= 64) return VISMA_ICP_OK;
    int expect = 0;
    if (!g_persist_slot[device_].compare_exchange_strong(expect, 1)) return VISMA_ICP_OK;   // somebody else's turn: ordinary launches
    sess_dev_slot_ = device_;
    sess_live_ = true;
    sess_pass_ = 0;
    sess_max_ = std::min(loop_budget_, 1 << 20);
    sess_tag0_ = cmd_tag_ + 1u;
    if (sess_tag0_ == 0u || sess_tag0_ + (unsigned)sess_max_ < sess_tag0_) { cmd_tag_ = 0u; sess_tag0_ = 1u; }   // (tags never 0, never wrap inside a session)
    cmd_tag_ = sess_tag0_ - 1u;
    sess_seq0_ = seq;
    sess_plane_ = plane;
    sess_prof_ = prof;
    sess_r2f_ = r2f_;
    for (int a = 0; a < 3; a++) sess_off_[a] = offset[a];
    sess_e0_ = -1;
    *reinterpret_cast<volatile unsigned *>(h_flag_) = 0u;
    return VISMA_ICP_OK;
}

// the next command, word by word: every word carries the tag, so the device accepts the block when all words show it
void HipEngine::post_command(const Xform64 &T64, unsigned cmd)
{
    const unsigned tag = ++cmd_tag_;
    volatile unsigned long long *c = h_cmd_;
    const unsigned long long t = (unsigned long long)tag << 32;
    for (int k = 0; k < 12; k++) {
        unsigned long long b;
        std::memcpy(&b, &T64.m[k], sizeof(b));
        c[2 * k] = (b & 0xFFFFFFFFull) | t;
        c[2 * k + 1] = (b >> 32) | t;
    }
    c[kPersistWords - 1] = (unsigned long long)cmd | t;
    std::atomic_thread_fence(std::memory_order_release);
    if (cmd_direct_) VISMA_STORE_FENCE();                       // (the BAR is write-combining: the stores leave the core now, not when its buffers fill)
}

void HipEngine::finish_session()
{
    if (!sess_live_) return;
    sess_live_ = false;
    sess_early_ = false;
    if (host_gaps_ > 0) {
        std::fprintf(stderr, "[visma_icp] persistent launch: %d passes; host: statistics seen -> command posted %.2f us, command posted -> statistics seen %.2f us (averages)\n",
                     sess_pass_, host_gap_us_ / host_gaps_, wait_us_ / host_gaps_);
        host_gap_us_ = wait_us_ = 0.0;
        host_gaps_ = 0;
    }
    if (d_timeline_ && timeline_passes_ > 0 && !timeline_path_.empty()) {
        // record: {passes recorded, workgroups, passes run}, then [pass][workgroup]{begin, body done} (100 MHz ticks)
        if (hipStreamSynchronize(stream_) == hipSuccess) {
            const size_t n = (size_t)2 * timeline_passes_ * timeline_blocks_;
            std::vector<unsigned long long> buf(n + 3);
            buf[0] = (unsigned long long)timeline_passes_; buf[1] = (unsigned long long)timeline_blocks_; buf[2] = (unsigned long long)sess_pass_;
            if (hipMemcpy(buf.data() + 3, d_timeline_, n * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess)
                if (FILE *f = std::fopen(timeline_path_.c_str(), "ab")) { std::fwrite(buf.data(), sizeof(unsigned long long), buf.size(), f); std::fclose(f); }
        }
        (void)hipGetLastError();
        timeline_passes_ = 0;
    }
    if (sess_e0_ >= 0) sess_pending_.push_back({sess_e0_, std::max(sess_pass_, 1)});
    sess_e0_ = -1;
    if (sess_dev_slot_ >= 0) g_persist_slot[sess_dev_slot_].store(0);
    sess_dev_slot_ = -1;
}

// STOP to a launch that still waits for a command (one that has run its last pass ends by itself), then its end
int HipEngine::end_session()
{
    if (!sess_live_) return VISMA_ICP_OK;
    // (an early launch waits for a command before its FIRST pass too)
    if ((sess_pass_ >= 1 || sess_early_) && sess_pass_ < sess_max_) {
        Xform64 none{};
        post_command(none, kPersistStop);
    }
    hipError_t e = hipStreamSynchronize(stream_);
    finish_session();
    if (e != hipSuccess) { err_ = std::string("persistent launch: ") + hipGetErrorString(e); (void)hipGetLastError(); return VISMA_ICP_ERR_HIP; }
    return VISMA_ICP_OK;
}

int HipEngine::get_correspondences(int32_t *idx, float *d2)
{
    HIP_TRY(hipSetDevice(device_));
    if (sess_live_) { int src = end_session(); if (src) return src; }   // (never behind a launch that waits for this thread)
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
                                      nullptr, nullptr, (Pt64 *)d_pos_, 1, cert_prev(), nullptr, ru_state(), ring_table()));
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

int HipEngine::run_loop(const LoopParams &lp, const Mat4 *Tc0s, int nprob, LoopResult *out)
{
    HIP_TRY(hipSetDevice(device_));
    if (sess_live_) { int src = end_session(); if (src) return src; }
    if (nprob < 1) { err_ = "nprob < 1"; return VISMA_ICP_ERR_INVALID; }
    if (!d_src_ || !d_tgt_) { err_ = "clouds not set"; return VISMA_ICP_ERR_STATE; }
    if (ring_lanes_auto_) ring_lanes_ = 8;                   // (no statistics reach the host between the passes of this loop)
    last_was_batch_ = false;
    if (lp.plane && !d_nrm_) { err_ = "point-to-plane needs target normals"; return VISMA_ICP_ERR_STATE; }
    caller_sweep_ = nprob > 1;
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
    bool sweep_tried = false;
    // (a loop the persistent sweep launch may take over runs its first pass alone: the launch starts behind it)
    const bool sweep_candidate = sweep_persist_ && use_grid_ && fused_fold_ && !tshard_ && !comm_ && ipc_n_ <= 1 && !lp.plane &&
                                 lp.solver == VISMA_ICP_SOLVER_KABSCH && lp.passes >= 3;
    while (done < lp.passes) {
        int n = std::min(chunk, lp.passes - done);
        if (done == 0 && sweep_candidate) n = 1;
        // ---- the persistent sweep launch (hip_engine.hpp: sweep_persist_): every pass after the first of every problem in ONE
        // launch.  Tried once per loop; a launch that gave up (a wait ran out: its workgroups were not all resident) leaves the
        // states where they got to, and the loop carries on below with one launch per pass.
        if (done >= 1 && !sweep_tried && sweep_candidate && pass_lanes(nprob) == kCoopLanes && coop_ok() && device_ >= 0 && device_ < 64 &&
            !std::getenv("VISMA_ICP_COOP_KERNEL")) {
            sweep_tried = true;
            const int nb = grid_launch_blocks(ns_, kCoopLanes, reduce_max_blocks());
            const int cap = nn_wave_sweep_capacity();
            int expect = 0;
            if ((int64_t)nb * kBlock >= ns_ && (int64_t)nb * nprob <= (int64_t)((double)cap * persist_cu_share()) &&
                g_persist_slot[device_].compare_exchange_strong(expect, 1)) {
                struct SlotGuard { int d; ~SlotGuard() { g_persist_slot[d].store(0); } } guard{device_};
                if (nprob > sweep_relay_cap_) {
                    if (d_sweep_relay_) (void)hipFree(d_sweep_relay_);
                    d_sweep_relay_ = nullptr; sweep_relay_cap_ = 0;
                    HIP_TRY(hipMalloc(&d_sweep_relay_, sizeof(unsigned long long) * (32 * (size_t)nprob + 8)));
                    sweep_relay_cap_ = nprob;
                    sweep_tag_ = 0;
                }
                const int rem = lp.passes - done;
                if (sweep_tag_ == 0u || sweep_tag_ + (unsigned)rem + 2u < sweep_tag_) {
                    // (tags never 0, never wrap inside a launch: a cleared relay validates nothing)
                    HIP_TRY(hipMemsetAsync(d_sweep_relay_, 0, sizeof(unsigned long long) * (32 * (size_t)sweep_relay_cap_ + 8), stream_));
                    sweep_tag_ = 1u;
                }
                unsigned long long *dead = (unsigned long long *)d_sweep_relay_ + 32 * (size_t)sweep_relay_cap_;
                HIP_TRY(hipMemsetAsync(dead, 0, sizeof(unsigned long long), stream_));
                FoldArgs fa{};
                rc = make_fold(nb, nprob, st->stats, (long long)(sizeof(DevIcpState) / sizeof(double)), nullptr, 0, &fa);
                if (rc) return rc;
                SweepArgs sa{};
                sa.relay = (unsigned long long *)d_sweep_relay_;
                sa.dead = dead;
                sa.max_passes = rem;
                sa.tag0 = sweep_tag_;
                sa.passes0 = done;                             // (every problem still active has run `done` passes)
                sa.wait_ticks = (long long)(50.0 * 1e5);      // 50 ms (100 MHz): a pass of a resident launch takes microseconds
                sweep_tag_ += (unsigned)rem + 1u;
                int e0 = -1;
                if (profiling_) { e0 = next_event_pair(); HIP_TRY(hipEventRecord(ev_[e0], stream_)); }
                HIP_TRY(launch_nn_wave_sweep(nb, nprob, (int)ns_, (const float *)d_sorted12_, (const unsigned *)d_start_, grid_, r2f_,
                                             (int32_t *)d_idx_, (float *)d_d2_, (double *)d_partials_,
                                             profiling_ ? (unsigned long long *)d_cand_ : nullptr, st, loop_out_stride_,
                                             (const Pt64 *)d_src64_, (const Pt64 *)d_sorted64_, fa, (Pt64 *)d_pos_, sa, stream_));
                if (profiling_) { HIP_TRY(hipEventRecord(ev_[e0 + 1], stream_)); pending_.push_back({e0, 0}); }
                last_kernel_ = 2;
                pos_fresh_ = d_pos_ != nullptr;
                prev_T_valid_ = false;
                unsigned long long dead_h = 0ull;
                HIP_TRY(hipMemcpyAsync(h_state_, d_state_, sizeof(DevIcpState) * nprob, hipMemcpyDeviceToHost, stream_));
                HIP_TRY(hipMemcpyAsync(&dead_h, dead, sizeof(dead_h), hipMemcpyDeviceToHost, stream_));
                HIP_TRY(hipStreamSynchronize(stream_));
                sweep_launches_total_ += 1.0;
                rc = maybe_collect_timing();
                if (rc) return rc;
                if (!dead_h) { done = lp.passes; break; }
                // gave up: every problem is at least where its slowest state says; carry on from there with launches
                sweep_aborts_total_ += 1.0;
                int least = lp.passes;
                bool any = false;
                for (int b = 0; b < nprob; b++)
                    if (h_state_[b].active) { any = true; least = std::min(least, h_state_[b].passes); }
                if (!any) break;
                done = std::max(done, std::min(least, lp.passes - 1));
                n = std::min(chunk, lp.passes - done);
            }
        }
        for (int j = 0; j < n; j++) {
            int nblocks = 1, e0 = -1;
            if (profiling_) { e0 = next_event_pair(); HIP_TRY(hipEventRecord(ev_[e0], stream_)); }
            bool fused = false, solved_in_fold = false;
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
                    // closed-form update on one GPU: the workgroup that completes a problem's fold advances its state
                    // in the same launch (no solve_state_kernel between two search launches)
                    if (solve_in_fold(lp, lanes, nprob)) fa.solve = st;
                }
                solved_in_fold = fused && fa.solve != nullptr;
                HIP_TRY(launch_nn_grid_reduce((const float4 *)d_src_, ns_, search_sorted(),
                                              (const unsigned *)d_start_, grid_, (const float4 *)d_nrm_,
                                              T32_, T64, nullptr, r2f_, plane, (int32_t *)d_idx_,
                                              (float *)d_d2_, (double *)d_partials_,
                                              reduce_max_blocks(), &nblocks, lanes,
                                              profiling_ ? (unsigned long long *)d_cand_ : nullptr, st,
                                              nprob, loop_out_stride_, stream_, f64_src(), f64_sorted(), r2d_, (const Pt64 *)d_nrm64_,
                                              exact_ ? 1 : 0, fused ? &fa : nullptr, tshard_ ? shard_d64() : nullptr,
                                              (Pt64 *)d_pos_, cert_enabled_ ? 1 : (1 | 8), nullptr, nullptr, nprob == 1 ? ru_state() : nullptr,
                                              ring_table()));
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
                if (!solved_in_fold) HIP_TRY(launch_solve_state(st, nprob, stream_));
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

int HipEngine::run_loop_batch(const LoopParams &lp, const std::vector<BatchProblem> &pb, LoopResult *out)
{
    HIP_TRY(hipSetDevice(device_));
    if (sess_live_) { int src = end_session(); if (src) return src; }
    const int B = (int)pb.size();
    if (B < 1) return VISMA_ICP_OK;
    if (comm_) { err_ = "batched loop is single-GPU"; return VISMA_ICP_ERR_STATE; }
    last_was_batch_ = true;
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
        const size_t tstride = 1 + (size_t)(max_nb + kFoldGroup - 1) / kFoldGroup;
        int rc2 = ensure_tile_buffers((size_t)total_blocks, tstride * B);
        if (rc2) return rc2;
        bfa.tickets = (unsigned *)d_tickets_;
        bfa.partials2 = (double *)d_partials2_;
        bfa.ticket_stride = (int)tstride;
        bfa.stats_out = st->stats;
        bfa.stats_stride = (long long)(sizeof(DevIcpState) / sizeof(double));
        if (solve_in_fold(lp)) bfa.solve = st;               // (see run_loop)
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
            if (fused_fold_) { if (!bfa.solve) HIP_TRY(launch_solve_state(st, B, stream_)); }
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

int HipEngine::make_fold(int bpp, int nprob, double *stats_out, long long stats_stride, double *host_out,
              unsigned long long seq, FoldArgs *out)
{
    const size_t tstride = 1 + (size_t)(bpp + kFoldGroup - 1) / kFoldGroup;
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

}  // namespace drv
}  // namespace visma
