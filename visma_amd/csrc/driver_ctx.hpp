// driver_ctx.hpp -- struct visma_icp_ctx: the host driver's state and its loops (internal).
#pragma once
#include "engine.hpp"

using namespace visma;
using namespace visma::drv;

// ---- the driver -----------------------------------------------------------------
struct visma_icp_ctx {
    std::unique_ptr<Engine> eng;
    std::string err;
    double centre[3] = {0, 0, 0};
    double radius_hint = 0.0;         // visma_icp_set_radius_hint: the search radius the next registration will use
    double last_radius = 0.0;         // ... else the one the last registration used (a caller's loop keeps it)
    int search_precision = 1;         // 0 fp32 search, 1 f64 for small clouds (auto, default), 2 f64 always
    bool fixed_centre = false;        // centre given by the caller (target-sharded ranks share one)
    bool target_sharded = false;
    bool have_src = false, have_tgt = false;
    // Frame bookkeeping: visma_icp_set_clouds_f64 centres BOTH clouds on one point; the fp32 / device
    // setters upload a cloud as given (centre 0).  A cloud uploaded in the other frame cannot be combined
    // with it: the setter that changes the frame invalidates the other cloud (it has to be set again).
    bool centred_upload = false;
    void enter_uncentred_frame(bool setting_source)
    {
        if (centred_upload && (centre[0] != 0.0 || centre[1] != 0.0 || centre[2] != 0.0)) {
            if (setting_source) have_tgt = false; else have_src = false;
        }
        centred_upload = false;
        centre[0] = centre[1] = centre[2] = 0.0;
    }
    visma_icp_allreduce_fn host_allreduce = nullptr;
    void *host_allreduce_user = nullptr;
    int rank = 0, nranks = 1;
    int64_t ns_total = 0;
    Mat4 last_Tc = Mat4::identity();
    bool last_plane = false;
    std::vector<int32_t> src_order;   // engine position -> caller's source index (Morton order)
    double last_aux_kernel_ms = 0.0;  // kernel time of the last mesh-distance call
    double last_aux_build_ms = 0.0;   // ... and of building its search structure
    int mesh_method = 0;              // 0 choose, 1 brute force, 2 BVH

    int fail(int code, const std::string &msg) { err = msg; return code; }
    int eng_fail(int code) { err = eng->error(); return code; }

    // one NN pass + reduction (+ cross-rank sum); fills stats, fitness, rmse
    // world_frame: express the statistics in the caller's frame (needed by the
    // Gauss-Newton updates, whose Euler / exp-map retraction is not invariant to
    // the choice of origin); the closed-form solve uses the centred frame.
    int pass(const Mat4 &Tc, double max_dist, bool plane, bool world_frame, double *stats,
             double *fit, double *rmse, int64_t *k)
    {
        int rc = eng->nn_pass(Tc, max_dist);
        if (rc) return eng_fail(rc);
        last_Tc = Tc;
        last_plane = plane;
        const double zero[3] = {0, 0, 0};
        rc = eng->reduce(Tc, plane, world_frame ? centre : zero, stats);
        if (rc) return eng_fail(rc);
        if (host_allreduce && !eng->has_device_allreduce()) {
            if (host_allreduce(host_allreduce_user, stats, VISMA_ICP_NSTATS) != 0)
                return fail(VISMA_ICP_ERR_ENGINE, "host all-reduce callback failed");
        }
        const double K = stats[0];
        const int64_t denom = ns_total > 0 ? ns_total : eng->ns();
        *k = (int64_t)std::llround(K);
        if (K > 0.0) {  // Registration.cpp:87-94
            *fit = K / (double)denom;
            *rmse = std::sqrt(stats[1] / K);
        } else {
            *fit = 0.0;
            *rmse = 0.0;
        }
        return VISMA_ICP_OK;
    }

    // 0 = synchronous host loop, 1 = on-device loop, 2 = auto: host loop for one
    // problem (spin-wait on mapped memory beats a one-thread f64 SVD on the GPU:
    // measured 46 vs 68 us per iteration at 5k x 20k), device loop for sweeps of
    // many transforms (their solves run in parallel and nothing syncs per pass)
    int loop_mode = 2;
    bool device_loop_possible() const
    {
        return eng->supports_device_loop() && !host_allreduce && (!target_sharded || eng->shard_loop_on_device());
    }
    bool use_device_loop() const { return loop_mode == 1 && device_loop_possible(); }
    bool use_device_loop_batched() const { return loop_mode != 0 && device_loop_possible(); }
    // one rank, nothing summed on the host between a pass and the next: the engine may keep one launch alive across
    // the passes of a loop (ranks that share a GPU would wait for each other's workgroups)
    // (source-sharded ranks that exchange through their peer-to-peer mailboxes, one rank per GPU: the exchange happens
    //  inside the launch, so it may stay alive there too -- the engine knows whether the ranks share a device)
    bool solo() const { return !host_allreduce && !target_sharded && (nranks == 1 || eng->loop_across_ranks_ok()); }
    static bool wants_world_frame(int solver, bool plane) { return plane || solver != VISMA_ICP_SOLVER_KABSCH; }

    // T_centred <- update o T_centred, with the update expressed in `world` or centred frame
    Mat4 apply_update(const Mat4 &upd, const Mat4 &Tc, bool world_frame) const
    {
        if (!world_frame) return upd * Tc;
        return to_centred(upd * from_centred(Tc, centre), centre);
    }

    Mat4 solve(const double *stats, int solver, bool scaling, bool plane) const
    {
        bool ok;
        if (plane) return gn_from_stats(stats, false, &ok);  // TransformationEstimation.cpp:94-102
        switch (solver) {
        case VISMA_ICP_SOLVER_GN_EULER: return gn_from_stats(stats, false, &ok);
        case VISMA_ICP_SOLVER_GN_EXPMAP: return gn_from_stats(stats, true, &ok);
        default: return kabsch_from_stats(stats, scaling);
        }
    }

    int run(const double *init, double max_dist, int max_iter, double rel_fit, double rel_rmse,
            int solver, bool scaling, bool plane, visma_icp_result *out)
    {
        std::memset(out, 0, sizeof(*out));
        std::memcpy(out->transformation, init, sizeof(double) * 16);
        if (!(max_dist > 0.0)) return VISMA_ICP_OK;                 // Registration.cpp:148-151
        last_radius = max_dist;
        if (plane && !eng->has_normals()) return VISMA_ICP_OK;      // Registration.cpp:152-157
        if (!have_src || !have_tgt) return fail(VISMA_ICP_ERR_STATE, "clouds not set");
        Mat4 Tc = to_centred(Mat4::from(init), centre);
        const bool world = wants_world_frame(solver, plane);
        if (use_device_loop()) {
            Engine::LoopParams lp;
            lp.Tc0 = Tc;
            std::memcpy(lp.centre, centre, sizeof(centre));
            lp.max_dist = max_dist; lp.rel_fit = rel_fit; lp.rel_rmse = rel_rmse;
            lp.max_iter = max_iter; lp.solver = solver; lp.passes = max_iter + 1;
            lp.scaling = scaling; lp.plane = plane; lp.world = world; lp.check_stop = true;
            lp.ns_total = ns_total > 0 ? ns_total : eng->ns();
            Engine::LoopResult r;
            int rc = eng->run_loop(lp, nullptr, 1, &r);
            if (rc) return eng_fail(rc);
            last_Tc = r.Tc;
            last_plane = plane;
            const Mat4 T = from_centred(r.Tc, centre);
            std::memcpy(out->transformation, T.m, sizeof(T.m));
            out->fitness = r.fit;
            out->inlier_rmse = r.rmse;
            out->num_correspondences = r.k;
            out->iterations = r.iters;
            out->nn_passes = r.passes;
            return VISMA_ICP_OK;
        }
        double stats[VISMA_ICP_NSTATS], fit, rmse;
        int64_t k;
        Engine::LoopScope scope(eng.get(), solo() ? max_iter + 1 : 0);   // (at most max_iter + 1 passes follow, nothing else)
        int rc = pass(Tc, max_dist, plane, world, stats, &fit, &rmse, &k);  // Registration.cpp:166-168
        if (rc) return rc;
        int it = 0;
        for (int i = 0; i < max_iter; i++) {                          // Registration.cpp:169-184
            const Mat4 upd = solve(stats, solver, scaling, plane);
            Tc = apply_update(upd, Tc, world);
            const double bfit = fit, brmse = rmse;
            rc = pass(Tc, max_dist, plane, world, stats, &fit, &rmse, &k);
            if (rc) return rc;
            it = i + 1;
            if (std::fabs(bfit - fit) < rel_fit && std::fabs(brmse - rmse) < rel_rmse) break;
        }
        const Mat4 T = from_centred(Tc, centre);
        std::memcpy(out->transformation, T.m, sizeof(T.m));
        out->fitness = fit;
        out->inlier_rmse = rmse;
        out->num_correspondences = k;
        out->iterations = it;
        out->nn_passes = it + 1;
        return VISMA_ICP_OK;
    }
};

#define CTX_CHECK()                                                             \
    if (!ctx) { g_create_error = "ctx is NULL"; return VISMA_ICP_ERR_INVALID; }
