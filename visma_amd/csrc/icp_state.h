// icp_state.h -- one step of the device-resident ICP loop on a DevIcpState: the per-iteration solve, the compose
// T <- update * T and the stop test of O3D/Core/Registration/Registration.cpp:169-184, from the 38 statistics the fold
// left in the state.  Runs in one thread: in solve_state_kernel / finalize_solve_kernel (icp_loop.hip), and -- round 5 --
// in the fold epilogue of the batch search launches themselves (device_common.h: fused_fold<..., SOLVE>), where the
// workgroup that completes a problem's fold advances that problem at once.
// The solve is the SAME code as the host's (host_math.hpp is host+device).
#pragma once

#include "host_math.hpp"
#include "kernels.h"

namespace visma {

// KABSCH_ONLY: the closed-form update alone (the caller has checked !plane && solver == 0): without the 6 x 6
// Gauss-Newton paths the step needs ~60 registers instead of > 128 (inside a search kernel held to 128 it must not spill)
template <bool KABSCH_ONLY = false>
__device__ __forceinline__ void advance_state(DevIcpState *st)
{
    const double *stats = st->stats;
    const double K = stats[0];
    // fitness / rmse of the pass just finished (Registration.cpp:87-94)
    double fit = 0.0, rmse = 0.0;
    if (K > 0.0) {
        fit = K / (double)st->ns_total;
        rmse = sqrt(stats[1] / K);
    }
    st->K = K;
    st->fit = fit;
    st->rmse = rmse;
    st->passes += 1;
    bool stop = false;
    if (st->check_stop && st->passes >= 2 && fabs(st->fit_prev - fit) < st->rel_fit &&
        fabs(st->rmse_prev - rmse) < st->rel_rmse)
        stop = true;                                   // Registration.cpp:179-183
    if (st->iter >= st->max_iter) stop = true;         // loop bound, :169
    if (stop) {
        st->active = 0;
        return;
    }
    // update = estimation.ComputeTransformation(...)  (:172-173), from the moments
    Mat4 upd;
    bool ok = true;
    if constexpr (KABSCH_ONLY) {
        upd = kabsch_from_stats(stats, st->scaling != 0);
    } else {
        if (st->plane || st->solver == 1)
            upd = gn_from_stats(stats, false, &ok);
        else if (st->solver == 2)
            upd = gn_from_stats(stats, true, &ok);
        else
            upd = kabsch_from_stats(stats, st->scaling != 0);
    }
    // transformation = update * transformation  (:174), in f64
    Mat4 Tc = Mat4::identity();
    for (int i = 0; i < 12; i++) Tc.m[i] = st->Tc[i];
    Mat4 Tn;
    if (st->world_frame)
        Tn = to_centred(upd * from_centred(Tc, st->centre), st->centre);
    else
        Tn = upd * Tc;
    // (the pose the pass just folded was searched at: what the next pass's certificates measure their motion from)
    for (int i = 0; i < 12; i++) st->Tc_prev[i] = st->Tc[i];
    st->have_prev = 1;
    for (int i = 0; i < 12; i++) st->Tc[i] = Tn.m[i];
    st->iter += 1;
    st->fit_prev = fit;
    st->rmse_prev = rmse;
}

}  // namespace visma
