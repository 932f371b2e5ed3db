// tracker.h -- LM loop over the pyramid on the fused engine (see tracker.cpp).
#ifndef MBAVO_TRACKER_H
#define MBAVO_TRACKER_H

#include "engine.h"

namespace mbavo
{
    int optimize_trajectory(Engine &eng, const mbavo_track_opts &opts, const mbavo_level *levels, int F,
                            const double *h_cap, const double *h_exp, double t0, double dt, double *knots_t,
                            double *knots_R, int N, int *start_idx_out, double *final_cost, mbavo_trace_rec *trace,
                            int trace_cap);
    // lm_batch.hip: the same loop for B one-level problems with all control state on the device
    // `shared`: what a GROUP of a bigger batch takes from the whole batch (c_api.cpp: mbavo_lm_batch) so that every group runs the
    // same kernel form with the same strides
    struct LmBatchShared { int max_N = 0, max_S = 1; };
    int lm_batch(Engine &eng, int B, const mbavo_problem *probs, const mbavo_lm_batch_opts &opt, mbavo_lm_batch_result *results,
                 mbavo_trace_rec *trace, int trace_cap, const LmBatchShared *shared = nullptr);
}

#endif
