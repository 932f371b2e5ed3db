// lm_state.h -- per-problem state of the Levenberg-Marquardt loop on the device and the scalar rules that drive it
// (device code shared by lm_batch.hip: B problems, one kernel per step of the loop -- and engine.hip: ONE problem, the
// whole loop of a pyramid level inside one resident kernel).
//
//   LM radius            ba_tracker/levenberg_marquardt_strategy.cpp:9-45
//   step evaluator       ba_tracker/trust_region_step_evaluator.cpp:45-126
//   loop                 ba_tracker/blur_aware_direct_tracker.cpp:590-699,799-924 (see tracker.cpp for the quirks kept)
#ifndef MBAVO_LM_STATE_H
#define MBAVO_LM_STATE_H

#include "../../include/mbavo.h"
#include <cfloat>
#include <hip/hip_runtime.h>

namespace mbavo
{
    struct LmState
    {
        double radius, decrease_factor;                                                        // LM strategy
        double minimum_cost, current_cost, reference_cost, candidate_cost, acc_ref, acc_cand;  // step evaluator
        double eval_cost, cand_cost, model, abs_dec, quality, initial_cost;
        int num_nonmono, iter, done, fresh, pending_accept, num_bad, n_accept, n_reject, n_invalid, ntrace;
    };
    static_assert(sizeof(LmState) % 8 == 0, "moved between leaders as 8-byte words");

    struct LmOpts
    {
        int max_it, max_nonmono, solver, trace_cap, max_n, max_N;
        double min_q, min_dec, chi;
        double fast_ratio; // solver 0: pivot ratio up to which the LDL^T result stands in for the Jacobi SVD's (0: never)
        double refined_ratio = 0.0; // batched LM: ... up to which the LDL^T result refined in double-double does (lm_solvers.h)
    };

    namespace
    {
        __device__ __forceinline__ void lm_clamp(LmState &s) { s.radius = fmax(fmin(1e32, s.radius), 10.0); }
        __device__ __forceinline__ void lm_reset(LmState &s) { s.radius = 1e4; s.decrease_factor = 2.0; }
        __device__ __forceinline__ void lm_accepted(LmState &s, double q)
        {
            // (2 q - 1)^3 as two products: the cube of levenberg_marquardt_strategy.cpp:29 within one unit in the last place, like
            // the device's pow() -- which is ~200 dependent instructions (~0.8 us) on the chain of every accepted step
            const double c = 2.0 * q - 1.0;
            s.radius = s.radius / fmax(1.0 / 3.0, 1.0 - c * c * c);
            lm_clamp(s);
            s.decrease_factor = 2.0;
        }
        __device__ __forceinline__ void lm_rejected(LmState &s)
        {
            s.radius = s.radius / s.decrease_factor;
            lm_clamp(s);
            s.decrease_factor *= 2.0;
        }
        __device__ __forceinline__ void tr_reset(LmState &s, double c)
        {
            s.minimum_cost = s.current_cost = s.reference_cost = s.candidate_cost = c;
            s.acc_ref = s.acc_cand = 0.0;
            s.num_nonmono = 0;
        }
        __device__ __forceinline__ double tr_quality(const LmState &s, double cost, double mcc)
        {
            if (cost >= DBL_MAX) return -DBL_MAX;
            const double now = (s.current_cost - cost) / mcc;
            const double hist = (s.reference_cost - cost) / (s.acc_ref + mcc);
            return now < hist ? hist : now; // std::max(now, hist) as the reference's libstdc++ evaluates it: an unordered comparison hands back `now` (NaN included; trust_region_step_evaluator.cpp:74, tests/golden tr_edge_*)
        }
        __device__ __forceinline__ void tr_accepted(LmState &s, double cost, double mcc, int max_nonmono)
        {
            s.current_cost = cost;
            s.acc_cand += mcc;
            s.acc_ref += mcc;
            if (s.current_cost < s.minimum_cost)
            {
                s.minimum_cost = s.candidate_cost = s.current_cost;
                s.num_nonmono = 0;
                s.acc_cand = 0.0;
            }
            else
            {
                ++s.num_nonmono;
                if (s.current_cost > s.candidate_cost)
                {
                    s.candidate_cost = s.current_cost;
                    s.acc_cand = 0.0;
                }
            }
            if (s.num_nonmono == max_nonmono)
            {
                s.reference_cost = s.candidate_cost;
                s.acc_ref = s.acc_cand;
            }
        }
        // one record of the loop's trace (what record() writes in tracker.cpp); the count runs on past the capacity
        __device__ __forceinline__ void trace_push(LmState &s, mbavo_trace_rec *trace, int cap, int lane, int level, int kind, double cc,
                                                   double model, double q)
        {
            if (trace && s.ntrace < cap && lane == 0)
            {
                mbavo_trace_rec &r = trace[s.ntrace];
                r.level = level; r.iter = s.iter; r.kind = kind; r.num_outliers = s.num_bad;
                r.radius = s.radius; r.eval_cost = s.eval_cost; r.candidate_cost = cc; r.model_change = model; r.quality = q;
            }
            ++s.ntrace;
        }
    } // namespace
} // namespace mbavo

#endif
