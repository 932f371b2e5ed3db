// host_math.h -- host-side pieces of the tracking path: normal-equation assembly and
// solve, LM / trust-region control, spline evaluation and update.
//
//   merge_blocks_host            ba_tracker/merge_hessian_gradient_cost.cpp:39-86
//   solve_normal_equation_host   ba_tracker/solve_normal_equation.h:10-35 (Eigen JacobiSVD / LDLT there;
//                                own one-sided Jacobi SVD and pivoted LDL^T here, no Eigen dependency)
//   LevenbergMarquardtStrategy   ba_tracker/levenberg_marquardt_strategy.{h,cpp}
//   TrustRegionStepEvaluator     ba_tracker/trust_region_step_evaluator.{h,cpp}
//   SplineSE3                    core/common/Spline.h:14-369 (flat double storage instead of Eigen types)
#ifndef MBAVO_HOST_MATH_H
#define MBAVO_HOST_MATH_H

#include "options.h"
#include <vector>

namespace mbavo
{
    void merge_blocks_host(int F, int k, const double *frame_blocks, const int *start_idx, int N,
                           double *total_cost, double *H_colmajor, double *g);

    // x = -pinv(A) b (type 0) or -A^{-1} b via LDL^T (type 1); returns rank, or -1 for an unknown type
    // fast_ratio: pivot ratio up to which LDL^T stands in for the Jacobi SVD (solver type 0); 0 = never; < 0 = ask the
    // environment now (MBAVO_FAST_SOLVE, default 1e8).  The LM loops read it once per call and pass it down.
    int solve_normal_equation_host(const double *A_colmajor, const double *b, int n, int solver_type, double *x, double fast_ratio = -1.0);
    // (the admitted pivot ratio of the LDL^T stand-in: options.h opt_fast_ratio)
} // namespace mbavo

namespace SLAM
{
    namespace VO
    {
        class LevenbergMarquardtStrategy
        {
        public:
            LevenbergMarquardtStrategy();
            void reset();
            void step_accepted(double step_quality);
            void step_rejected();
            double get_radius();

        private:
            double mRadius, mMaxRadius, mMinRadius, mDecreaseFactor;
        };

        class TrustRegionStepEvaluator
        {
        public:
            explicit TrustRegionStepEvaluator(int max_consecutive_nonmonotonic_steps);
            void reset(double initial_cost);
            double StepQuality(double cost, double model_cost_change) const;
            void StepAccepted(double cost, double model_cost_change);

        private:
            const int max_consecutive_nonmonotonic_steps_;
            double minimum_cost_, current_cost_, reference_cost_, candidate_cost_;
            double accumulated_reference_model_cost_change_, accumulated_candidate_model_cost_change_;
            int num_consecutive_nonmonotonic_steps_;
        };
    } // namespace VO

    namespace Core
    {
        // SE(3) B-spline with uniformly spaced control knots; poses are body-to-world,
        // quaternions stored x,y,z,w like Eigen's coeffs() (Spline.h:125-133).
        class SplineSE3
        {
        public:
            SplineSE3() : mDt(0), mT0(0), mDegK(4) {}
            SplineSE3(double start_time, double dt) : mDt(dt), mT0(start_time), mDegK(4) {}
            SplineSE3 *clone() const { return new SplineSE3(*this); }

            void setStartTime(double t0) { mT0 = t0; }
            void setSamplingFreq(double dt) { mDt = dt; }
            void setSplineDegK(int k) { mDegK = k; }
            double getStartTime() const { return mT0; }
            double getSamplingFreq() const { return mDt; }
            int getSplineDegK() const { return mDegK; }
            size_t get_num_knots() const { return mT.size() / 3; }
            double *get_knot_data_t() { return mT.data(); }
            double *get_knot_data_R() { return mR.data(); }
            const double *get_knot_data_t() const { return mT.data(); }
            const double *get_knot_data_R() const { return mR.data(); }

            void InsertControlKnot(const double q_xyzw[4], const double t[3]);
            void PopFrontControlKnot();
            void Clear() { mT.clear(); mR.clear(); }

            // pose at time t; optional 4x3k / 3x3k row-major Jacobians; false if t is outside the knot range
            bool GetPose(double t, double q_xyzw[4], double t_out[3], double *jacobian_R = nullptr,
                         double *jacobian_t = nullptr) const;
            void TransformByRight(const double dq_xyzw[4], const double dt[3]);
            // Spline.h:183-200: re-express the spline so that its pose at time t becomes (q, t); false if t is out of range
            bool TransformTo(double t, const double q_xyzw[4], const double t_target[3]);
            void UpdateCtrlKnot_t(int start_knot_idx, int num_knots, const double *dt);
            void UpdateCtrlKnot_R(int start_knot_idx, int num_knots, const double *dR);
            void Plus_t(const double *dt, double *candidate_t) const;
            void Plus_R(const double *dR, double *candidate_R) const;
            void InvalidParameter(const double *data_t, const double *data_R);
            void ResetIdentity();

        private:
            std::vector<double> mT, mR;
            double mDt, mT0;
            int mDegK;
        };
    } // namespace Core
} // namespace SLAM

#endif
