// vo_frontend.h -- the caller of the hot path: BlurAwareDirectTracker::trackFrame and what it needs
// (SURVEY.md 8f rows 2-3).
//
//   Core::Transformation            core/states/Transformation.{h,cpp} (Eigen / Sophus replaced by flat doubles;
//                                   exp / log follow Sophus::SE3d's closed forms)
//   VO::BlurAwareDirectTrackerOptions  ba_tracker/blur_aware_direct_tracker.h:15-67
//   VO::BlurAwareDirectTracker      ba_tracker/blur_aware_direct_tracker.cpp:14-415 (ctor, trackFrame, isKeyframe,
//                                   tmpProcessKeyframe, optimizeTrajectory) on device-resident pyramids
//
// Core::Frame / Core::CameraBase (sensor and dataset classes, out of scope) are replaced by a POD frame view and the
// pinhole intrinsics already in the options; GUI members are dropped.
#ifndef MBAVO_VO_FRONTEND_H
#define MBAVO_VO_FRONTEND_H

#include "engine.h"
#include "host_math.h"
#include <vector>

struct mbavo_trace_rec; // include/mbavo.h

namespace mbavo
{
    // keyframe_ops.hip: semi-dense keypoints of one pyramid level, device in / device out; count to the host
    int detect_semidense(Engine &eng, const unsigned char *d_img, int H, int W, int level, int im_H0, int im_W0, int cell_H,
                         int cell_W, float thr, const float *d_depth_z, double *d_kp_xy, double *d_kp_z, int cap, int *h_count);

    // one grid cell's strongest pixel (k_detect_cells); keep = a pixel above the threshold exists (and, when the kernel is
    // given the depth map, its depth is valid)
    struct CellPick
    {
        int x, y, keep;
        float z;
    };
    // grid selection only, nothing read back: the picks of the level's cells (row-major cells) -> d_picks, *num_cells of
    // them.  The front end's keyframe path: depth test + ordered compaction of the few hundred picks happen on the host, so
    // the depth map is never uploaded and a keyframe costs ONE stream synchronisation.
    int detect_cells_enqueue(Engine &eng, const unsigned char *d_img, int H, int W, int level, int im_H0, int im_W0, int cell_H,
                             int cell_W, float thr, CellPick *d_picks, int *num_cells);    // A keyframe's (or a frame's) levels in one launch each: the pyramid below d_levels[0] (three levels per launch), and with
    // keyframe_levels_enqueue also every level's gradient image and -- d_picks non-null -- every level's grid selection (picks in
    // level order, cells_per_level[l] of them).  Same integer / fp32 operations as the per-level kernels.
    int pyramid_enqueue(Engine &eng, unsigned char *const *d_levels, int H0, int W0, int L, hipStream_t on = nullptr);
    int keyframe_levels_enqueue(Engine &eng, unsigned char *const *d_levels, float *const *d_grads, int H0, int W0, int L, int cell_H, int cell_W,
                                float thr, CellPick *d_picks, int *cells_per_level, hipStream_t on = nullptr);
}

namespace SLAM
{
    namespace Core
    {
        class Transformation
        { // translation (3) then unit quaternion x,y,z,w (Transformation.h: mInternalRepresentation[7])
        public:
            Transformation();
            Transformation(const double q_xyzw[4], const double t[3]); // normalises q (Transformation.cpp:39-45)
            Transformation inverse() const;
            Transformation operator*(const Transformation &T) const;
            void apply(const double P[3], double out[3]) const; // operator*(Vector3d)
            const double *getData() const { return d; }
            const double *getRotationData() const { return d + 3; }
            const double *getTranslationData() const { return d; }
            static Transformation exp(const double tangent[6]);         // [upsilon, omega]
            static void log(const Transformation &T, double tangent[6]);
            static Transformation fromData(const double t_then_q[7]) // the stored bits, not re-normalised (state restore)
            {
                Transformation T;
                for (int i = 0; i < 7; ++i) T.d[i] = t_then_q[i];
                return T;
            }

        private:
            double d[7];
        };
    } // namespace Core

    namespace VO
    {
        struct BlurAwareDirectTrackerOptions
        {
            double intrinsics[4];
            int im_size_HW[2];
            int num_pyramid_levels;
            int num_virtual_poses_per_frame[8];
            int patch_size[8];
            const int *local_patch_pattern_xy[8];
            double huber_k;
            int max_consecutive_nonmonotonic_steps = 5;
            int max_num_iterations = 50;
            double min_step_quality = 0.5;
            double min_abs_cost_decrease = 0.001;
            int solver_type = 0; // "SVD_JACOBI" (0) / "LDLT" (1)
            int spline_deg_k = 2;
            double dt_frame, dt_ctrl_knot;
            double max_chi_square_error;
            double keyframe_max_flow_mag0, keyframe_max_flow_mag1, keyframe_max_flow_mag2, keyframe_max_blur_kernel_mag;
            // FeatureDetectorOptions as tmpProcessKeyframe sets them (blur_aware_direct_tracker.cpp:353-358)
            float score_threshold = 25.f;
            int grid_selection_cell_H = 30, grid_selection_cell_W = 30;
            // scheduling / solver form (include/mbavo.h: the ABI 3 tail of mbavo_vo_options; zero = default)
            double fast_solve_ratio = 0.0;
            int speculate = 0, persist_levels = 0, keyframe_levels_at_once = 0, speculate_keyframe = 0, ride_along = 0, resum = 0;
        };

        struct FrameView
        { // what trackFrame reads from a Core::Frame: the level-0 image (host memory) and its timing
            const unsigned char *image;
            double capture_time, exposure_time;
        };

        struct TrackInfo
        {
            int is_keyframe, num_keypoints0, num_trace, start_idx;
            double avg_flow, avg_kernel, final_cost;
        };

        struct TrackerState
        { // what trackFrame carries from frame to frame besides the keyframe's own data (checkpoint / resume of a tracker)
            double t0, dt;
            int N, is_first;
            double knots_t[3 * 16], knots_R[4 * 16];
            double T_keyframe[7], T_prev_b2w[7], velocity[6], prev_timestamp;
        };

        class BlurAwareDirectTracker
        {
        public:
            BlurAwareDirectTracker(mbavo::Engine &engine, const BlurAwareDirectTrackerOptions &options);
            ~BlurAwareDirectTracker();
            BlurAwareDirectTracker(const BlurAwareDirectTracker &) = delete;
            BlurAwareDirectTracker &operator=(const BlurAwareDirectTracker &) = delete;

            // returns 0 or an error code; *T_out = pose of the blurred frame in the world (first keyframe) frame
            int trackFrame(const FrameView &sharp_frame, const FrameView &blur_frame, const float *depth_z,
                           Core::Transformation *T_out, TrackInfo *info = nullptr);
            bool isKeyframe(double *avg_flow = nullptr, double *avg_kernel = nullptr) const;
            Core::SplineSE3 *getSplineTrajectory() { return &mSpline; }
            BlurAwareDirectTrackerOptions &getOptions() { return mOptions; }
            double getFinalEnergy() const { return mEvaluationPointCost; }
            int numKeypoints(int level) const { return mNumKeypoints[level]; }
            const double *deviceKeypointsXY(int level) const { return mKpXY[level]; }
            const double *deviceKeypointsZ(int level) const { return mKpZ[level]; }
            int status() const { return mStatus; } // allocation status of the constructor
            // LM records of the last trackFrame's optimizeTrajectory (the reference logs them; the long-horizon parity runs compare them)
            int lastTrace(mbavo_trace_rec *out, int cap) const;
            // checkpoint / resume: the inter-frame state, and the keyframe re-made from its sharp frame + depth map
            void getState(TrackerState &s) const;
            int setState(const TrackerState &s);
            int setKeyframe(const FrameView &keyframe, const float *depth_z) { return mStatus ? mStatus : tmpProcessKeyframe(keyframe, depth_z); }

        private:
            int tmpProcessKeyframe(const FrameView &keyframe, const float *depth_z);
            // Keyframe pre-processing AHEAD of the decision (round 5): when the constant-velocity prediction already says "keyframe",
            // the sharp frame's upload, pyramid, gradient images and grid selection are enqueued on a second stream into the SPARE
            // keyframe set while the LM loop runs (its evaluations are latency-bound and leave most of the GPU idle); if the
            // decision after the optimisation is "keyframe", what is left is the depth test of the picks on the host and the
            // keypoint upload, and the sets swap.  Same kernels on the same inputs: identical keypoints, bit for bit.
            int speculateKeyframe(const FrameView &keyframe);
            int finishKeyframe(const float *depth_z, bool spare_set);
            int ensureGridBuffers();
            int uploadCurrentFrame(const FrameView &frame);
            int optimizeTrajectory(int *num_trace, int *start_idx);

            mbavo::Engine &mEngine;
            BlurAwareDirectTrackerOptions mOptions;
            Core::SplineSE3 mSpline;
            Core::Transformation mTKeyframe, mTprevB2W;
            double mNeighFrameVelocity[6], mSplineVelocity[6];
            double mPrevTimestamp, mEvaluationPointCost;
            bool mIsFirstFrame;
            double mCurCap, mCurExp;
            int mStatus;
            mbavo_trace_rec *mTrace; // kTraceCap records
            int mNumTrace;
            static constexpr int kTraceCap = 512;

            // device-resident pyramids (keyframe: image + gradient; current frame: image) and keypoints
            unsigned char *mRef[8], *mCur[8];
            float *mGrad[8], *mDepth;
            double *mKpArena; // every level's keypoints [xy (2 cap) | z (cap)] back to back, in the layout of the pinned staging: one upload per keyframe
            const unsigned char **mCurPtr[8]; // device array of 1 device pointer per level
            double *mKpXY[8], *mKpZ[8];
            int *mPattern[8];
            int mKpCap[8], mNumKeypoints[8];
            std::vector<double> mHostKpXY0, mHostKpZ0; // level-0 keypoints for the keyframe test
            // grid selection: the cells' picks of all levels (device + pinned host copy) and the pinned staging of the
            // compacted keypoints [level][xy | z]
            // the spare keyframe set (speculateKeyframe): swapped with the active pointers above when its keyframe is adopted
            unsigned char *mRef2[8];
            float *mGrad2[8];
            double *mKpArena2, *mKpXY2[8], *mKpZ2[8];
            hipStream_t mKfStream;
            bool mKfInFlight;
            mbavo::CellPick *mPicksDev, *mPicksHost;
            double *mKpStage;
            size_t mPickOff[9], mStageOff[9];
        };
    } // namespace VO
} // namespace SLAM

#endif
