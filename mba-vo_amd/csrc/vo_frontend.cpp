// vo_frontend.cpp -- BlurAwareDirectTracker::trackFrame and its helpers on device-resident pyramids
// (ba_tracker/blur_aware_direct_tracker.cpp:14-415); see vo_frontend.h.
#include "vo_frontend.h"
#include "timing.h"
#include "../../include/mbavo.h"
#include "se3_math.h"
#include "tracker.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>

namespace SLAM
{
    namespace Core
    {
        using mbavo::Quat;

        static void normalized(const double q[4], double o[4])
        { // Eigen normalized(): q / sqrt(squaredNorm) when the norm is positive
            const double z = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
            if (z > 0)
            {
                const double n = std::sqrt(z);
                for (int i = 0; i < 4; ++i) o[i] = q[i] / n;
            }
            else
                for (int i = 0; i < 4; ++i) o[i] = q[i];
        }

        Transformation::Transformation()
        {
            for (int i = 0; i < 6; ++i) d[i] = 0;
            d[6] = 1;
        }

        Transformation::Transformation(const double q[4], const double t[3])
        {
            normalized(q, d + 3);
            d[0] = t[0]; d[1] = t[1]; d[2] = t[2];
        }

        Transformation Transformation::inverse() const
        { // Transformation.cpp:83-90
            const double qc[4] = {-d[3], -d[4], -d[5], d[6]}, nt[3] = {-d[0], -d[1], -d[2]};
            double ti[3];
            mbavo::qrotate(mbavo::load_quat(qc), nt, ti);
            return Transformation(qc, ti);
        }

        Transformation Transformation::operator*(const Transformation &T) const
        { // Transformation.cpp:109-119
            const Quat a = mbavo::load_quat(d + 3);
            const Quat q = mbavo::qmul(a, mbavo::load_quat(T.d + 3));
            double t[3];
            mbavo::qrotate(a, T.d, t);
            t[0] += d[0]; t[1] += d[1]; t[2] += d[2];
            const double qq[4] = {q.x, q.y, q.z, q.w};
            return Transformation(qq, t);
        }

        void Transformation::apply(const double P[3], double out[3]) const
        {
            mbavo::qrotate(mbavo::load_quat(d + 3), P, out);
            out[0] += d[0]; out[1] += d[1]; out[2] += d[2];
        }

        static void hat_and_square(const double w[3], double O[9], double O2[9])
        {
            const double h[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
            for (int i = 0; i < 9; ++i) O[i] = h[i];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c)
                {
                    double a = 0;
                    for (int j = 0; j < 3; ++j) a += h[r * 3 + j] * h[j * 3 + c];
                    O2[r * 3 + c] = a;
                }
        }

        Transformation Transformation::exp(const double a[6])
        { // Transformation.cpp:171-177 -> Sophus::SE3d::exp: t = V(omega) * upsilon
            const double *om = a + 3;
            const Quat q = mbavo::so3_exp(om);
            const double theta = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
            double O[9], O2[9], V[9];
            hat_and_square(om, O, O2);
            if (theta < 1e-10)
            {
                const double x = q.x, y = q.y, z = q.z, w = q.w; // V = so3.matrix()
                const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
                for (int i = 0; i < 9; ++i) V[i] = R[i];
            }
            else
            {
                const double th2 = theta * theta;
                const double c1 = (1 - std::cos(theta)) / th2, c2 = (theta - std::sin(theta)) / (th2 * theta);
                for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * O[i] + c2 * O2[i];
            }
            double t[3];
            for (int r = 0; r < 3; ++r) t[r] = V[r * 3] * a[0] + V[r * 3 + 1] * a[1] + V[r * 3 + 2] * a[2];
            const double qq[4] = {q.x, q.y, q.z, q.w};
            return Transformation(qq, t);
        }

        void Transformation::log(const Transformation &T, double out[6])
        { // Transformation.cpp:164-169 -> Sophus::SE3d::log
            const double *q = T.d + 3, *t = T.d;
            const double sn = q[0] * q[0] + q[1] * q[1] + q[2] * q[2], w = q[3], n = std::sqrt(sn);
            double two_atan;
            if (sn < 1e-10 * 1e-10)
                two_atan = 2.0 / w - 2.0 * sn / (w * (w * w));
            else if (std::fabs(w) < 1e-10)
                two_atan = (w > 0 ? M_PI : -M_PI) / n;
            else
                two_atan = 2.0 * std::atan(n / w) / n;
            const double theta = two_atan * n;
            const double om[3] = {two_atan * q[0], two_atan * q[1], two_atan * q[2]};
            double O[9], O2[9];
            hat_and_square(om, O, O2);
            double c2;
            if (std::fabs(theta) < 1e-10)
                c2 = 1.0 / 12.0;
            else
            {
                const double h = 0.5 * theta;
                c2 = (1 - theta * std::cos(h) / (2 * std::sin(h))) / (theta * theta);
            }
            for (int r = 0; r < 3; ++r)
            {
                double acc = 0;
                for (int c = 0; c < 3; ++c) acc += (((r == c) ? 1.0 : 0.0) - 0.5 * O[r * 3 + c] + c2 * O2[r * 3 + c]) * t[c];
                out[r] = acc;
            }
            out[3] = om[0]; out[4] = om[1]; out[5] = om[2];
        }
    } // namespace Core

    namespace VO
    {
#define VO_HIP(expr)                                                                          \
    do                                                                                        \
    {                                                                                         \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
        {                                                                                     \
            fprintf(stderr, "mbavo vo: %s failed: %s\n", #expr, hipGetErrorString(e_));       \
            return (int)e_;                                                                   \
        }                                                                                     \
    } while (0)

        static int grid_cells(int H0, int W0, int lv, int cell_H, int cell_W)
        { // FeatureDetectorBase.cpp:56-64
            const int sf = (int)std::pow(2, lv);
            const int ch = (int)(cell_H / std::pow(1.414, lv)), cw = (int)(cell_W / std::pow(1.414, lv));
            if (ch < 1 || cw < 1) return 0;
            return ((H0 / sf) / ch + 1) * ((W0 / sf) / cw + 1);
        }

        BlurAwareDirectTracker::BlurAwareDirectTracker(mbavo::Engine &engine, const BlurAwareDirectTrackerOptions &options)
            : mEngine(engine), mOptions(options), mPrevTimestamp(0), mEvaluationPointCost(0), mIsFirstFrame(true),
              mCurCap(0), mCurExp(0), mStatus(0), mTrace(new mbavo_trace_rec[kTraceCap]), mNumTrace(0), mDepth(nullptr), mKpArena(nullptr), mKpArena2(nullptr), mKfStream(nullptr), mKfInFlight(false), mPicksDev(nullptr), mPicksHost(nullptr), mKpStage(nullptr)
        { // blur_aware_direct_tracker.cpp:14-34; the shared storages are the engine's
            for (int i = 0; i < 6; ++i) mNeighFrameVelocity[i] = mSplineVelocity[i] = 0;
            for (int l = 0; l < 8; ++l)
            {
                mRef[l] = mCur[l] = nullptr; mGrad[l] = nullptr; mCurPtr[l] = nullptr; mKpXY[l] = mKpZ[l] = nullptr;
                mRef2[l] = nullptr; mGrad2[l] = nullptr; mKpXY2[l] = mKpZ2[l] = nullptr;
                mPattern[l] = nullptr; mKpCap[l] = mNumKeypoints[l] = 0;
            }
            mSpline.setSamplingFreq(mOptions.dt_ctrl_knot);
            const int H = mOptions.im_size_HW[0], W = mOptions.im_size_HW[1], L = mOptions.num_pyramid_levels;
            if (L < 1 || L > 8 || H < 2 || W < 2) { mStatus = MBAVO_E_ARG; return; }
            auto alloc = [&](void **p, size_t bytes) { if (mStatus == 0) { hipError_t e = hipMalloc(p, bytes ? bytes : 1); if (e != hipSuccess) mStatus = (int)e; } };
            const bool grid = mOptions.grid_selection_cell_H > 0 && mOptions.grid_selection_cell_W > 0;
            if (!grid) alloc((void **)&mDepth, sizeof(float) * (size_t)H * W); // (grid selection tests the depth on the host)
            mPickOff[0] = mStageOff[0] = 0;
            for (int l = 0; l < L && mStatus == 0; ++l)
            {
                const size_t n = (size_t)(H >> l) * (W >> l);
                mKpCap[l] = grid ? grid_cells(H, W, l, mOptions.grid_selection_cell_H, mOptions.grid_selection_cell_W) : (int)n;
                if (mKpCap[l] < 1) { mStatus = MBAVO_E_ARG; return; }
                alloc((void **)&mRef[l], n); alloc((void **)&mCur[l], n); alloc((void **)&mGrad[l], n * 2 * sizeof(float));
                alloc((void **)&mCurPtr[l], sizeof(void *));
                const int P = mOptions.patch_size[l];
                if (P < 1 || !mOptions.local_patch_pattern_xy[l]) { mStatus = MBAVO_E_ARG; return; }
                alloc((void **)&mPattern[l], sizeof(int) * 2 * P);
                mPickOff[l + 1] = mPickOff[l] + (size_t)mKpCap[l];
                mStageOff[l + 1] = mStageOff[l] + 3 * (size_t)mKpCap[l];
                if (mStatus == 0)
                {
                    hipError_t e = hipMemcpy(mPattern[l], mOptions.local_patch_pattern_xy[l], sizeof(int) * 2 * P, hipMemcpyHostToDevice);
                    if (e == hipSuccess) e = hipMemcpy(mCurPtr[l], &mCur[l], sizeof(void *), hipMemcpyHostToDevice);
                    if (e != hipSuccess) mStatus = (int)e;
                }
            }
            alloc((void **)&mKpArena, sizeof(double) * mStageOff[L]);
            for (int l = 0; l < L && mStatus == 0; ++l)
            {
                mKpXY[l] = mKpArena + mStageOff[l];
                mKpZ[l] = mKpXY[l] + 2 * (size_t)mKpCap[l];
            }
            // the spare keyframe set + its stream (grid selection only: the other detector path reads the depth map on the device)
            if (grid && mStatus == 0 && mbavo::opt_flag(mOptions.speculate_keyframe, mbavo::read_env_overrides().kf_speculate, true))
            {
                for (int l = 0; l < L && mStatus == 0; ++l)
                {
                    const size_t n = (size_t)(H >> l) * (W >> l);
                    alloc((void **)&mRef2[l], n); alloc((void **)&mGrad2[l], n * 2 * sizeof(float));
                }
                alloc((void **)&mKpArena2, sizeof(double) * mStageOff[L]);
                for (int l = 0; l < L && mStatus == 0; ++l)
                {
                    mKpXY2[l] = mKpArena2 + mStageOff[l];
                    mKpZ2[l] = mKpXY2[l] + 2 * (size_t)mKpCap[l];
                }
                if (mStatus == 0 && hipStreamCreateWithFlags(&mKfStream, hipStreamNonBlocking) != hipSuccess) { mKfStream = nullptr; (void)hipGetLastError(); }
            }
        }

        int BlurAwareDirectTracker::ensureGridBuffers()
        { // first keyframe with grid selection: picks of every cell of every level, device + pinned; pinned keypoint staging
            if (mPicksDev) return 0;
            const int L = mOptions.num_pyramid_levels;
            VO_HIP(hipMalloc((void **)&mPicksDev, sizeof(mbavo::CellPick) * mPickOff[L]));
            VO_HIP(hipHostMalloc((void **)&mPicksHost, sizeof(mbavo::CellPick) * mPickOff[L], hipHostMallocDefault));
            VO_HIP(hipHostMalloc((void **)&mKpStage, sizeof(double) * mStageOff[L], hipHostMallocDefault));
            return 0;
        }

        int BlurAwareDirectTracker::lastTrace(mbavo_trace_rec *out, int cap) const
        {
            const int n = mNumTrace < cap ? mNumTrace : cap;
            if (n > 0) memcpy(out, mTrace, sizeof(mbavo_trace_rec) * n);
            return n;
        }

        void BlurAwareDirectTracker::getState(TrackerState &s) const
        {
            memset(&s, 0, sizeof(s));
            s.t0 = mSpline.getStartTime(); s.dt = mSpline.getSamplingFreq();
            s.N = (int)mSpline.get_num_knots(); s.is_first = mIsFirstFrame ? 1 : 0;
            if (s.N > 16) s.N = 16;
            if (s.N > 0)
            {
                memcpy(s.knots_t, mSpline.get_knot_data_t(), sizeof(double) * 3 * s.N);
                memcpy(s.knots_R, mSpline.get_knot_data_R(), sizeof(double) * 4 * s.N);
            }
            memcpy(s.T_keyframe, mTKeyframe.getData(), sizeof(double) * 7);
            memcpy(s.T_prev_b2w, mTprevB2W.getData(), sizeof(double) * 7);
            for (int i = 0; i < 6; ++i) s.velocity[i] = mNeighFrameVelocity[i];
            s.prev_timestamp = mPrevTimestamp;
        }

        int BlurAwareDirectTracker::setState(const TrackerState &s)
        {
            if (s.N < 0 || s.N > 16) return MBAVO_E_ARG;
            mSpline.Clear();
            mSpline.setStartTime(s.t0); mSpline.setSamplingFreq(s.dt); mSpline.setSplineDegK(mOptions.spline_deg_k);
            for (int i = 0; i < s.N; ++i) mSpline.InsertControlKnot(s.knots_R + 4 * i, s.knots_t + 3 * i);
            mTKeyframe = Core::Transformation::fromData(s.T_keyframe);
            mTprevB2W = Core::Transformation::fromData(s.T_prev_b2w);
            for (int i = 0; i < 6; ++i) mNeighFrameVelocity[i] = mSplineVelocity[i] = s.velocity[i];
            mPrevTimestamp = s.prev_timestamp;
            mIsFirstFrame = s.is_first != 0;
            return 0;
        }

        BlurAwareDirectTracker::~BlurAwareDirectTracker()
        {
            delete[] mTrace;
            if (mKfStream) { (void)hipStreamSynchronize(mKfStream); (void)hipStreamDestroy(mKfStream); }
            (void)hipFree(mDepth); (void)hipFree(mKpArena); (void)hipFree(mKpArena2);
            for (int l = 0; l < 8; ++l) { (void)hipFree(mRef2[l]); (void)hipFree(mGrad2[l]); }
            (void)hipFree(mPicksDev); (void)hipHostFree(mPicksHost); (void)hipHostFree(mKpStage);
            for (int l = 0; l < 8; ++l)
            {
                (void)hipFree(mRef[l]); (void)hipFree(mCur[l]); (void)hipFree(mGrad[l]); (void)hipFree(mCurPtr[l]);
                (void)hipFree(mPattern[l]);
            }
        }

        int BlurAwareDirectTracker::finishKeyframe(const float *depth_z, bool spare_set)
        { // the host half of the grid-selection path: depth test (:398-404) and ordered compaction of the cells' picks (already in
          // mPicksHost), ONE keypoint upload; spare_set: the picks belong to the spare keyframe set, which becomes the active one
            const int W = mOptions.im_size_HW[1], L = mOptions.num_pyramid_levels;
            hipStream_t st = mEngine.stream();
            for (int l = 0; l < L; ++l)
            {
                const double scale = std::pow(2, l);
                double *xy = mKpStage + mStageOff[l], *z = xy + 2 * (size_t)mKpCap[l];
                int n = 0;
                for (int c = 0; c < mKpCap[l]; ++c)
                {
                    const mbavo::CellPick &p = mPicksHost[mPickOff[l] + c];
                    if (!p.keep) continue;
                    const int x0 = (int)((float)p.x * scale + 0.5), y0 = (int)((float)p.y * scale + 0.5); // :398-400
                    const float zz = depth_z[(size_t)y0 * W + x0];
                    if ((double)zz < 1e-2) continue;
                    xy[2 * n] = (double)p.x; xy[2 * n + 1] = (double)p.y; z[n] = (double)zz;
                    ++n;
                }
                mNumKeypoints[l] = n;
            }
            if (spare_set)
            { // adopt: the spare set's images are complete (its stream was drained by the caller); swap the pointers
                for (int l = 0; l < 8; ++l) { std::swap(mRef[l], mRef2[l]); std::swap(mGrad[l], mGrad2[l]); std::swap(mKpXY[l], mKpXY2[l]); std::swap(mKpZ[l], mKpZ2[l]); }
                std::swap(mKpArena, mKpArena2);
            }
            // ONE upload for all levels: the device arena has the staging's layout (eight small copies cost ~4 us of host time each)
            VO_HIP(hipMemcpyAsync(mKpArena, mKpStage, sizeof(double) * mStageOff[L], hipMemcpyHostToDevice, st));
            mHostKpXY0.assign(mKpStage, mKpStage + 2 * (size_t)mNumKeypoints[0]);
            mHostKpZ0.assign(mKpStage + 2 * (size_t)mKpCap[0], mKpStage + 2 * (size_t)mKpCap[0] + mNumKeypoints[0]);
            return 0;
        }

        int BlurAwareDirectTracker::speculateKeyframe(const FrameView &kf)
        { // upload + pyramid + gradient images + grid selection of `kf` into the SPARE set, on the spare stream; nothing waits
            if (!mKfStream || !mRef2[0]) return 1;
            mbavo::PhaseScope ps(mbavo::PhaseTimers::kKeyframe);
            const int H = mOptions.im_size_HW[0], W = mOptions.im_size_HW[1], L = mOptions.num_pyramid_levels;
            int rc = ensureGridBuffers();
            if (rc != 0) return rc;
            VO_HIP(hipMemcpyAsync(mRef2[0], kf.image, (size_t)H * W, hipMemcpyHostToDevice, mKfStream));
            int ncell[8] = {};
            rc = mbavo::keyframe_levels_enqueue(mEngine, mRef2, mGrad2, H, W, L, mOptions.grid_selection_cell_H, mOptions.grid_selection_cell_W,
                                                mOptions.score_threshold, mPicksDev, ncell, mKfStream);
            if (rc != 0) return rc;
            for (int l = 0; l < L; ++l)
                if (ncell[l] != mKpCap[l]) return MBAVO_E_ARG;
            VO_HIP(hipMemcpyAsync(mPicksHost, mPicksDev, sizeof(mbavo::CellPick) * mPickOff[L], hipMemcpyDeviceToHost, mKfStream));
            mKfInFlight = true;
            return 0;
        }

        int BlurAwareDirectTracker::tmpProcessKeyframe(const FrameView &kf, const float *depth_z)
        { // blur_aware_direct_tracker.cpp:342-415: pyramid, gradients, semi-dense keypoints with depth -- all on device
            mbavo::PhaseScope ps(mbavo::PhaseTimers::kKeyframe);
            const int H = mOptions.im_size_HW[0], W = mOptions.im_size_HW[1], L = mOptions.num_pyramid_levels;
            hipStream_t st = mEngine.stream();
            if (mKfInFlight) { VO_HIP(hipStreamSynchronize(mKfStream)); mKfInFlight = false; } // (a discarded speculation still owns the pick buffers)
            VO_HIP(hipMemcpyAsync(mRef[0], kf.image, (size_t)H * W, hipMemcpyHostToDevice, st));
            if (mOptions.grid_selection_cell_H > 0 && mOptions.grid_selection_cell_W > 0)
            { // grid selection: every level's kernels back to back, ONE read-back (the cells' picks) and ONE synchronisation;
              // depth test (:398-404) and ordered compaction of the few hundred picks on the host, keypoints uploaded from
              // pinned staging -- the 4 H x W bytes of depth never cross the bus
                int rc = ensureGridBuffers();
                if (rc != 0) return rc;
                const bool levels_at_once = mbavo::opt_flag(mOptions.keyframe_levels_at_once, mbavo::read_env_overrides().kf_multi, true);
                if (levels_at_once)
                { // pyramid, gradient images and grid selection of ALL levels: three launches (keyframe_levels_at_once = -1: three per level)
                    int ncell[8] = {};
                    rc = mbavo::keyframe_levels_enqueue(mEngine, mRef, mGrad, H, W, L, mOptions.grid_selection_cell_H, mOptions.grid_selection_cell_W,
                                                        mOptions.score_threshold, mPicksDev, ncell);
                    if (rc != 0) return rc;
                    for (int l = 0; l < L; ++l)
                        if (ncell[l] != mKpCap[l]) return MBAVO_E_ARG;
                }
                else
                for (int l = 0; l < L; ++l)
                {
                    const int Hl = H >> l, Wl = W >> l;
                    int nc = 0;
                    if (l > 0 && (rc = mbavo_pyramid_down_u8(mRef[l - 1], H >> (l - 1), W >> (l - 1), mRef[l], st)) != 0) return rc;
                    if ((rc = mbavo_image_gradients_u8(mRef[l], Hl, Wl, mGrad[l], st)) != 0) return rc;
                    rc = mbavo::detect_cells_enqueue(mEngine, mRef[l], Hl, Wl, l, H, W, mOptions.grid_selection_cell_H,
                                                     mOptions.grid_selection_cell_W, mOptions.score_threshold, mPicksDev + mPickOff[l], &nc);
                    if (rc != 0) return rc;
                    if (nc != mKpCap[l]) return MBAVO_E_ARG;
                }
                VO_HIP(hipMemcpyAsync(mPicksHost, mPicksDev, sizeof(mbavo::CellPick) * mPickOff[L], hipMemcpyDeviceToHost, st));
                VO_HIP(hipStreamSynchronize(st));
                return finishKeyframe(depth_z, false);
            }
            VO_HIP(hipMemcpyAsync(mDepth, depth_z, sizeof(float) * (size_t)H * W, hipMemcpyHostToDevice, st));
            for (int l = 0; l < L; ++l)
            {
                const int Hl = H >> l, Wl = W >> l;
                int rc;
                if (l > 0 && (rc = mbavo_pyramid_down_u8(mRef[l - 1], H >> (l - 1), W >> (l - 1), mRef[l], st)) != 0) return rc;
                if ((rc = mbavo_image_gradients_u8(mRef[l], Hl, Wl, mGrad[l], st)) != 0) return rc;
                rc = mbavo::detect_semidense(mEngine, mRef[l], Hl, Wl, l, H, W, mOptions.grid_selection_cell_H,
                                             mOptions.grid_selection_cell_W, mOptions.score_threshold, mDepth, mKpXY[l], mKpZ[l],
                                             mKpCap[l], &mNumKeypoints[l]);
                if (rc != 0) return rc;
                if (mNumKeypoints[l] > mKpCap[l]) mNumKeypoints[l] = mKpCap[l];
            }
            mHostKpXY0.resize(2 * (size_t)mNumKeypoints[0]);
            mHostKpZ0.resize(mNumKeypoints[0]);
            if (mNumKeypoints[0] > 0)
            {
                VO_HIP(hipMemcpyAsync(mHostKpXY0.data(), mKpXY[0], sizeof(double) * 2 * mNumKeypoints[0], hipMemcpyDeviceToHost, st));
                VO_HIP(hipMemcpyAsync(mHostKpZ0.data(), mKpZ[0], sizeof(double) * mNumKeypoints[0], hipMemcpyDeviceToHost, st));
            }
            VO_HIP(hipStreamSynchronize(st));
            return 0;
        }

        int BlurAwareDirectTracker::uploadCurrentFrame(const FrameView &f)
        { // :112-117: pyramid of the blurred frame, on device
            mbavo::PhaseScope ps(mbavo::PhaseTimers::kUpload);
            const int H = mOptions.im_size_HW[0], W = mOptions.im_size_HW[1];
            hipStream_t st = mEngine.stream();
            VO_HIP(hipMemcpyAsync(mCur[0], f.image, (size_t)H * W, hipMemcpyHostToDevice, st));
            const bool levels_at_once = mbavo::opt_flag(mOptions.keyframe_levels_at_once, mbavo::read_env_overrides().kf_multi, true);
            if (levels_at_once)
            {
                const int rc = mbavo::pyramid_enqueue(mEngine, mCur, H, W, mOptions.num_pyramid_levels);
                if (rc != 0) return rc;
            }
            else
            for (int l = 1; l < mOptions.num_pyramid_levels; ++l)
            {
                const int rc = mbavo_pyramid_down_u8(mCur[l - 1], H >> (l - 1), W >> (l - 1), mCur[l], st);
                if (rc != 0) return rc;
            }
            mCurCap = f.capture_time;
            mCurExp = f.exposure_time;
            return 0;
        }

        int BlurAwareDirectTracker::optimizeTrajectory(int *num_trace, int *start_idx)
        { // :544-588 -> tracker.cpp
            mbavo_level lv[8];
            const int L = mOptions.num_pyramid_levels;
            for (int l = 0; l < L; ++l)
            {
                lv[l].H = mOptions.im_size_HW[0] >> l; lv[l].W = mOptions.im_size_HW[1] >> l;
                lv[l].K = mNumKeypoints[l]; lv[l].P = mOptions.patch_size[l]; lv[l].S = mOptions.num_virtual_poses_per_frame[l];
                lv[l].d_ref_img = mRef[l]; lv[l].d_ref_dIxy = mGrad[l]; lv[l].d_cur_imgs = (const unsigned char *const *)mCurPtr[l];
                lv[l].d_kp_xy = mKpXY[l]; lv[l].d_kp_z = mKpZ[l]; lv[l].d_pattern = mPattern[l];
            }
            mbavo_track_opts o;
            memset(&o, 0, sizeof(o));
            o.num_levels = L; o.spline_deg_k = mOptions.spline_deg_k; o.max_num_iterations = mOptions.max_num_iterations;
            o.max_consecutive_nonmonotonic_steps = mOptions.max_consecutive_nonmonotonic_steps; o.solver_type = mOptions.solver_type;
            for (int i = 0; i < 4; ++i) o.intrinsics[i] = mOptions.intrinsics[i];
            o.huber_k = mOptions.huber_k; o.min_step_quality = mOptions.min_step_quality;
            o.min_abs_cost_decrease = mOptions.min_abs_cost_decrease; o.max_chi_square_error = mOptions.max_chi_square_error;
            o.fast_solve_ratio = mOptions.fast_solve_ratio; o.speculate = mOptions.speculate; o.persist_levels = mOptions.persist_levels;
            o.ride_along = mOptions.ride_along; o.resum = mOptions.resum;
            const int n = mbavo::optimize_trajectory(mEngine, o, lv, 1, &mCurCap, &mCurExp, mSpline.getStartTime(),
                                                     mSpline.getSamplingFreq(), mSpline.get_knot_data_t(), mSpline.get_knot_data_R(),
                                                     (int)mSpline.get_num_knots(), start_idx, &mEvaluationPointCost, mTrace, kTraceCap);
            mNumTrace = n < 0 ? 0 : (n < kTraceCap ? n : kTraceCap);
            if (n < 0) return n;
            if (num_trace) *num_trace = n;
            return 0;
        }

        bool BlurAwareDirectTracker::isKeyframe(double *avg_flow_out, double *avg_kernel_out) const
        { // :205-262.  A point projecting behind the camera leaves the reference's output uninitialised; it stays (0,0) here.
            const double *K = mOptions.intrinsics;
            const int n = mNumKeypoints[0];
            const double times[3] = {mCurCap, mCurCap - 0.5 * mCurExp, mCurCap + 0.5 * mCurExp};
            Core::Transformation Tinv[3];
            for (int j = 0; j < 3; ++j)
            {
                double q[4], t[3];
                if (!mSpline.GetPose(times[j], q, t)) return false;
                Tinv[j] = Core::Transformation(q, t).inverse();
            }
            double flow = 0, kern = 0;
            for (int i = 0; i < n; ++i)
            {
                const double x = mHostKpXY0[2 * i], y = mHostKpXY0[2 * i + 1], z = mHostKpZ0[i];
                const double P[3] = {(x - K[2]) / K[0] * z, (y - K[3]) / K[1] * z, z}; // CameraPinhole.cpp:79-94
                double Pc[3], p[3][2] = {{0, 0}, {0, 0}, {0, 0}};
                for (int j = 0; j < 3; ++j)
                {
                    Tinv[j].apply(P, Pc);
                    if (Pc[2] < 0) continue; // CameraPinhole.cpp:24-43
                    p[j][0] = K[0] * (Pc[0] / (Pc[2] + 1e-8)) + K[2];
                    p[j][1] = K[1] * (Pc[1] / (Pc[2] + 1e-8)) + K[3];
                }
                flow += (p[0][0] - x) * (p[0][0] - x) + (p[0][1] - y) * (p[0][1] - y);
                kern += (p[1][0] - p[2][0]) * (p[1][0] - p[2][0]) + (p[1][1] - p[2][1]) * (p[1][1] - p[2][1]);
            }
            const double avg_flow = sqrtf((float)(flow / n)), avg_kernel = sqrtf((float)(kern / n));
            if (avg_flow_out) *avg_flow_out = avg_flow;
            if (avg_kernel_out) *avg_kernel_out = avg_kernel;
            if (avg_flow > mOptions.keyframe_max_flow_mag0 && avg_kernel < mOptions.keyframe_max_blur_kernel_mag) return true;
            if (avg_flow > mOptions.keyframe_max_flow_mag1) return true;
            return false;
        }

        int BlurAwareDirectTracker::trackFrame(const FrameView &sharp, const FrameView &blur, const float *depth_z,
                                               Core::Transformation *T_out, TrackInfo *info)
        { // :88-203
            mbavo::PhaseScope ps_all(mbavo::PhaseTimers::kTrack);
            if (mStatus != 0) return mStatus;
            if (!T_out || !sharp.image || !depth_z) return MBAVO_E_ARG;
            if (info) memset(info, 0, sizeof(*info));
            int rc;
            if (mIsFirstFrame)
            {
                mIsFirstFrame = false;
                if ((rc = tmpProcessKeyframe(sharp, depth_z)) != 0) return rc;
                mPrevTimestamp = sharp.capture_time;
                if (mSpline.get_num_knots() == 0)
                { // :99-106: two identity control knots
                    mSpline.setSamplingFreq(mOptions.dt_frame);
                    mSpline.setSplineDegK(mOptions.spline_deg_k);
                    mSpline.setStartTime(sharp.capture_time);
                    const double qi[4] = {0, 0, 0, 1}, ti[3] = {0, 0, 0};
                    mSpline.InsertControlKnot(qi, ti);
                    mSpline.InsertControlKnot(qi, ti);
                }
                *T_out = mTKeyframe;
                if (info) { info->is_keyframe = 1; info->num_keypoints0 = mNumKeypoints[0]; }
                return 0;
            }
            if (!blur.image) return MBAVO_E_ARG;
            if ((rc = uploadCurrentFrame(blur)) != 0) return rc;

            // constant-velocity prediction from the two previous frames (:119-141)
            const double dt_frame = blur.capture_time - mPrevTimestamp;
            for (int i = 0; i < 6; ++i) { mNeighFrameVelocity[i] *= dt_frame; mSplineVelocity[i] = mNeighFrameVelocity[i]; }
            const Core::Transformation dTspline = Core::Transformation::exp(mSplineVelocity);
            mSpline.setStartTime(blur.capture_time - 0.5 * blur.exposure_time);
            mSpline.TransformByRight(dTspline.getRotationData(), dTspline.getTranslationData());

            // the keyframe test on the PREDICTED spline: if it already says "keyframe", the sharp frame's pre-processing starts now, on
            // the spare stream, and runs under the LM loop (a wrong prediction costs the enqueue and some idle GPU time, never a result)
            bool speculated = false;
            if (mKfStream && sharp.image && isKeyframe(nullptr, nullptr))
            {
                if (mKfInFlight) { VO_HIP(hipStreamSynchronize(mKfStream)); mKfInFlight = false; } // (a discarded earlier speculation still owns the pick buffers)
                const int sr = speculateKeyframe(sharp);
                if (sr < 0 || sr > 1) return sr;
                speculated = sr == 0;
            }

            int ntrace = 0, start = 0;
            if ((rc = optimizeTrajectory(&ntrace, &start)) != 0) return rc;

            double af = 0, ak = 0;
            const bool is_kf = isKeyframe(&af, &ak);

            double q[4], t[3];
            if (!mSpline.GetPose(blur.capture_time, q, t)) return MBAVO_E_RANGE;
            const Core::Transformation T_b2w(q, t);
            const Core::Transformation dTn = mTprevB2W.inverse() * T_b2w; // :150-155
            double lg[6];
            Core::Transformation::log(dTn, lg);
            for (int i = 0; i < 6; ++i) mNeighFrameVelocity[i] = lg[i] / dt_frame;
            mTprevB2W = T_b2w;

            if (is_kf)
            { // :176-188: the sharp companion becomes the keyframe, the spline is re-expressed relative to it
                if (speculated)
                { // the spare set holds this frame's pyramid, gradients and picks: drain its stream (long done), finish on the host
                    mbavo::PhaseScope ps_kf(mbavo::PhaseTimers::kKeyframe);
                    VO_HIP(hipStreamSynchronize(mKfStream));
                    mKfInFlight = false;
                    if ((rc = finishKeyframe(depth_z, true)) != 0) return rc;
                }
                else
                {
                    if (mKfInFlight) { VO_HIP(hipStreamSynchronize(mKfStream)); mKfInFlight = false; } // (its D2H copy targets mPicksHost)
                    if ((rc = tmpProcessKeyframe(sharp, depth_z)) != 0) return rc;
                }
                mSpline.GetPose(blur.capture_time, q, t);
                mTKeyframe = mTKeyframe * Core::Transformation(q, t);
                const double qi[4] = {0, 0, 0, 1}, ti[3] = {0, 0, 0};
                mSpline.TransformTo(blur.capture_time, qi, ti);
                mTprevB2W = Core::Transformation();
            }
            mPrevTimestamp = blur.capture_time;
            mSpline.GetPose(blur.capture_time, q, t);
            *T_out = mTKeyframe * Core::Transformation(q, t);
            if (info)
            {
                info->is_keyframe = is_kf ? 1 : 0; info->num_keypoints0 = mNumKeypoints[0]; info->num_trace = ntrace;
                info->start_idx = start; info->avg_flow = af; info->avg_kernel = ak; info->final_cost = mEvaluationPointCost;
            }
            return 0;
        }
    } // namespace VO
} // namespace SLAM
