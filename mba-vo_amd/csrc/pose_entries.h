// pose_entries.h -- the blur samples' pose entries (PoseEntry: pose, translation weights, R * A) from the control knots: device
// code shared by the evaluation kernels (engine.hip: k_pose_table, the prologues of the fused kernels) and the batched LM's solve
// kernel (lm_batch.hip, round 4: the workgroup that produces a candidate also produces its pose entries, once for both passes
// of the slot).  compute_virtual_camera_poses.cu:9-110, core/common/SplineFunctor.h.
#ifndef MBAVO_POSE_ENTRIES_H
#define MBAVO_POSE_ENTRIES_H

#include "engine.h"
#include "pixel_math.h"
#include "se3_math.h"

#ifndef MBAVO_PSTAMP
#define MBAVO_PSTAMP(i) do { } while (0)
#endif

namespace mbavo
{
    // ------------------------------------------------------------------ pose table
    // Latency-bound: one sample's pose and pose-to-knot Jacobians are a chain of fp64 log / atan / exp / sin / cos and
    // quaternion products, ~2 600 dependent instructions if one lane does it all.  Two stages per workgroup
    // (se3_math.h "two STAGES"): A -- one lane per (sample, segment) evaluates A_g = exp(c_g log(R_g^-1 R_g+1)) with both
    // Jacobians, segment g on wave g; B -- one WAVE per knot (compile-time knot index, so each knot only touches the
    // segments it differentiates), one LANE per (sample, column of that knot's 4x3 block), reading the segments from LDS.
    // The pose itself is written by knot 0 / column 0.  kPoseSPB samples per workgroup: 21 x 3 columns fill a wave.
    constexpr int kPoseSPB = 21;

    // The table holds R * A (pixel_math.h PoseEntry): this lane's column of A = d(body rotation) / d(knot rotation), turned
    // into the keyframe's axes by the sample's own rotation matrix -- once per (sample, column) here instead of a transposed
    // product per pixel-sample in the fused kernels.
    template <int KD>
    __device__ __forceinline__ void store_rotated_column(const Quat &q, const double av[3], int c, PoseEntry<KD> &pe)
    {
        const double qv[4] = {q.x, q.y, q.z, q.w};
        double R[9];
        rotation_entries(qv, R);
        for (int b = 0; b < 3; ++b)
        {
            double r = R[3 * b] * av[0];
            r += R[3 * b + 1] * av[1];
            r += R[3 * b + 2] * av[2];
            pe.A[b * 3 * KD + c] = r;
        }
    }

    template <int KD, int KNOT>
    __device__ __forceinline__ Quat pose_table_entry(const double *kR, double u, int col, const SplineSeg *sg, PoseEntry<KD> &pe)
    {
        JacC<1> blk;
        const Quat q = spline_rotation_knot_from_segs<KD, 1, KNOT>(kR, u, col, sg, blk);
        // tangent form of this column: A[a][3*knot + col] = 2 * L3(q)^T[a] . blk      (pixel_math.h)
        const double L3[4][3] = {{q.w, -q.z, q.y}, {q.z, q.w, -q.x}, {-q.y, q.x, q.w}, {-q.x, -q.y, -q.z}};
        const Quat &v = blk.c[0];
        double av[3];
        for (int a = 0; a < 3; ++a)
        {
            double r = L3[0][a] * v.x;
            r += L3[1][a] * v.y;
            r += L3[2][a] * v.z;
            r += L3[3][a] * v.w;
            av[a] = 2.0 * r;
        }
        store_rotated_column<KD>(q, av, 3 * KNOT + col, pe);
        return q;
    }

    // sample time, segment index (clamped into the knot range, reported through *oob) and normalised time of sample
    // `smp` of frame f (compute_virtual_camera_poses.cu:33: S == 1 samples the START of the exposure)
    template <int KD>
    __device__ __forceinline__ void pose_sample_segment(const ProblemDesc &d, int f, int smp, int &idx, double &u, bool &oob)
    {
        const double t_cap = d.cap[f], t_mu = d.exp_t[f];
        const double t = t_cap - t_mu * 0.5 + smp * t_mu / (d.S - 1 + 1e-8);
        spline_segment(t, d.t0, d.dt, idx, u);
        oob = idx < 0 || idx + KD > d.N;
        if (oob) idx = idx < 0 ? 0 : d.N - KD; // the reference reads out of bounds here; clamp for memory safety and report
    }

    // stage B for one (sample, knot = wave, column) lane + the pose record by knot 0 / column 0
    template <int KD, bool WITH_J>
    __device__ __forceinline__ void pose_stage_b(const double *knots_t, const double *knots_R, int idx, double u, int wave, int col,
                                                 const SplineSeg *sg, PoseEntry<KD> &pe)
    {
        double kR[4 * KD];
        for (int i = 0; i < 4 * KD; ++i) kR[i] = knots_R[4 * idx + i];
        Quat q;
        if constexpr (WITH_J)
        {
            if constexpr (KD == 2)
                q = wave == 0 ? pose_table_entry<KD, 0>(kR, u, col, sg, pe) : pose_table_entry<KD, 1>(kR, u, col, sg, pe);
            else
                switch (wave)
                {
                case 0: q = pose_table_entry<KD, 0>(kR, u, col, sg, pe); break;
                case 1: q = pose_table_entry<KD, 1>(kR, u, col, sg, pe); break;
                case 2: q = pose_table_entry<KD, KD == 4 ? 2 : 0>(kR, u, col, sg, pe); break;
                default: q = pose_table_entry<KD, KD == 4 ? 3 : 1>(kR, u, col, sg, pe); break;
                }
        }
        else
        {
            q = spline_rotation_from_segs<KD>(kR, sg);
            for (int i = 0; i < 9 * KD; ++i) pe.A[i] = 0.0;
        }
        if (wave == 0 && col == 0)
        {
            double kt[3 * KD];
            for (int i = 0; i < 3 * KD; ++i) kt[i] = knots_t[3 * idx + i];
            double c[KD], p[3], qv[4] = {q.x, q.y, q.z, q.w}, R[9];
            trans_coeffs<KD>(u, c);
            spline_translation<KD>(kt, c, p);
            rotation_entries(qv, R);
            double rt[3];
            rotated_translation(p, qv, rt);
            for (int i = 0; i < 3; ++i) { pe.t[i] = p[i]; pe.rt[i] = rt[i]; }
            for (int i = 0; i < 4; ++i) pe.q[i] = qv[i];
            for (int i = 0; i < 9; ++i) pe.R[i] = R[i];
            for (int i = 0; i < KD; ++i) pe.c[i] = c[i];
        }
    }

    // the whole chain on one lane (spline_rotation_knot: logs, exps and products back to back); used where there is a single
    // segment (k = 2) inside the fused kernel's prologue
    template <int KD, bool WITH_J>
    __device__ __forceinline__ void pose_unstaged(const double *knots_t, const double *knots_R, int idx, double u, int wave, int col,
                                                  PoseEntry<KD> &pe)
    {
        double kR[4 * KD];
        for (int i = 0; i < 4 * KD; ++i) kR[i] = knots_R[4 * idx + i];
        Quat q;
        if constexpr (WITH_J)
        {
            JacC<1> blk;
            q = wave == 0 ? spline_rotation_knot<KD, 1, 0>(kR, u, col, blk) : spline_rotation_knot<KD, 1, KD - 1>(kR, u, col, blk);
            static_assert(KD == 2, "one wave per knot beyond k = 2 goes through the staged form");
            const double L3[4][3] = {{q.w, -q.z, q.y}, {q.z, q.w, -q.x}, {-q.y, q.x, q.w}, {-q.x, -q.y, -q.z}};
            const Quat &v = blk.c[0];
            double av[3];
            for (int a = 0; a < 3; ++a)
            {
                double r = L3[0][a] * v.x;
                r += L3[1][a] * v.y;
                r += L3[2][a] * v.z;
                r += L3[3][a] * v.w;
                av[a] = 2.0 * r;
            }
            store_rotated_column<KD>(q, av, 3 * wave + col, pe);
        }
        else
            q = spline_rotation<KD, false>(kR, u, nullptr);
        if (wave == 0 && col == 0)
        {
            double kt[3 * KD];
            for (int i = 0; i < 3 * KD; ++i) kt[i] = knots_t[3 * idx + i];
            double c[KD], p[3], qv[4] = {q.x, q.y, q.z, q.w}, R[9];
            trans_coeffs<KD>(u, c);
            spline_translation<KD>(kt, c, p);
            rotation_entries(qv, R);
            double rt[3];
            rotated_translation(p, qv, rt);
            for (int i = 0; i < 3; ++i) { pe.t[i] = p[i]; pe.rt[i] = rt[i]; }
            for (int i = 0; i < 4; ++i) pe.q[i] = qv[i];
            for (int i = 0; i < 9; ++i) pe.R[i] = R[i];
            for (int i = 0; i < KD; ++i) pe.c[i] = c[i];
        }
    }


    // S pose entries of (problem d, frame) -> dst[0 .. S-1] (LDS), in the two stages of k_pose_table: segment g of sample
    // `lane` on wave g into `segs` (LDS scratch: S x (KD - 1) SplineSeg), a workgroup barrier, then knot = wave and
    // lane = (sample, column).  EVERY wave of the workgroup calls this (it contains barriers); the caller synchronises
    // the workgroup once more afterwards.
    // STAGE2: k = 2 through the two stages as well (the lane-per-pixel kernels: their 16-wave / 128-register budget does not hold
    // the one-lane chain -- 36 vector spills --, the segment evaluation as a real call does)
    template <int KD, bool WITH_J, bool STAGE2 = false>
    __device__ __forceinline__ void frame_pose_entries(const ProblemDesc &d, const double *knots_t, const double *knots_R, int frame,
                                                       PoseEntry<KD> *dst, SplineSeg *segs, int wave, int lane, int *status, bool report)
    {
        constexpr int NCOL = WITH_J ? 3 : 1, NKW = WITH_J ? KD : 1, NSEG = KD - 1;
        const int S = d.S;
        if constexpr (KD == 2 && !STAGE2)
        { // one segment only: nothing to spread out, every (sample, column) lane evaluates it itself -- no barrier, no LDS
          // round trip (the staged form measured +1.1 us per evaluation here: the sample-parallel kernels)
            (void)segs;
            for (int s0 = 0; s0 < S; s0 += kPoseSPB)
            {
                const int sl = lane / NCOL, col = lane - sl * NCOL, smp = s0 + sl;
                if (wave < NKW && sl < kPoseSPB && smp < S)
                {
                    int idx;
                    double u;
                    bool oob;
                    pose_sample_segment<KD>(d, frame, smp, idx, u, oob);
                    if (oob && report && wave == 0 && col == 0) atomicAdd(status, 1);
                    pose_unstaged<KD, WITH_J>(knots_t, knots_R, idx, u, wave, col, dst[smp]);
                }
            }
            return;
        }
        for (int s0 = 0; s0 < S; s0 += kPoseSPB)
        {
            if (wave < NSEG && lane < kPoseSPB && s0 + lane < S)
            {
                int idx;
                double u;
                bool oob;
                pose_sample_segment<KD>(d, frame, s0 + lane, idx, u, oob);
                spline_segment_eval<WITH_J>(knots_R + 4 * (idx + wave), knots_R + 4 * (idx + wave + 1), seg_weight<KD>(u, wave),
                                            segs[lane * NSEG + wave]);
            }
            __syncthreads();
            MBAVO_PSTAMP(2);
            const int sl = lane / NCOL, col = lane - sl * NCOL, smp = s0 + sl;
            if (wave < NKW && sl < kPoseSPB && smp < S)
            {
                int idx;
                double u;
                bool oob;
                pose_sample_segment<KD>(d, frame, smp, idx, u, oob);
                if (oob && report && wave == 0 && col == 0) atomicAdd(status, 1);
                pose_stage_b<KD, WITH_J>(knots_t, knots_R, idx, u, wave, col, segs + sl * NSEG, dst[smp]);
            }
            if (s0 + kPoseSPB < S) __syncthreads(); // the next pass overwrites the segments
        }
    }

} // namespace mbavo

#endif
