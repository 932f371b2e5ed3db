// ref_shim.cpp -- ORACLE INFRASTRUCTURE (not product code).
//
// extern "C" driver around the parts of the REFERENCE that compile with plain
// g++ (no CUDA / Eigen / Sophus / OpenCV): the header-only __CPU_AND_CUDA_CODE__
// math and the two dependency-free .cpp files.  The reference sources are
// compiled from where they lie under /root/reference (see oracle/Makefile);
// nothing from them is copied into this repository.  Output: oracle/_ref/.
//
// What it covers (reference file -> exported symbol):
//   core/common/Quaternion.h            ref_quat_log / ref_quat_exp / ref_quat_mul / ref_quat_rotate
//   core/common/SplineFunctor.h         ref_spline_segment, ref_c{2,4}_vec3, ref_c{2,4}_rot3
//   ba_tracker/compute_pixel_intensity.h ref_bilinear, ref_pixel_intensity
//   core/measurements/ImagePyramid.h    ref_pyramid_u8
//   core/image_proc/Gradient.h          ref_image_gradients_u8
//   ba_tracker/levenberg_marquardt_strategy.cpp, trust_region_step_evaluator.cpp  ref_lm_*, ref_tr_*
//   stage drivers over whole problems, with the REFERENCE's per-sample code inside and only the kernels' launch
//   geometry / block reductions restated (the .cu files cannot be compiled):
//     ref_compute_virtual_camera_poses      kernel body of compute_virtual_camera_poses.cu:26-109 (spline functors)
//     ref_compute_local_patches_xy          kernel body of compute_local_patches_xy.cu:19-49 (Vector3d, Quaterniond)
//     ref_compute_pixel_jacobian_residual   kernel body of compute_hessian_gradients_cost.cu:51-153
//                                           (compute_pixel_intensity<double>, Core::MatrixMatrixMultiply)
//     ref_evaluate_omp                      the three drivers above over a whole problem, keypoint chunks spread over OpenMP
//                                           threads (bench.py's CPU baseline on all host threads: a loop inside the compiled
//                                           code instead of a Python thread pool around it)
// The CUDA kernels (.cu), merge (Eigen), Spline.h (Sophus) cannot be built here.
#include <cmath>
#include <cstring>
#include <cstdio>
#include <vector>
#if defined(_OPENMP)
#include <omp.h>
#endif

#include "ba_tracker/compute_pixel_intensity.h"
#include "ba_tracker/levenberg_marquardt_strategy.h"
#include "ba_tracker/trust_region_step_evaluator.h"
#include "core/common/SmallBlas.h"
#include "core/common/SplineFunctor.h"
#include "core/image_proc/Gradient.h"
#include "core/measurements/ImagePyramid.h"

using namespace SLAM;
using namespace SLAM::Core;

extern "C" {

void ref_sizes(int out[4])
{
    out[0] = (int)sizeof(Vector2d);
    out[1] = (int)sizeof(Vector3d);
    out[2] = (int)sizeof(VectorX<double, 4>);
    out[3] = (int)sizeof(VectorX<int, 2>);
}

void ref_quat_mul(const double a[4], const double b[4], double o[4])
{
    Quaterniond r = Quaterniond(a[0], a[1], a[2], a[3]) * Quaterniond(b[0], b[1], b[2], b[3]);
    o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
}

void ref_quat_rotate(const double q[4], const double p[3], double o[3])
{
    Vector3d r = Quaterniond(q[0], q[1], q[2], q[3]) * Vector3d(p[0], p[1], p[2]);
    o[0] = r(0); o[1] = r(1); o[2] = r(2);
}

void ref_quat_log(const double q[4], double tangent[3], double *jac)
{
    Vector3d r = Quaterniond(q[0], q[1], q[2], q[3]).log(jac);
    tangent[0] = r(0); tangent[1] = r(1); tangent[2] = r(2);
}

void ref_quat_exp(const double tg[3], double q[4], double *jac)
{
    Vector3d t(tg[0], tg[1], tg[2]);
    Quaterniond r = Quaterniond::exp(t, jac);
    q[0] = r.x; q[1] = r.y; q[2] = r.z; q[3] = r.w;
}

void ref_spline_segment(double t, double t0, double dt, int *idx, double *u)
{
    SplineSegmentStartKnotIdxAndNormalizedU(t, t0, dt, *idx, *u);
}

void ref_c2_vec3(const double *knots, double u, double p[3], double *jac)
{
    Vector3d r = C2SplineVec3Functor(knots, u, jac);
    p[0] = r(0); p[1] = r(1); p[2] = r(2);
}

void ref_c4_vec3(const double *knots, double u, double p[3], double *jac)
{
    Vector3d r = C4SplineVec3Functor(knots, u, jac);
    p[0] = r(0); p[1] = r(1); p[2] = r(2);
}

void ref_c2_rot3(const double *knots, double u, double q[4], double *jac)
{
    double le[24], X[16], Y[16], Z[16];
    Quaterniond r = jac ? C2SplineRot3Functor(knots, u, jac, le, X, Y, Z) : C2SplineRot3Functor(knots, u);
    q[0] = r.x; q[1] = r.y; q[2] = r.z; q[3] = r.w;
}

void ref_c4_rot3(const double *knots, double u, double q[4], double *jac)
{
    double le[72], X[16], Y[16], Z[16];
    Quaterniond r = jac ? C4SplineRot3Functor(knots, u, jac, le, X, Y, Z) : C4SplineRot3Functor(knots, u);
    q[0] = r.x; q[1] = r.y; q[2] = r.z; q[3] = r.w;
}

int ref_bilinear(const unsigned char *I, const float *dIxy, int H, int W, double x, double y, double out[3])
{
    VectorX<double, 2> p; p.values[0] = x; p.values[1] = y;
    Vector3d r(0, 0, 0);
    bool ok = VO::bilinear_interpolation<double>(I, dIxy, H, W, p, r);
    out[0] = r(0); out[1] = r(1); out[2] = r(2);
    return ok ? 1 : 0;
}

int ref_pixel_intensity(const unsigned char *I_ref, const float *dIxy_ref, int H, int W,
                        const double R[4], const double t[3], double plane_depth,
                        double fx, double fy, double cx, double cy,
                        double cur_x, double cur_y, double *intensity, double *jac7)
{
    VectorX<double, 2> p; p.values[0] = cur_x; p.values[1] = cur_y;
    bool ok = VO::compute_pixel_intensity<double>(I_ref, dIxy_ref, H, W, R, t, plane_depth, fx, fy, cx, cy,
                                                  p, intensity, jac7);
    return ok ? 1 : 0;
}

// levels: writes level l (l = 1 .. nLevels-1) into out[l-1] (caller-sized H/2^l * W/2^l)
void ref_pyramid_u8(const unsigned char *src, int H, int W, int nLevels, unsigned char **out)
{
    Image<unsigned char> img(H, W, 1);
    std::memcpy(img.getData(), src, (size_t)H * W);
    ImagePyramid<unsigned char> pyr;
    pyr.setNumOfPyramidLevels(nLevels);
    pyr.computePyramid(&img);
    for (int l = 1; l < nLevels; ++l) {
        Image<unsigned char> *im = pyr.getImagePtr(l);
        std::memcpy(out[l - 1], im->getData(), im->nHeight() * im->nWidth());
    }
}

void ref_image_gradients_u8(const unsigned char *src, int H, int W, float *dIxy, float *mag)
{
    Image<unsigned char> img(H, W, 1);
    std::memcpy(img.getData(), src, (size_t)H * W);
    Image<float> grad(H, W, 2);
    Image<float> m(H, W, 1);
    compute_image_gradients<unsigned char, float>(&img, &grad, mag ? &m : nullptr);
    std::memcpy(dIxy, grad.getData(), sizeof(float) * (size_t)H * W * 2);
    if (mag) std::memcpy(mag, m.getData(), sizeof(float) * (size_t)H * W);
}

void *ref_lm_new() { return new VO::LevenbergMarquardtStrategy(); }
void ref_lm_delete(void *p) { delete (VO::LevenbergMarquardtStrategy *)p; }
void ref_lm_reset(void *p) { ((VO::LevenbergMarquardtStrategy *)p)->reset(); }
void ref_lm_accepted(void *p, double q) { ((VO::LevenbergMarquardtStrategy *)p)->step_accepted(q); }
void ref_lm_rejected(void *p) { ((VO::LevenbergMarquardtStrategy *)p)->step_rejected(); }
double ref_lm_radius(void *p) { return ((VO::LevenbergMarquardtStrategy *)p)->get_radius(); }

void *ref_tr_new(int m) { return new VO::TrustRegionStepEvaluator(m); }
void ref_tr_delete(void *p) { delete (VO::TrustRegionStepEvaluator *)p; }
void ref_tr_reset(void *p, double c) { ((VO::TrustRegionStepEvaluator *)p)->reset(c); }
double ref_tr_quality(void *p, double c, double m) { return ((VO::TrustRegionStepEvaluator *)p)->StepQuality(c, m); }
void ref_tr_accepted(void *p, double c, double m) { ((VO::TrustRegionStepEvaluator *)p)->StepAccepted(c, m); }


// ---- stage drivers (whole problems) ----------------------------------------------------------------------------
// block reduction of reduction.h:13-55 as the kernels use it: pairwise tree for power-of-two sizes; other sizes are
// racy in the reference and DEFINED here as the plain sum
static double ref_tree_sum(double *buf, int n)
{
    if (n <= 0) return 0.0;
    if ((n & (n - 1)) == 0)
    {
        for (int s = n / 2; s >= 1; s /= 2)
            for (int i = 0; i < s; ++i) buf[i] = buf[i] + buf[i + s];
        return buf[0];
    }
    double acc = 0.0;
    for (int i = 0; i < n; ++i) acc += buf[i];
    return acc;
}

void ref_compute_virtual_camera_poses(int S, int F, const double *cap, const double *exp_t, int k, double t0, double dt,
                                      const double *knots_t, const double *knots_R, double *poses, double *J_t,
                                      double *J_R)
{
    double le[72], X[16], Y[16], Z[16];
    for (int f = 0; f < F; ++f)
        for (int i = 0; i < S; ++i)
        {
            const int v = f * S + i;
            const double t_cap = cap[f], t_mu = exp_t[f];
            const double t = t_cap - t_mu * 0.5 + i * t_mu / (S - 1 + 1e-8); // compute_virtual_camera_poses.cu:33
            int idx;
            double u;
            SplineSegmentStartKnotIdxAndNormalizedU(t, t0, dt, idx, u);
            double *jt = J_t ? J_t + (size_t)v * 9 * k : nullptr;
            double *jr = J_R ? J_R + (size_t)v * 12 * k : nullptr;
            Vector3d p;
            Quaterniond q;
            if (k == 2)
            {
                p = C2SplineVec3Functor(knots_t + idx * 3, u, jt);
                q = jr ? C2SplineRot3Functor(knots_R + idx * 4, u, jr, le, X, Y, Z) : C2SplineRot3Functor(knots_R + idx * 4, u);
            }
            else
            {
                p = C4SplineVec3Functor(knots_t + idx * 3, u, jt);
                q = jr ? C4SplineRot3Functor(knots_R + idx * 4, u, jr, le, X, Y, Z) : C4SplineRot3Functor(knots_R + idx * 4, u);
            }
            double *o = poses + (size_t)v * 7;
            o[0] = p(0); o[1] = p(1); o[2] = p(2);
            o[3] = q.x; o[4] = q.y; o[5] = q.z; o[6] = q.w;
        }
}

// jacobians may be null (cost-only).  A pixel whose location or any of whose S samples is out of bounds gets
// residual 0 and a zero row (the reference leaves stale values there).
void ref_compute_pixel_jacobian_residual(const unsigned char *I_ref, const float *dIxy_ref, const unsigned char *const *I_cur,
                                         int S, int F, const double *poses, int k, const double *J_t, const double *J_R,
                                         const double *centres, const double *kp_z, int K, const int *pattern, int P,
                                         const double intr[4], int H, int W, double *residuals, double *jacobians)
{
    const int n6k = 6 * k;
    double *ints = new double[S];
    double *chain = new double[(size_t)S * n6k];
    for (int f = 0; f < F; ++f)
        for (int i = 0; i < K; ++i)
            for (int pp = 0; pp < P; ++pp)
            {
                const size_t g = ((size_t)f * K + i) * P + pp;
                residuals[g] = 0.0;
                if (jacobians) std::memset(jacobians + g * n6k, 0, sizeof(double) * n6k);
                const int px = centres[((size_t)f * K + i) * 2] + pattern[2 * pp];     // :69-70, int truncation
                const int py = centres[((size_t)f * K + i) * 2 + 1] + pattern[2 * pp + 1];
                if (px < 0 || px > W - 1 || py < 0 || py > H - 1) continue;
                const Vector2d cur(px, py);
                bool ok = true;
                for (int s = 0; s < S && ok; ++s)
                {
                    const double *t_c2r = poses + ((size_t)f * S + s) * 7, *R_c2r = t_c2r + 3;
                    double val, j7[7];
                    ok = VO::compute_pixel_intensity<double>(I_ref, dIxy_ref, H, W, R_c2r, t_c2r, kp_z[i], intr[0], intr[1],
                                                             intr[2], intr[3], cur, &val, jacobians ? j7 : nullptr);
                    if (!ok) break;
                    ints[s] = val;
                    if (jacobians)
                    { // :136-142
                        double *c = chain + (size_t)s * n6k;
                        MatrixMatrixMultiply<double, double, double, 0>(j7, 1, 3, J_t + ((size_t)f * S + s) * 9 * k, 3, 3 * k, c, 0, 0, 1,
                                                                         3 * k);
                        MatrixMatrixMultiply<double, double, double, 0>(j7 + 3, 1, 4, J_R + ((size_t)f * S + s) * 12 * k, 4, 3 * k, c, 0,
                                                                         3 * k, 1, 3 * k);
                    }
                }
                if (!ok) continue;
                const double sum = ref_tree_sum(ints, S);
                const double intensity_cur = *(I_cur[f] + py * W + px);
                residuals[g] = sum / float(S) - intensity_cur; // :120
                if (jacobians)
                    for (int c = 0; c < n6k; ++c)
                    {
                        for (int s = 0; s < S; ++s) ints[s] = chain[(size_t)s * n6k + c];
                        jacobians[g * n6k + c] = ref_tree_sum(ints, S) / float(S); // :145-153
                    }
            }
    delete[] ints;
    delete[] chain;
}

// kernel body of compute_local_patches_xy.cu:19-49 with the reference's Vector3d / Quaterniond classes
void ref_compute_local_patches_xy(int S, int F, const double *poses, const double *kp_xy, const double *kp_z, int K,
                                  const double intr[4], double *centres)
{
    for (int f = 0; f < F; ++f)
        for (int i = 0; i < K; ++i)
        {
            const int pose_idx = f * S + S / 2;
            Vector3d P3dr;
            P3dr(0) = kp_z[i] * (kp_xy[2 * i] - intr[2]) / intr[0];
            P3dr(1) = kp_z[i] * (kp_xy[2 * i + 1] - intr[3]) / intr[1];
            P3dr(2) = kp_z[i];
            const double *cam_pose = poses + (size_t)pose_idx * 7;
            const Vector3d t_c2r(cam_pose[0], cam_pose[1], cam_pose[2]);
            const Quaterniond R_c2r(cam_pose[3], cam_pose[4], cam_pose[5], cam_pose[6]);
            const Quaterniond R_r2c = R_c2r.conjugate();
            const Vector3d t_r2c = -(R_r2c * t_c2r);
            const Vector3d P3dc = R_r2c * P3dr + t_r2c;
            centres[((size_t)f * K + i) * 2] = P3dc(0) / P3dc(2) * intr[0] + intr[2];
            centres[((size_t)f * K + i) * 2 + 1] = P3dc(1) / P3dc(2) * intr[1] + intr[3];
        }
}

// One whole H/g evaluation on `threads` OpenMP threads: poses and patch centres once, then chunks of `chunk` keypoints of a
// frame in parallel -- per chunk the reference's per-sample code (ref_compute_pixel_jacobian_residual above), the per-patch
// Huber / packed outer product through `patch_fn` (the oracle's restatement of compute_hessian_gradients_cost.cu:165-239,
// handed over as a function pointer: the two checker libraries do not link against each other) and the chunk's frame sum.
// Every thread adds its chunks in ascending order into its own frame blocks; the threads' blocks are added in thread order
// (static schedule: deterministic for a given thread count).  frame_blocks: F x E, scaled by 1 / (K F P).
typedef void (*ref_patch_fn)(int F, int K, int P, int k, const double *res, const double *jac, double huber_a, double inv, double *pb);
int ref_evaluate_omp(const unsigned char *I_ref, const float *dIxy_ref, const unsigned char *const *I_cur, int S, int F,
                     const double *cap, const double *exp_t, int k, double t0, double dt, const double *knots_t,
                     const double *knots_R, const double *kp_xy, const double *kp_z, int K, const int *pattern, int P,
                     const double intr[4], int H, int W, double huber_a, ref_patch_fn patch_fn, int threads, int chunk,
                     double *frame_blocks)
{
    const int nd = 6 * k + 1, E = nd * (nd + 1) / 2, n6k = 6 * k;
    std::vector<double> poses((size_t)F * S * 7), Jt((size_t)F * S * 9 * k), JR((size_t)F * S * 12 * k), centres((size_t)F * K * 2 + 2);
    ref_compute_virtual_camera_poses(S, F, cap, exp_t, k, t0, dt, knots_t, knots_R, poses.data(), Jt.data(), JR.data());
    ref_compute_local_patches_xy(S, F, poses.data(), kp_xy, kp_z, K, intr, centres.data());
    if (chunk < 1) chunk = 64;
    const int nchunk = (K + chunk - 1) / chunk;
    const long jobs = (long)F * nchunk;
    if (threads < 1) threads = 1;
    const double inv = (double)K * F * P > 0 ? 1.0 / ((double)K * F * P) : 0.0;
    std::vector<double> acc((size_t)threads * F * E, 0.0);
    int used = 1;
#if defined(_OPENMP)
#pragma omp parallel num_threads(threads)
#endif
    {
        int tid = 0, nth = 1;
#if defined(_OPENMP)
        tid = omp_get_thread_num();
        nth = omp_get_num_threads();
#pragma omp single
        used = nth;
#endif
        std::vector<double> res((size_t)chunk * P), jac((size_t)chunk * P * n6k), pb((size_t)chunk * E);
        double *mine = acc.data() + (size_t)tid * F * E;
        const long lo = jobs * tid / nth, hi = jobs * (tid + 1) / nth; // contiguous share, ascending
        for (long j = lo; j < hi; ++j)
        {
            const int f = (int)(j / nchunk), i0 = (int)(j % nchunk) * chunk, n = (K - i0) < chunk ? (K - i0) : chunk;
            ref_compute_pixel_jacobian_residual(I_ref, dIxy_ref, I_cur + f, S, 1, poses.data() + (size_t)f * S * 7, k,
                                                Jt.data() + (size_t)f * S * 9 * k, JR.data() + (size_t)f * S * 12 * k,
                                                centres.data() + ((size_t)f * K + i0) * 2, kp_z + i0, n, pattern, P, intr, H, W,
                                                res.data(), jac.data());
            patch_fn(1, n, P, k, res.data(), jac.data(), huber_a, inv, pb.data());
            double *dst = mine + (size_t)f * E;
            for (int i = 0; i < n; ++i)
                for (int e = 0; e < E; ++e) dst[e] += pb[(size_t)i * E + e];
        }
    }
    for (size_t e = 0; e < (size_t)F * E; ++e)
    {
        double v = 0.0;
        for (int t = 0; t < used; ++t) v += acc[(size_t)t * F * E + e];
        frame_blocks[e] = v;
    }
    return used;
}
} // extern "C"
