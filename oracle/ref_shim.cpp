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
// The CUDA kernels (.cu), merge (Eigen), Spline.h (Sophus) cannot be built here.
#include <cmath>
#include <cstring>
#include <cstdio>

#include "ba_tracker/compute_pixel_intensity.h"
#include "ba_tracker/levenberg_marquardt_strategy.h"
#include "ba_tracker/trust_region_step_evaluator.h"
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

} // extern "C"
