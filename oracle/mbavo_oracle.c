/*
 * mbavo_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See mbavo_oracle.h for the rules that govern this file.
 *
 * Restates, operation for operation, the arithmetic of the reference hot path
 * (paths relative to /root/reference/src):
 *   core/common/Quaternion.h, SplineFunctor.h, SmallBlas.h
 *   ba_tracker/compute_pixel_intensity.h
 *   ba_tracker/compute_virtual_camera_poses.cu, compute_local_patches_xy.cu
 *   ba_tracker/compute_hessian_gradients_cost.cu, reduction.h
 *   ba_tracker/merge_hessian_gradient_cost.cpp, spline_update_step.cpp
 *   ba_tracker/levenberg_marquardt_strategy.cpp, trust_region_step_evaluator.cpp
 *   ba_tracker/blur_aware_direct_tracker.cpp (LM loop), generate_synthetic_data.cpp
 *   core/measurements/ImagePyramid.h, core/image_proc/Gradient.h
 * fp32 islands (sqrtf, float bilinear weights, Huber sqrtf) are kept where the
 * reference has them.  Compile with -ffp-contract=off.
 */
#include "mbavo_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------ */
/* small dense products: C (op) A*B, k ascending, accumulator starts at 0    */
/* (SmallBlas.h:152-225 + SmallBlasGeneric.h:101-139: every variant sums in  */
/* ascending k from 0, so one loop nest reproduces the rounding)            */
/* ------------------------------------------------------------------------ */
static void mm(const double *A, int ra, int ca, const double *B, int cb,
               double *C, int sr, int sc, int cstride, int accumulate)
{
    for (int r = 0; r < ra; ++r)
        for (int c = 0; c < cb; ++c) {
            double acc = 0.0;
            for (int k = 0; k < ca; ++k)
                acc += A[r * ca + k] * B[k * cb + c];
            double *dst = &C[(r + sr) * cstride + sc + c];
            if (accumulate) *dst += acc; else *dst = acc;
        }
}

/* ------------------------------------------------------------------------ */
/* quaternions, memory order x y z w (Quaternion.h:13-18)                    */
/* ------------------------------------------------------------------------ */
void orc_quat_mul(const double a[4], const double b[4], double o[4])
{ /* Quaternion.h:45-51 */
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
    const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
    const double x = aw * bx + ax * bw + ay * bz - az * by;
    const double y = aw * by + ay * bw + az * bx - ax * bz;
    const double z = aw * bz + az * bw + ax * by - ay * bx;
    const double w = aw * bw - ax * bx - ay * by - az * bz;
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}

static void quat_conj(const double q[4], double o[4])
{ o[0] = -q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = q[3]; }

void orc_quat_rotate(const double q[4], const double p[3], double o[3])
{ /* Quaternion.h:53-60: V = Q * (p,0) * conj(Q) */
    double qp[4] = { p[0], p[1], p[2], 0.0 }, qc[4], t[4], v[4];
    quat_conj(q, qc);
    orc_quat_mul(q, qp, t);
    orc_quat_mul(t, qc, v);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
}

void orc_quat_log(const double q[4], double tangent[3], double *J)
{ /* Quaternion.h:61-157, jacobian 3x4 row-major (wx,wy,wz) x (qx,qy,qz,qw) */
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double squared_n = x * x + y * y + z * z;
    double lambda;
    double dx = 0, dy = 0, dz = 0, dw = 0;
    if (squared_n < 1e-20) {
        const double www = w * w * w;
        lambda = 2. / w - 2. / 3. * squared_n / www;
        if (J) { /* the reference's series derivative, reproduced as written (:80-88) */
            dx = 2. / w - 4. / 3. * x / www;
            dy = 2. / w - 4. / 3. * y / www;
            dz = 2. / w - 4. / 3. * z / www;
            dw = -2 / (w * w) + 2 * squared_n / (www * w);
        }
    } else {
        const double n = sqrt(squared_n);
        if (fabs(w) < 1e-10) {
            if (w > 0) {
                lambda = M_PI / n;
                if (J) { dx = -lambda / squared_n * x; dy = -lambda / squared_n * y; dz = -lambda / squared_n * z; }
            } else {
                lambda = -M_PI / n;
                if (J) { dx = lambda / squared_n * x; dy = lambda / squared_n * y; dz = lambda / squared_n * z; }
            }
        } else {
            lambda = 2.0 * atan(n / w) / n;
            if (J) {
                const double dlambda_dn = (2 * w - lambda) / n;
                dx = dlambda_dn * x / n;
                dy = dlambda_dn * y / n;
                dz = dlambda_dn * z / n;
                dw = -2.;
            }
        }
    }
    if (J) {
        J[0] = dx * x + lambda; J[1] = dy * x;          J[2] = dz * x;           J[3] = dw * x;
        J[4] = dx * y;          J[5] = dy * y + lambda; J[6] = dz * y;           J[7] = dw * y;
        J[8] = dx * z;          J[9] = dy * z;          J[10] = dz * z + lambda; J[11] = dw * z;
    }
    tangent[0] = lambda * x; tangent[1] = lambda * y; tangent[2] = lambda * z;
}

void orc_quat_exp(const double tg[3], double q[4], double *J)
{ /* Quaternion.h:159-233, jacobian 4x3 row-major (qx,qy,qz,qw) x (wx,wy,wz) */
    double imag, real;
    const double theta_sq = tg[0] * tg[0] + tg[1] * tg[1] + tg[2] * tg[2];
    if (theta_sq < 1e-20) {
        const double theta_po4 = theta_sq * theta_sq;
        imag = 0.5 - 1. / 48. * theta_sq + 1. / 3840. * theta_po4;
        real = 1. - 1. / 8. * theta_sq + 1. / 384. * theta_po4;
        if (J) { memset(J, 0, sizeof(double) * 12); J[0] = 0.5; J[4] = 0.5; J[8] = 0.5; }
    } else {
        const double theta = sqrt(theta_sq);
        const double half_theta = 0.5 * theta;
        const double sin_half = sin(half_theta);
        imag = sin_half / theta;
        real = cos(half_theta);
        if (J) {
            const double x = tg[0], y = tg[1], z = tg[2];
            const double dth_dx = x / theta, dth_dy = y / theta, dth_dz = z / theta;
            const double dimag_dth = 0.5 * real / theta - imag / theta;
            const double dreal_dth = -0.5 * sin_half;
            const double dimag_dx = dimag_dth * dth_dx, dimag_dy = dimag_dth * dth_dy, dimag_dz = dimag_dth * dth_dz;
            J[0] = dimag_dx * x + imag; J[1] = dimag_dy * x;        J[2] = dimag_dz * x;
            J[3] = dimag_dx * y;        J[4] = dimag_dy * y + imag; J[5] = dimag_dz * y;
            J[6] = dimag_dx * z;        J[7] = dimag_dy * z;        J[8] = dimag_dz * z + imag;
            J[9] = dreal_dth * dth_dx;  J[10] = dreal_dth * dth_dy; J[11] = dreal_dth * dth_dz;
        }
    }
    q[0] = imag * tg[0]; q[1] = imag * tg[1]; q[2] = imag * tg[2]; q[3] = real;
}

static void left_matrix(const double q[4], double Q[16])
{ /* Quaternion.h:239-260 */
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    Q[0] = w;  Q[1] = -z; Q[2] = y;   Q[3] = x;
    Q[4] = z;  Q[5] = w;  Q[6] = -x;  Q[7] = y;
    Q[8] = -y; Q[9] = x;  Q[10] = w;  Q[11] = z;
    Q[12] = -x; Q[13] = -y; Q[14] = -z; Q[15] = w;
}
static void right_matrix(const double q[4], double Q[16])
{ /* Quaternion.h:262-283 */
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    Q[0] = w;  Q[1] = z;  Q[2] = -y;  Q[3] = x;
    Q[4] = -z; Q[5] = w;  Q[6] = x;   Q[7] = y;
    Q[8] = y;  Q[9] = -x; Q[10] = w;  Q[11] = z;
    Q[12] = -x; Q[13] = -y; Q[14] = -z; Q[15] = w;
}
/* SplineFunctor.h:96-153: the three helper macros */
static void m44_half_cols3(const double X[16], double Y[12])
{ for (int r = 0; r < 4; ++r) for (int c = 0; c < 3; ++c) Y[r * 3 + c] = X[r * 4 + c] * 0.5; }
static void m44_k44(const double X[16], double Y[16])
{ for (int r = 0; r < 4; ++r) { Y[r*4+0] = -X[r*4+0]; Y[r*4+1] = -X[r*4+1]; Y[r*4+2] = -X[r*4+2]; Y[r*4+3] = X[r*4+3]; } }
static void m_scale16(double X[16], double s)
{ for (int i = 0; i < 16; ++i) X[i] = X[i] * s; }

/* Sophus::SO3d::exp (third-party, unpinned; published series/closed form with
 * Constants<double>::epsilon() = 1e-10): call sites Spline.h:302,326 */
void orc_so3_exp(const double om[3], double q[4])
{
    const double theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
    double imag, real;
    if (theta_sq < 1e-10 * 1e-10) {
        const double theta_po4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
        real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_po4;
    } else {
        const double theta = sqrt(theta_sq);
        const double half_theta = 0.5 * theta;
        const double sin_half = sin(half_theta);
        imag = sin_half / theta;
        real = cos(half_theta);
    }
    q[0] = imag * om[0]; q[1] = imag * om[1]; q[2] = imag * om[2]; q[3] = real;
}

/* ------------------------------------------------------------------------ */
/* spline functors                                                           */
/* ------------------------------------------------------------------------ */
void orc_spline_segment(double t, double t0, double dt, int *start_idx, double *u)
{ /* SplineFunctor.h:13-19 */
    const double tn = (t - t0) / dt;
    *start_idx = (int)tn;
    *u = tn - *start_idx;
}

void orc_c2_vec3(const double *d, double u, double p[3], double *J)
{ /* SplineFunctor.h:21-43 */
    const double omu = 1 - u;
    p[0] = omu * d[0] + u * d[3];
    p[1] = omu * d[1] + u * d[4];
    p[2] = omu * d[2] + u * d[5];
    if (J) {
        memset(J, 0, sizeof(double) * 18);
        J[0] = omu; J[3] = u; J[7] = omu; J[10] = u; J[14] = omu; J[17] = u;
    }
}

void orc_c4_vec3(const double *d, double u, double p[3], double *J)
{ /* SplineFunctor.h:45-94 */
    const double uu = u * u, uuu = uu * u;
    const double s = 1. / 6.;
    const double c0 = s - 0.5 * u + 0.5 * uu - s * uuu;
    const double c1 = 4 * s - uu + 0.5 * uuu;
    const double c2 = s + 0.5 * u + 0.5 * uu - 0.5 * uuu;
    const double c3 = s * uuu;
    for (int a = 0; a < 3; ++a)
        p[a] = c0 * d[a] + c1 * d[3 + a] + c2 * d[6 + a] + c3 * d[9 + a];
    if (J) {
        memset(J, 0, sizeof(double) * 36);
        for (int a = 0; a < 3; ++a) {
            J[a * 12 + a] = c0; J[a * 12 + 3 + a] = c1; J[a * 12 + 6 + a] = c2; J[a * 12 + 9 + a] = c3;
        }
    }
}

void orc_c2_rot3(const double *d, double u, double q[4], double *J)
{ /* SplineFunctor.h:155-217 */
    const double *R0 = d, *R1 = d + 4;
    double R0c[4], R01[4], om[3], A0[4];
    double dlog[12], dexp[12];
    double X[16], Y[16], Z[16];
    quat_conj(R0, R0c);
    orc_quat_mul(R0c, R1, R01);
    orc_quat_log(R01, om, J ? dlog : NULL);
    om[0] = om[0] * u; om[1] = om[1] * u; om[2] = om[2] * u;
    orc_quat_exp(om, A0, J ? dexp : NULL);
    if (J) {
        memset(J, 0, sizeof(double) * 24);
        /* d/dR0 */
        left_matrix(R0, X); m44_half_cols3(X, Y);
        right_matrix(A0, X);
        mm(X, 4, 4, Y, 3, J, 0, 0, 6, 1);
        right_matrix(R1, Z); m44_k44(Z, X);
        mm(X, 4, 4, Y, 3, Z, 0, 0, 3, 0);
        mm(dlog, 3, 4, Z, 3, X, 0, 0, 3, 0);
        m_scale16(X, u);
        mm(dexp, 4, 3, X, 3, Y, 0, 0, 3, 0);
        left_matrix(R0, X);
        mm(X, 4, 4, Y, 3, J, 0, 0, 6, 1);
        /* d/dR1 */
        left_matrix(R1, X); m44_half_cols3(X, Y);
        left_matrix(R0c, X);
        mm(X, 4, 4, Y, 3, Z, 0, 0, 3, 0);
        mm(dlog, 3, 4, Z, 3, X, 0, 0, 3, 0);
        m_scale16(X, u);
        mm(dexp, 4, 3, X, 3, Y, 0, 0, 3, 0);
        left_matrix(R0, X);
        mm(X, 4, 4, Y, 3, J, 0, 3, 6, 1);
    }
    orc_quat_mul(R0, A0, q);
}

void orc_c4_rot3(const double *d, double u, double q[4], double *J)
{ /* SplineFunctor.h:219-365 */
    const double uu = u * u, uuu = uu * u;
    const double s = 1. / 6.;
    const double c1 = 5 * s + 0.5 * u - 0.5 * uu + s * uuu;
    const double c2 = s + 0.5 * u + 0.5 * uu - 2 * s * uuu;
    const double c3 = s * uuu;
    const double *R0 = d, *R1 = d + 4, *R2 = d + 8, *R3 = d + 12;
    double R0c[4], R1c[4], R2c[4], R01[4], R12[4], R23[4];
    double om01[3], om12[3], om23[3], A0[4], A1[4], A2[4];
    double dl01[12], dl12[12], dl23[12], de0[12], de1[12], de2[12];
    double X[16], Y[16], Z[16], T[4], T2[4];
    quat_conj(R0, R0c); quat_conj(R1, R1c); quat_conj(R2, R2c);
    orc_quat_mul(R0c, R1, R01);
    orc_quat_mul(R1c, R2, R12);
    orc_quat_mul(R2c, R3, R23);
    orc_quat_log(R01, om01, J ? dl01 : NULL);
    orc_quat_log(R12, om12, J ? dl12 : NULL);
    orc_quat_log(R23, om23, J ? dl23 : NULL);
    for (int a = 0; a < 3; ++a) { om01[a] = om01[a] * c1; om12[a] = om12[a] * c2; om23[a] = om23[a] * c3; }
    orc_quat_exp(om01, A0, J ? de0 : NULL);
    orc_quat_exp(om12, A1, J ? de1 : NULL);
    orc_quat_exp(om23, A2, J ? de2 : NULL);

    if (J) {
        double A012[4], A12[4], R0A0[4], R0A0A1[4];
        orc_quat_mul(A0, A1, T); orc_quat_mul(T, A2, A012);   /* (A0*A1*A2) */
        orc_quat_mul(A1, A2, A12);
        orc_quat_mul(R0, A0, R0A0);
        orc_quat_mul(R0A0, A1, R0A0A1);                        /* (R0*A0*A1) */
        memset(J, 0, sizeof(double) * 48);

        /* ---- R0 (:263-282) */
        left_matrix(R0, X); m44_half_cols3(X, Y);
        right_matrix(A012, X);
        mm(X, 4, 4, Y, 3, J, 0, 0, 12, 1);
        right_matrix(R1, Z); m44_k44(Z, X);
        mm(X, 4, 4, Y, 3, Z, 0, 0, 3, 0);
        mm(dl01, 3, 4, Z, 3, X, 0, 0, 3, 0);
        m_scale16(X, c1);
        mm(de0, 4, 3, X, 3, Y, 0, 0, 3, 0);
        right_matrix(A12, X);
        mm(X, 4, 4, Y, 3, Z, 0, 0, 3, 0);
        left_matrix(R0, Y);
        mm(Y, 4, 4, Z, 3, J, 0, 0, 12, 1);

        /* ---- R1 first term (:284-295) */
        left_matrix(R1, X); m44_half_cols3(X, Y);
        left_matrix(R0c, X);
        mm(X, 4, 4, Y, 3, Z, 0, 0, 3, 0);
        mm(dl01, 3, 4, Z, 3, X, 0, 0, 3, 0);
        m_scale16(X, c1);
        mm(de0, 4, 3, X, 3, Y, 0, 0, 3, 0);
        right_matrix(A12, X);
        mm(X, 4, 4, Y, 3, Z, 0, 0, 3, 0);
        left_matrix(R0, X);
        mm(X, 4, 4, Z, 3, J, 0, 3, 12, 1);
        /* ---- R1 second term (:297-310) */
        left_matrix(R1, X); m44_half_cols3(X, Y);
        right_matrix(R2, X); m44_k44(X, Z);
        mm(Z, 4, 4, Y, 3, X, 0, 0, 3, 0);
        mm(dl12, 3, 4, X, 3, Y, 0, 0, 3, 0);
        m_scale16(Y, c2);
        mm(de1, 4, 3, Y, 3, Z, 0, 0, 3, 0);
        right_matrix(A2, X);
        mm(X, 4, 4, Z, 3, Y, 0, 0, 3, 0);
        left_matrix(R0A0, X);
        mm(X, 4, 4, Y, 3, J, 0, 3, 12, 1);

        /* ---- R2 first term (:312-323) */
        left_matrix(R2, X); m44_half_cols3(X, Y);
        left_matrix(R1c, X);
        mm(X, 4, 4, Y, 3, Z, 0, 0, 3, 0);
        mm(dl12, 3, 4, Z, 3, X, 0, 0, 3, 0);
        m_scale16(X, c2);
        mm(de1, 4, 3, X, 3, Y, 0, 0, 3, 0);
        right_matrix(A2, X);
        mm(X, 4, 4, Y, 3, Z, 0, 0, 3, 0);
        left_matrix(R0A0, X);
        mm(X, 4, 4, Z, 3, J, 0, 6, 12, 1);
        /* ---- R2 second term (:325-335) */
        left_matrix(R2, X); m44_half_cols3(X, Y);
        right_matrix(R3, X); m44_k44(X, Z);
        mm(Z, 4, 4, Y, 3, X, 0, 0, 3, 0);
        mm(dl23, 3, 4, X, 3, Y, 0, 0, 3, 0);
        mm(de2, 4, 3, Y, 3, X, 0, 0, 3, 0);
        m_scale16(X, c3);
        left_matrix(R0A0A1, Y);
        mm(Y, 4, 4, X, 3, J, 0, 6, 12, 1);

        /* ---- R3 (:337-347) */
        left_matrix(R3, X); m44_half_cols3(X, Y);
        left_matrix(R2c, X);
        mm(X, 4, 4, Y, 3, Z, 0, 0, 3, 0);
        mm(dl23, 3, 4, Z, 3, X, 0, 0, 3, 0);
        mm(de2, 4, 3, X, 3, Z, 0, 0, 3, 0);
        m_scale16(Z, c3);
        left_matrix(R0A0A1, X);
        mm(X, 4, 4, Z, 3, J, 0, 9, 12, 1);
    }
    /* return R0 * A0 * A1 * A2 (:364) */
    orc_quat_mul(R0, A0, T); orc_quat_mul(T, A1, T2); orc_quat_mul(T2, A2, q);
}

/* ------------------------------------------------------------------------ */
/* per pixel-sample math                                                     */
/* ------------------------------------------------------------------------ */
int orc_bilinear(const unsigned char *I, const float *dIxy, int H, int W,
                 double x, double y, double out[3])
{ /* compute_pixel_intensity.h:25-72; A6: tap addresses clamped (their weight is 0 there) */
    if (x < 0 || x > W - 1 || y < 0 || y > H - 1) return 0;
    const int xi = (int)x, yi = (int)y;
    const float dx = (float)(x - xi);
    const float dy = (float)(y - yi);
    const float dxdy = dx * dy;
    const float w00 = 1.0f - dx - dy + dxdy;
    const float w01 = dx - dxdy;
    const float w10 = dy - dxdy;
    const float w11 = dxdy;
    const int x1 = xi + 1 < W ? xi + 1 : W - 1;
    const int y1 = yi + 1 < H ? yi + 1 : H - 1;
    const int i00 = yi * W + xi, i01 = yi * W + x1, i10 = y1 * W + xi, i11 = y1 * W + x1;
    float v;
    v = w11 * I[i11] + w10 * I[i10] + w01 * I[i01] + w00 * I[i00];
    out[0] = v;
    if (dIxy) {
        v = w11 * dIxy[2 * i11] + w10 * dIxy[2 * i10] + w01 * dIxy[2 * i01] + w00 * dIxy[2 * i00];
        out[1] = v;
        v = w11 * dIxy[2 * i11 + 1] + w10 * dIxy[2 * i10 + 1] + w01 * dIxy[2 * i01 + 1] + w00 * dIxy[2 * i00 + 1];
        out[2] = v;
    }
    return 1;
}

int orc_pixel_intensity(const unsigned char *I_ref, const float *dIxy_ref, int H, int W,
                        const double R[4], const double t[3], double D,
                        double fx, double fy, double cx, double cy,
                        double cur_x, double cur_y, double *intensity, double *jac)
{ /* compute_pixel_intensity.h:91-209 */
    const double qx = R[0], qy = R[1], qz = R[2], qw = R[3];
    const double x = t[0], y = t[1], z = t[2];

    double x_hat = (cur_x - cx) / fx;
    double y_hat = (cur_y - cy) / fy;
    const double z_hat = 1. / (double)sqrtf((float)(1. + x_hat * x_hat + y_hat * y_hat)); /* A4 */
    x_hat *= z_hat;
    y_hat *= z_hat;

    const double lambda = 2. * x_hat * (qx * qz - qw * qy) +
                          2. * y_hat * (qx * qw + qy * qz) +
                          z_hat * (qw * qw - qx * qx - qy * qy + qz * qz);
    const double s = (D - z) / lambda;
    const double pc[3] = { s * x_hat, s * y_hat, s * z_hat };
    double pr[3];
    orc_quat_rotate(R, pc, pr);
    const double Px = pr[0] + x, Py = pr[1] + y, Pz = pr[2] + z;

    const double iz = 1. / (Pz + 1e-8); /* A7 */
    const double p2x = Px * iz, p2y = Py * iz;
    const double rx = fx * p2x + cx, ry = fy * p2y + cy;

    double IdI[3] = { 0, 0, 0 };
    if (!orc_bilinear(I_ref, dIxy_ref, H, W, rx, ry, IdI)) return 0;
    *intensity = IdI[0];

    if (jac) {
        const double T0 = qx * x_hat + qy * y_hat + qz * z_hat;
        const double T1 = qy * x_hat - qw * z_hat - qx * y_hat;
        const double T2 = qw * y_hat - qx * z_hat + qz * x_hat;
        const double T3 = qw * x_hat + qy * z_hat - qz * y_hat;
        const double T4 = qw * z_hat + qx * y_hat - qy * x_hat;

        const double C1 = 1. / (2. * (x_hat * (-qw * qy + qx * qz) + y_hat * (qw * qx + qy * qz)) +
                                z_hat * (qw * qw - qx * qx - qy * qy + qz * qz));

        const double K = -2. * (qw * qz - qx * qy);
        const double L = 2. * (qw * qy + qx * qz);
        const double M = 2. * (qw * qz + qx * qy);
        const double N = -2. * (qw * qx - qy * qz);
        const double P = -2. * (qw * qy - qx * qz);
        const double Q = 2. * (qw * qx + qy * qz);

        const double R0 = x_hat * (qw * qw + qx * qx - qy * qy - qz * qz) + y_hat * K + z_hat * L;
        const double R1 = x_hat * M + y_hat * (qw * qw - qx * qx + qy * qy - qz * qz) + z_hat * N;
        const double R2 = x_hat * P + y_hat * Q + z_hat * (qw * qw - qx * qx - qy * qy + qz * qz);

        const double Dz = D - z;
        const double dPx_dqx = 2. * Dz * C1 * (T0 - T2 * C1 * R0);
        const double dPy_dqx = 2. * Dz * C1 * (T1 - T2 * C1 * R1);
        const double dPz_dqx = 2. * Dz * C1 * (T2 - T2 * C1 * R2);

        const double dPx_dqy = 2. * Dz * C1 * (T4 + T3 * C1 * R0);
        const double dPy_dqy = 2. * Dz * C1 * (T0 + T3 * C1 * R1);
        const double dPz_dqy = 2. * Dz * C1 * (T3 * C1 * R2 - T3);

        const double dPx_dqz = -2. * Dz * C1 * (T2 + T0 * C1 * R0);
        const double dPy_dqz = 2. * Dz * C1 * (T3 - T0 * C1 * R1);
        const double dPz_dqz = 2. * Dz * C1 * (T0 - T0 * C1 * R2);

        const double dPx_dqw = 2. * Dz * C1 * (T3 - T4 * C1 * R0);
        const double dPy_dqw = 2. * Dz * C1 * (T2 - T4 * C1 * R1);
        const double dPz_dqw = 2. * Dz * C1 * (T4 - T4 * C1 * R2);

        const double dI_dPx = IdI[1] * iz * fx;
        const double dI_dPy = IdI[2] * iz * fy;
        const double dI_dPz = -iz * iz * (IdI[1] * Px * fx + IdI[2] * Py * fy);

        jac[0] = dI_dPx;
        jac[1] = dI_dPy;
        jac[2] = dI_dPz * (1. - R2 * C1) - dI_dPx * R0 * C1 - dI_dPy * R1 * C1;
        jac[3] = dI_dPx * dPx_dqx + dI_dPy * dPy_dqx + dI_dPz * dPz_dqx;
        jac[4] = dI_dPx * dPx_dqy + dI_dPy * dPy_dqy + dI_dPz * dPz_dqy;
        jac[5] = dI_dPx * dPx_dqz + dI_dPy * dPy_dqz + dI_dPz * dPz_dqz;
        jac[6] = dI_dPx * dPx_dqw + dI_dPy * dPy_dqw + dI_dPz * dPz_dqw;
    }
    return 1;
}

/* ------------------------------------------------------------------------ */
/* reduction.h:13-55 -- power-of-two sizes use the pairwise tree of the       */
/* shared-memory reduce(); other sizes (where the reference silently drops   */
/* elements, quirk A10) are defined as the plain ascending sum.              */
/* ------------------------------------------------------------------------ */
static double tree_sum(double *buf, int n)
{
    if (n <= 0) return 0.0;
    if ((n & (n - 1)) == 0) {
        for (int s = n / 2; s >= 1; s /= 2)
            for (int i = 0; i < s; ++i) buf[i] = buf[i] + buf[i + s];
        return buf[0];
    }
    double acc = 0.0;
    for (int i = 0; i < n; ++i) acc += buf[i];
    return acc;
}

/* ------------------------------------------------------------------------ */
/* stage 1: virtual camera poses (compute_virtual_camera_poses.cu:9-110)     */
/* ------------------------------------------------------------------------ */
void orc_compute_virtual_camera_poses(int S, int F, const double *cap, const double *exp_t,
                                      int k, double t0, double dt,
                                      const double *knots_t, const double *knots_R,
                                      double *poses, double *J_t, double *J_R, int *start_idx)
{
    for (int f = 0; f < F; ++f)
        for (int i = 0; i < S; ++i) {
            const int v = f * S + i;
            const double t_cap = cap[f], t_mu = exp_t[f];
            const double t = t_cap - t_mu * 0.5 + i * t_mu / (S - 1 + 1e-8); /* A1 (:33) */
            int idx; double u;
            orc_spline_segment(t, t0, dt, &idx, &u);
            if (start_idx) start_idx[v] = idx;
            double p[3], q[4];
            double *jt = J_t ? J_t + (size_t)v * 9 * k : NULL;
            double *jr = J_R ? J_R + (size_t)v * 12 * k : NULL;
            if (k == 2) {
                orc_c2_vec3(knots_t + idx * 3, u, p, jt);
                orc_c2_rot3(knots_R + idx * 4, u, q, jr);
            } else {
                orc_c4_vec3(knots_t + idx * 3, u, p, jt);
                orc_c4_rot3(knots_R + idx * 4, u, q, jr);
            }
            double *o = poses + (size_t)v * 7;
            o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
            o[3] = q[0]; o[4] = q[1]; o[5] = q[2]; o[6] = q[3];
        }
}

/* ------------------------------------------------------------------------ */
/* stage 2: patch centres (compute_local_patches_xy.cu:9-50)                 */
/* ------------------------------------------------------------------------ */
static void patch_centre(const double *pose, double kx, double ky, double kz,
                         const double intr[4], double out[2])
{
    double P3dr[3];
    P3dr[0] = kz * (kx - intr[2]) / intr[0];
    P3dr[1] = kz * (ky - intr[3]) / intr[1];
    P3dr[2] = kz;
    const double t_c2r[3] = { pose[0], pose[1], pose[2] };
    const double R_c2r[4] = { pose[3], pose[4], pose[5], pose[6] };
    double R_r2c[4], rt[3], rp[3];
    quat_conj(R_c2r, R_r2c);
    orc_quat_rotate(R_r2c, t_c2r, rt);
    const double t_r2c[3] = { -rt[0], -rt[1], -rt[2] };
    orc_quat_rotate(R_r2c, P3dr, rp);
    const double X = rp[0] + t_r2c[0], Y = rp[1] + t_r2c[1], Z = rp[2] + t_r2c[2];
    out[0] = X / Z * intr[0] + intr[2];
    out[1] = Y / Z * intr[1] + intr[3];
}

void orc_compute_local_patches_xy(int S, int F, const double *poses,
                                  const double *kp_xy, const double *kp_z, int K,
                                  const double intr[4], double *centres)
{
    for (int f = 0; f < F; ++f) {
        const double *pose = poses + (size_t)(f * S + S / 2) * 7; /* A1 (:26) */
        for (int i = 0; i < K; ++i)
            patch_centre(pose, kp_xy[2 * i], kp_xy[2 * i + 1], kp_z[i], intr,
                         centres + ((size_t)f * K + i) * 2);
    }
}

/* ------------------------------------------------------------------------ */
/* stage 3: per-pixel residual + 1x6k Jacobian                               */
/* (compute_hessian_gradients_cost.cu:23-156).  A9: a pixel is valid iff the */
/* current pixel and ALL S warped samples are in bounds; else r = 0, J = 0.  */
/* ------------------------------------------------------------------------ */
/* Instrument of the long-horizon parity runs (tools/long_horizon.py): how close the evaluations since the last reset came to a
 * DISCONTINUITY of the objective -- [0] the smallest distance of a truncated pixel coordinate (A3, :69-70) from the next integer,
 * in pixels; [1] the smallest | |c - mu| - chi sigma | / (chi sigma) of the outlier test (blur_aware_direct_tracker.cpp:639-699).
 * A difference of that size between two implementations' inputs flips the pixel / the flag.  Single-threaded use only. */
static double g_margins[2] = { 1e300, 1e300 };
void orc_margins_reset(void) { g_margins[0] = g_margins[1] = 1e300; }
void orc_margins_get(double out[2]) { out[0] = g_margins[0]; out[1] = g_margins[1]; }

static int pixel_row(const unsigned char *I_ref, const float *dIxy_ref,
                     const unsigned char *I_cur, int S, const double *poses_f, int k,
                     const double *Jt_f, const double *JR_f, double cxp, double cyp, double z,
                     int dx, int dy, const double intr[4], int H, int W,
                     double *residual, double *Jrow /* 6k or NULL */, double *scratch /* S*(6k+1) */)
{
    const int n6k = 6 * k;
    const int px = (int)(cxp + dx); /* A3 (:69-70) */
    const int py = (int)(cyp + dy);
    {
        const double fx_ = cxp + dx, fy_ = cyp + dy;
        const double mx_ = fabs(fx_ - nearbyint(fx_)), my_ = fabs(fy_ - nearbyint(fy_));
        if (mx_ < g_margins[0]) g_margins[0] = mx_;
        if (my_ < g_margins[0]) g_margins[0] = my_;
    }
    *residual = 0.0;
    if (Jrow) memset(Jrow, 0, sizeof(double) * n6k);
    if (px < 0 || px > W - 1 || py < 0 || py > H - 1) return 0;

    double *ints = scratch;               /* S intensities */
    double *chain = scratch + S;          /* S x 6k per-sample chained rows */
    for (int s = 0; s < S; ++s) {
        const double *pose = poses_f + (size_t)s * 7;
        double j7[7], val;
        if (!orc_pixel_intensity(I_ref, dIxy_ref, H, W, pose + 3, pose, z, intr[0], intr[1], intr[2], intr[3],
                                 (double)px, (double)py, &val, Jrow ? j7 : NULL))
            return 0;
        ints[s] = val;
        if (Jrow) { /* :136-142: 1x3 * 3x3k and 1x4 * 4x3k */
            double *c = chain + (size_t)s * n6k;
            mm(j7, 1, 3, Jt_f + (size_t)s * 9 * k, 3 * k, c, 0, 0, 3 * k, 0);
            mm(j7 + 3, 1, 4, JR_f + (size_t)s * 12 * k, 3 * k, c, 0, 3 * k, 3 * k, 0);
        }
    }
    const double sum = tree_sum(ints, S);
    const double Icur = (double)I_cur[py * W + px];
    *residual = sum / (double)(float)S - Icur; /* A8 (:120) */
    if (Jrow) {
        double *col = ints; /* reuse */
        for (int i = 0; i < n6k; ++i) {
            for (int s = 0; s < S; ++s) col[s] = chain[(size_t)s * n6k + i];
            Jrow[i] = tree_sum(col, S) / (double)(float)S; /* :145-153 */
        }
    }
    return 1;
}

void orc_compute_pixel_jacobian_residual(const unsigned char *I_ref, const float *dIxy_ref,
                                         const unsigned char *const *I_cur, int S, int F,
                                         const double *poses, int k,
                                         const double *J_t, const double *J_R,
                                         const double *centres, const double *kp_z, int K,
                                         const int *pattern, int P,
                                         const double intr[4], int H, int W,
                                         double *residuals, double *jacobians)
{
    const int n6k = 6 * k;
    double *scratch = (double *)malloc(sizeof(double) * (size_t)S * (n6k + 1));
    for (int f = 0; f < F; ++f)
        for (int i = 0; i < K; ++i)
            for (int p = 0; p < P; ++p) {
                const size_t g = ((size_t)f * K + i) * P + p;
                pixel_row(I_ref, dIxy_ref, I_cur[f], S, poses + (size_t)f * S * 7, k,
                          J_t ? J_t + (size_t)f * S * 9 * k : NULL,
                          J_R ? J_R + (size_t)f * S * 12 * k : NULL,
                          centres[((size_t)f * K + i) * 2], centres[((size_t)f * K + i) * 2 + 1], kp_z[i],
                          pattern[2 * p], pattern[2 * p + 1], intr, H, W,
                          &residuals[g], jacobians ? jacobians + g * n6k : NULL, scratch);
            }
    free(scratch);
}

/* ------------------------------------------------------------------------ */
/* stage 4: per-patch packed [cost | g | upper(H)]                            */
/* (compute_hessian_gradients_cost.cu:165-239)                               */
/* ------------------------------------------------------------------------ */
static void huber(double r, double a, double *w, double *rho)
{ /* :189-199, A11 */
    const double aa = a * a;
    const double x = 0.5 * r * r;
    *w = 1.; *rho = x;
    if (x > aa) {
        *w = (double)sqrtf((float)(a / ((double)sqrtf((float)x) + 1e-8)));
        *rho = 2 * a * (double)sqrtf((float)x) - aa;
    }
}

static void patch_block(int P, int k, const double *res, const double *jac, double huber_a,
                        double inv, double *block, double *rows /* P*(6k+1) */, double *buf /* P */)
{
    const int ndim = 6 * k + 1;
    double rho_buf[1024];
    for (int p = 0; p < P; ++p) {
        double w, rho;
        huber(res[p], huber_a, &w, &rho);
        rho_buf[p] = rho;
        rows[(size_t)p * ndim] = w * res[p];
        if (jac)
            for (int i = 0; i < 6 * k; ++i) rows[(size_t)p * ndim + 1 + i] = w * jac[(size_t)p * 6 * k + i];
    }
    if (jac) {
        int e = 0;
        for (int i = 0; i < ndim; ++i)
            for (int j = i; j < ndim; ++j) {
                for (int p = 0; p < P; ++p) buf[p] = rows[(size_t)p * ndim + i] * rows[(size_t)p * ndim + j];
                block[e++] = tree_sum(buf, P) * inv;
            }
    }
    for (int p = 0; p < P; ++p) buf[p] = rho_buf[p];
    block[0] = tree_sum(buf, P) * inv; /* A12 (:232-238) */
}

void orc_compute_patch_cost_gradient_hessian(int F, int K, int P, int k,
                                             const double *residuals, const double *jacobians,
                                             double huber_a, double inv, double *patch_blocks)
{
    const int ndim = 6 * k + 1, E = ndim * (ndim + 1) / 2;
    double *rows = (double *)malloc(sizeof(double) * (size_t)P * ndim);
    double *buf = (double *)malloc(sizeof(double) * (size_t)(P > 0 ? P : 1));
    for (size_t pi = 0; pi < (size_t)F * K; ++pi)
        patch_block(P, k, residuals + pi * P, jacobians ? jacobians + pi * P * 6 * k : NULL,
                    huber_a, inv, patch_blocks + pi * E, rows, buf);
    free(rows); free(buf);
}

/* ------------------------------------------------------------------------ */
/* stage 5: per-frame sums (compute_hessian_gradients_cost.cu:247-283):      */
/* 256 strided lanes, then the 256-wide tree                                 */
/* ------------------------------------------------------------------------ */
void orc_compute_frame_cost_gradient_hessian(int F, int K, int k, const double *patch_blocks,
                                             int eval_gh, const unsigned char *outlier,
                                             double *frame_blocks)
{
    const int ndim = 6 * k + 1, E = ndim * (ndim + 1) / 2;
    const int ne = eval_gh ? E : 1;
    double lanes[256];
    for (int f = 0; f < F; ++f)
        for (int e = 0; e < ne; ++e) {
            for (int t = 0; t < 256; ++t) {
                double sum = 0;
                for (int i = t; i < K; i += 256) {
                    if (outlier && outlier[i] == 1) continue;
                    sum += patch_blocks[((size_t)f * K + i) * E + e];
                }
                lanes[t] = sum;
            }
            frame_blocks[(size_t)f * E + e] = tree_sum(lanes, 256);
        }
}

/* ------------------------------------------------------------------------ */
/* stage 6: merge into the global 6N x 6N system                             */
/* (merge_hessian_gradient_cost.cpp:8-87)                                    */
/* ------------------------------------------------------------------------ */
void orc_merge_hessian_gradient_cost(int F, int k, const double *frame_blocks, const int *start_idx,
                                     int N, double *total_cost, double *H, double *g)
{
    const int ndim = 6 * k + 1, E = ndim * (ndim + 1) / 2;
    const int n = 6 * N;
    *total_cost = 0;
    if (H) { memset(H, 0, sizeof(double) * (size_t)n * n); memset(g, 0, sizeof(double) * n); }
    for (int i = 0; i < F; ++i) {
        const int st = start_idx[i];
        const double *fb = frame_blocks + (size_t)i * E;
        *total_cost += fb[0];
        if (!H) continue;
        int shift = st * 3;
        for (int j = 0; j < 3 * k; ++j) g[shift++] += fb[j + 1];
        shift = (N + st) * 3;
        for (int j = 3 * k; j < 6 * k; ++j) g[shift++] += fb[j + 1];
        const int off0 = st * 3, off1 = (N + st) * 3;
        const double *dp = fb + ndim;
        for (int j = 0; j < ndim - 1; ++j) {
            const int r = j + (j < 3 * k ? off0 : off1 - 3 * k);
            for (int c_ = j; c_ < ndim - 1; ++c_, ++dp) {
                const int c = c_ + (c_ < 3 * k ? off0 : off1 - 3 * k);
                const double v = *dp;
                H[(size_t)c * n + r] += v; /* column-major H(r,c) */
                if (c == r) continue;
                H[(size_t)r * n + c] += v;
            }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* one evaluation (spline_update_step.cpp:97-349)                            */
/* ------------------------------------------------------------------------ */
void orc_evaluate(const orc_problem *p, double *patch_blocks, double *frame_blocks,
                  double *total_cost, double *H, double *g)
{
    const int S = p->S, F = p->F, K = p->K, P = p->P, k = p->k;
    const int with_j = H != NULL;
    const int num_residuals = (K - p->num_bad) * F * P; /* A13 (:116-117) */
    const double inv = num_residuals > 0 ? 1.0 / num_residuals : 0.0; /* empty problem (reference: division by zero) */
    double *poses = (double *)malloc(sizeof(double) * (size_t)F * S * 7);
    double *Jt = with_j ? (double *)malloc(sizeof(double) * (size_t)F * S * 9 * k) : NULL;
    double *JR = with_j ? (double *)malloc(sizeof(double) * (size_t)F * S * 12 * k) : NULL;
    double *centres = (double *)malloc(sizeof(double) * (size_t)F * K * 2);
    double *res = (double *)malloc(sizeof(double) * (size_t)F * K * P);
    double *jac = with_j ? (double *)malloc(sizeof(double) * (size_t)F * K * P * 6 * k) : NULL;

    orc_compute_virtual_camera_poses(S, F, p->cap, p->exp_t, k, p->t0, p->dt, p->knots_t, p->knots_R,
                                     poses, Jt, JR, NULL);
    orc_compute_local_patches_xy(S, F, poses, p->kp_xy, p->kp_z, K, p->intr, centres);
    orc_compute_pixel_jacobian_residual(p->ref_img, p->ref_dIxy, p->cur_imgs, S, F, poses, k, Jt, JR,
                                        centres, p->kp_z, K, p->pattern, P, p->intr, p->H, p->W, res, jac);
    orc_compute_patch_cost_gradient_hessian(F, K, P, k, res, jac, p->huber_a, inv, patch_blocks);
    orc_compute_frame_cost_gradient_hessian(F, K, k, patch_blocks, with_j, p->outlier, frame_blocks);
    orc_merge_hessian_gradient_cost(F, k, frame_blocks, p->start_idx, p->N, total_cost, H, g);

    free(poses); free(Jt); free(JR); free(centres); free(res); free(jac);
}

/* Fused / threaded variant for the cpu_baseline leg: identical per-patch
 * arithmetic, patch blocks summed per thread chunk in keypoint order (no
 * 256-lane tree), so results agree with orc_evaluate to summation rounding. */
void orc_evaluate_fast(const orc_problem *p, int num_threads, double *frame_blocks,
                       double *total_cost, double *H, double *g)
{
    const int S = p->S, F = p->F, K = p->K, P = p->P, k = p->k;
    const int n6k = 6 * k, ndim = n6k + 1, E = ndim * (ndim + 1) / 2;
    const int with_j = H != NULL;
    const double inv = (K - p->num_bad) * F * P > 0 ? 1.0 / ((K - p->num_bad) * F * P) : 0.0;
    if (num_threads < 1) num_threads = 1;
    double *poses = (double *)malloc(sizeof(double) * (size_t)F * S * 7);
    double *Jt = with_j ? (double *)malloc(sizeof(double) * (size_t)F * S * 9 * k) : NULL;
    double *JR = with_j ? (double *)malloc(sizeof(double) * (size_t)F * S * 12 * k) : NULL;
    double *part = (double *)calloc((size_t)num_threads * F * E, sizeof(double));
    orc_compute_virtual_camera_poses(S, F, p->cap, p->exp_t, k, p->t0, p->dt, p->knots_t, p->knots_R,
                                     poses, Jt, JR, NULL);
#ifdef _OPENMP
#pragma omp parallel num_threads(num_threads)
#endif
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        const int tid = 0, nt = 1;
#endif
        double *scratch = (double *)malloc(sizeof(double) * (size_t)S * (n6k + 1));
        double *rows = (double *)malloc(sizeof(double) * (size_t)P * ndim);
        double *buf = (double *)malloc(sizeof(double) * (size_t)P);
        double *res = (double *)malloc(sizeof(double) * (size_t)P);
        double *jac = (double *)malloc(sizeof(double) * (size_t)P * n6k);
        double *block = (double *)malloc(sizeof(double) * E);
        const int lo = (int)((long long)K * tid / nt), hi = (int)((long long)K * (tid + 1) / nt);
        for (int f = 0; f < F; ++f) {
            double *acc = part + ((size_t)tid * F + f) * E;
            const double *pose_mid = poses + (size_t)(f * S + S / 2) * 7;
            for (int i = lo; i < hi; ++i) {
                double c[2];
                patch_centre(pose_mid, p->kp_xy[2 * i], p->kp_xy[2 * i + 1], p->kp_z[i], p->intr, c);
                for (int q = 0; q < P; ++q)
                    pixel_row(p->ref_img, p->ref_dIxy, p->cur_imgs[f], S, poses + (size_t)f * S * 7, k,
                              with_j ? Jt + (size_t)f * S * 9 * k : NULL, with_j ? JR + (size_t)f * S * 12 * k : NULL,
                              c[0], c[1], p->kp_z[i], p->pattern[2 * q], p->pattern[2 * q + 1], p->intr,
                              p->H, p->W, &res[q], with_j ? jac + (size_t)q * n6k : NULL, scratch);
                patch_block(P, k, res, with_j ? jac : NULL, p->huber_a, inv, block, rows, buf);
                if (p->outlier && p->outlier[i] == 1) continue;
                if (with_j) for (int e = 0; e < E; ++e) acc[e] += block[e];
                else acc[0] += block[0];
            }
        }
        free(scratch); free(rows); free(buf); free(res); free(jac); free(block);
    }
    for (int f = 0; f < F; ++f)
        for (int e = 0; e < E; ++e) {
            double s = 0;
            for (int t = 0; t < num_threads; ++t) s += part[((size_t)t * F + f) * E + e];
            frame_blocks[(size_t)f * E + e] = s;
        }
    orc_merge_hessian_gradient_cost(F, k, frame_blocks, p->start_idx, p->N, total_cost, H, g);
    free(poses); free(Jt); free(JR); free(part);
}

/* Valid pixels per frame of one evaluation: a pixel counts iff pixel_row accepts it -- its integer location and all S warps in
 * bounds (compute_pixel_intensity.h:25-72, compute_pixel_jacobian_residual.cu:69-120).  What the product reports in
 * mbavo_eval_batch's d_valid; test infrastructure for the full-size comparisons (exact counts). */
void orc_count_valid(const orc_problem *p, int num_threads, double *valid /* F */)
{
    const int S = p->S, F = p->F, K = p->K, P = p->P, k = p->k;
    if (num_threads < 1) num_threads = 1;
    double *poses = (double *)malloc(sizeof(double) * (size_t)F * S * 7);
    long long *part = (long long *)calloc((size_t)num_threads * F, sizeof(long long));
    orc_compute_virtual_camera_poses(S, F, p->cap, p->exp_t, k, p->t0, p->dt, p->knots_t, p->knots_R, poses, NULL, NULL, NULL);
#ifdef _OPENMP
#pragma omp parallel num_threads(num_threads)
#endif
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        const int tid = 0, nt = 1;
#endif
        double *scratch = (double *)malloc(sizeof(double) * (size_t)S * (6 * k + 1));
        const int lo = (int)((long long)K * tid / nt), hi = (int)((long long)K * (tid + 1) / nt);
        for (int f = 0; f < F; ++f) {
            const double *pose_mid = poses + (size_t)(f * S + S / 2) * 7;
            long long n = 0;
            for (int i = lo; i < hi; ++i) {
                double c[2], r;
                patch_centre(pose_mid, p->kp_xy[2 * i], p->kp_xy[2 * i + 1], p->kp_z[i], p->intr, c);
                for (int q = 0; q < P; ++q)
                    n += pixel_row(p->ref_img, p->ref_dIxy, p->cur_imgs[f], S, poses + (size_t)f * S * 7, k, NULL, NULL, c[0], c[1], p->kp_z[i],
                                   p->pattern[2 * q], p->pattern[2 * q + 1], p->intr, p->H, p->W, &r, NULL, scratch);
            }
            part[(size_t)tid * F + f] = n;
        }
        free(scratch);
    }
    for (int f = 0; f < F; ++f) {
        long long n = 0;
        for (int t = 0; t < num_threads; ++t) n += part[(size_t)t * F + f];
        valid[f] = (double)n;
    }
    free(poses); free(part);
}

/* ------------------------------------------------------------------------ */
/* solve_normal_equation.h:10-35.  Eigen (3.3.x, version unpinned by the     */
/* reference) is absent: JacobiSVD (two-sided Jacobi with the real 2x2       */
/* kernel, JacobiSVD.h / Jacobi.h / RealSvd2x2.h) and LDLT (diagonal         */
/* pivoting, LDLT.h) restated from the published sources.  PARITY UNPINNED.   */
/* All matrices column-major n x n.                                           */
/* ------------------------------------------------------------------------ */
typedef struct { double c, s; } jrot; /* J = [c s; -s c] */

static void rot_rows(double *M, int n, int p, int q, jrot j)
{ /* M <- J * M on rows p,q */
    for (int c = 0; c < n; ++c) {
        const double x = M[(size_t)c * n + p], y = M[(size_t)c * n + q];
        M[(size_t)c * n + p] = j.c * x + j.s * y;
        M[(size_t)c * n + q] = -j.s * x + j.c * y;
    }
}
static void rot_cols(double *M, int n, int p, int q, jrot j)
{ /* M <- M * J on columns p,q */
    for (int r = 0; r < n; ++r) {
        const double x = M[(size_t)p * n + r], y = M[(size_t)q * n + r];
        M[(size_t)p * n + r] = j.c * x - j.s * y;
        M[(size_t)q * n + r] = j.s * x + j.c * y;
    }
}
static jrot make_jacobi(double x, double y, double z)
{ /* Jacobi.h makeJacobi(real x, y, z) for the selfadjoint [x y; y z] */
    jrot j;
    const double deno = 2.0 * fabs(y);
    if (deno < DBL_MIN) { j.c = 1; j.s = 0; return j; }
    const double tau = (x - z) / deno;
    const double w = sqrt(tau * tau + 1.0);
    const double t = tau > 0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
    const double sign_t = t > 0 ? 1.0 : -1.0;
    const double n = 1.0 / sqrt(t * t + 1.0);
    j.s = -sign_t * (y / fabs(y)) * fabs(t) * n;
    j.c = n;
    return j;
}

static int svd_solve(const double *A, const double *b, int n, double *x)
{
    double *W = (double *)malloc(sizeof(double) * (size_t)n * n);
    double *U = (double *)calloc((size_t)n * n, sizeof(double));
    double *V = (double *)calloc((size_t)n * n, sizeof(double));
    double *sv = (double *)malloc(sizeof(double) * n);
    double scale = 0;
    for (size_t i = 0; i < (size_t)n * n; ++i) if (fabs(A[i]) > scale) scale = fabs(A[i]);
    if (!(scale > 0) || !isfinite(scale)) scale = 1;
    for (size_t i = 0; i < (size_t)n * n; ++i) W[i] = A[i] / scale;
    for (int i = 0; i < n; ++i) { U[(size_t)i * n + i] = 1; V[(size_t)i * n + i] = 1; }
    const double precision = 2.0 * DBL_EPSILON, tiny = DBL_MIN;
    double max_diag = 0;
    for (int i = 0; i < n; ++i) if (fabs(W[(size_t)i * n + i]) > max_diag) max_diag = fabs(W[(size_t)i * n + i]);
    int finished = 0, sweeps = 0;
    while (!finished && sweeps++ < 1000) {
        finished = 1;
        for (int p = 1; p < n; ++p)
            for (int q = 0; q < p; ++q) {
                double thr = precision * max_diag; if (thr < tiny) thr = tiny;
                const double wpq = W[(size_t)q * n + p], wqp = W[(size_t)p * n + q];
                if (fabs(wpq) > thr || fabs(wqp) > thr) {
                    finished = 0;
                    /* RealSvd2x2.h real_2x2_jacobi_svd on [W(p,p) W(p,q); W(q,p) W(q,q)] */
                    double m00 = W[(size_t)p * n + p], m01 = wpq, m10 = wqp, m11 = W[(size_t)q * n + q];
                    jrot rot1;
                    const double t = m00 + m11, d = m10 - m01;
                    if (fabs(d) < tiny) { rot1.s = 0; rot1.c = 1; }
                    else { const double u = t / d; const double tmp = sqrt(1.0 + u * u); rot1.s = 1.0 / tmp; rot1.c = u / tmp; }
                    const double a00 = rot1.c * m00 + rot1.s * m10, a01 = rot1.c * m01 + rot1.s * m11;
                    const double a11 = -rot1.s * m01 + rot1.c * m11;
                    const jrot jr = make_jacobi(a00, a01, a11);
                    jrot jl; /* rot1 * transpose(jr) */
                    jl.c = rot1.c * jr.c - rot1.s * (-jr.s);
                    jl.s = rot1.c * (-jr.s) + rot1.s * jr.c;
                    rot_rows(W, n, p, q, jl);
                    { jrot jlt = { jl.c, -jl.s }; rot_cols(U, n, p, q, jlt); }
                    rot_cols(W, n, p, q, jr);
                    rot_cols(V, n, p, q, jr);
                    const double dp = fabs(W[(size_t)p * n + p]), dq = fabs(W[(size_t)q * n + q]);
                    if (dp > max_diag) max_diag = dp;
                    if (dq > max_diag) max_diag = dq;
                }
            }
    }
    for (int i = 0; i < n; ++i) {
        const double a = W[(size_t)i * n + i];
        sv[i] = fabs(a);
        if (a < 0) for (int r = 0; r < n; ++r) U[(size_t)i * n + r] = -U[(size_t)i * n + r];
        sv[i] *= scale;
    }
    int nonzero = n;
    for (int i = 0; i < n; ++i) { /* selection sort, descending (JacobiSVD.h tail) */
        int pos = i; double mx = sv[i];
        for (int j = i + 1; j < n; ++j) if (sv[j] > mx) { mx = sv[j]; pos = j; }
        if (mx == 0) { nonzero = i; break; }
        if (pos != i) {
            double ts = sv[i]; sv[i] = sv[pos]; sv[pos] = ts;
            for (int r = 0; r < n; ++r) {
                double tu = U[(size_t)i * n + r]; U[(size_t)i * n + r] = U[(size_t)pos * n + r]; U[(size_t)pos * n + r] = tu;
                double tv = V[(size_t)i * n + r]; V[(size_t)i * n + r] = V[(size_t)pos * n + r]; V[(size_t)pos * n + r] = tv;
            }
        }
    }
    /* SVDBase::rank() with the default threshold diagSize*epsilon, then V S^-1 U^T b */
    int rank = 0;
    if (n > 0) {
        double pthr = sv[0] * ((double)(n > 1 ? n : 1) * DBL_EPSILON);
        if (pthr < tiny) pthr = tiny;
        int i = nonzero - 1;
        while (i >= 0 && sv[i] < pthr) --i;
        rank = i + 1;
    }
    for (int r = 0; r < n; ++r) x[r] = 0;
    for (int i = 0; i < rank; ++i) {
        double dot = 0;
        for (int r = 0; r < n; ++r) dot += U[(size_t)i * n + r] * b[r];
        dot = dot / sv[i];
        for (int r = 0; r < n; ++r) x[r] += V[(size_t)i * n + r] * dot;
    }
    free(W); free(U); free(V); free(sv);
    return rank;
}

static int ldlt_solve(const double *A, const double *b, int n, double *x)
{ /* LDLT.h ldlt_inplace<Lower>::unblocked + solve; lower triangle of A is read */
    double *M = (double *)malloc(sizeof(double) * (size_t)n * n);
    int *perm = (int *)malloc(sizeof(int) * n);
    double *tmp = (double *)malloc(sizeof(double) * n);
    memcpy(M, A, sizeof(double) * (size_t)n * n);
#define LM(r, c) M[(size_t)(c) * n + (r)]
    for (int kk = 0; kk < n; ++kk) {
        int big = kk; double bv = fabs(LM(kk, kk));
        for (int i = kk + 1; i < n; ++i) if (fabs(LM(i, i)) > bv) { bv = fabs(LM(i, i)); big = i; }
        perm[kk] = big;
        if (big != kk) { /* symmetric swap of rows/cols kk <-> big within the lower triangle */
            const int s = n - big - 1;
            for (int c = 0; c < kk; ++c) { double t = LM(kk, c); LM(kk, c) = LM(big, c); LM(big, c) = t; }
            for (int r = 0; r < s; ++r) { double t = LM(big + 1 + r, kk); LM(big + 1 + r, kk) = LM(big + 1 + r, big); LM(big + 1 + r, big) = t; }
            { double t = LM(kk, kk); LM(kk, kk) = LM(big, big); LM(big, big) = t; }
            for (int i = kk + 1; i < big; ++i) { double t = LM(i, kk); LM(i, kk) = LM(big, i); LM(big, i) = t; }
        }
        const int rs = n - kk - 1;
        if (kk > 0) {
            for (int c = 0; c < kk; ++c) tmp[c] = LM(c, c) * LM(kk, c);
            double dot = 0; for (int c = 0; c < kk; ++c) dot += LM(kk, c) * tmp[c];
            LM(kk, kk) -= dot;
            for (int r = 0; r < rs; ++r) {
                double d2 = 0; for (int c = 0; c < kk; ++c) d2 += LM(kk + 1 + r, c) * tmp[c];
                LM(kk + 1 + r, kk) -= d2;
            }
        }
        const double akk = LM(kk, kk);
        if (fabs(akk) > 0) for (int r = 0; r < rs; ++r) LM(kk + 1 + r, kk) /= akk;
    }
    /* solve: x = P^T L^-T D^-1 L^-1 P b */
    for (int i = 0; i < n; ++i) x[i] = b[i];
    for (int i = 0; i < n; ++i) if (perm[i] != i) { double t = x[i]; x[i] = x[perm[i]]; x[perm[i]] = t; }
    for (int c = 0; c < n; ++c) for (int r = c + 1; r < n; ++r) x[r] -= LM(r, c) * x[c];
    for (int i = 0; i < n; ++i) { const double d = LM(i, i); x[i] = fabs(d) > DBL_MIN ? x[i] / d : 0.0; }
    for (int c = n - 1; c >= 0; --c) for (int r = c + 1; r < n; ++r) x[c] -= LM(r, c) * x[r];
    for (int i = n - 1; i >= 0; --i) if (perm[i] != i) { double t = x[i]; x[i] = x[perm[i]]; x[perm[i]] = t; }
#undef LM
    free(M); free(perm); free(tmp);
    return n;
}

int orc_solve_normal_equation(const double *A, const double *b, int n, int solver_type, double *x)
{
    int r;
    if (solver_type == 0) r = svd_solve(A, b, n, x);
    else if (solver_type == 1) r = ldlt_solve(A, b, n, x);
    else return -1;
    for (int i = 0; i < n; ++i) x[i] = -x[i]; /* solve_normal_equation.h:33 */
    return r;
}

/* ------------------------------------------------------------------------ */
/* LM strategy (levenberg_marquardt_strategy.cpp:9-45)                       */
/* ------------------------------------------------------------------------ */
void orc_lm_init(orc_lm *s) { s->radius = 1e4; s->min_radius = 10; s->max_radius = 1e32; s->decrease_factor = 2.0; }
void orc_lm_reset(orc_lm *s) { s->radius = 1e4; s->decrease_factor = 2.0; }
static double clampd(double v, double lo, double hi) { double m = hi < v ? hi : v; return m > lo ? m : lo; }
void orc_lm_accepted(orc_lm *s, double q)
{
    const double a = 1.0 / 3.0, b = 1.0 - pow(2.0 * q - 1.0, 3);
    s->radius = s->radius / (a > b ? a : b);
    s->radius = clampd(s->radius, s->min_radius, s->max_radius);
    s->decrease_factor = 2.0;
}
void orc_lm_rejected(orc_lm *s)
{
    s->radius = s->radius / s->decrease_factor;
    s->radius = clampd(s->radius, s->min_radius, s->max_radius);
    s->decrease_factor *= 2.0;
}

/* trust_region_step_evaluator.cpp:45-126 */
void orc_tr_init(orc_tr *e, int m) { memset(e, 0, sizeof(*e)); e->max_nonmono = m; }
void orc_tr_reset(orc_tr *e, double c)
{
    e->minimum_cost = c; e->current_cost = c; e->reference_cost = c; e->candidate_cost = c;
    e->acc_ref = 0; e->acc_cand = 0; e->num_nonmono = 0;
}
double orc_tr_quality(const orc_tr *e, double cost, double model_cost_change)
{
    if (cost >= DBL_MAX) return -DBL_MAX;
    const double rel = (e->current_cost - cost) / model_cost_change;
    const double hist = (e->reference_cost - cost) / (e->acc_ref + model_cost_change);
    return rel < hist ? hist : rel; /* std::max(rel, hist) as libstdc++ evaluates it: a NaN first argument comes back (:74) */
}
void orc_tr_accepted(orc_tr *e, double cost, double mcc)
{
    e->current_cost = cost;
    e->acc_cand += mcc;
    e->acc_ref += mcc;
    if (e->current_cost < e->minimum_cost) {
        e->minimum_cost = e->current_cost;
        e->num_nonmono = 0;
        e->candidate_cost = e->current_cost;
        e->acc_cand = 0;
    } else {
        ++e->num_nonmono;
        if (e->current_cost > e->candidate_cost) { e->candidate_cost = e->current_cost; e->acc_cand = 0; }
    }
    if (e->num_nonmono == e->max_nonmono) { e->reference_cost = e->candidate_cost; e->acc_ref = e->acc_cand; }
}

/* Spline.h:307-330 */
void orc_plus_t(const double *t, const double *d, int N, double *out)
{ for (int i = 0; i < 3 * N; ++i) out[i] = t[i] + d[i]; }
void orc_plus_R(const double *R, const double *d, int N, double *out)
{ /* R_i * SO3::exp(omega_i); Eigen quaternion product == Hamilton product */
    for (int i = 0; i < N; ++i) {
        double dq[4];
        orc_so3_exp(d + 3 * i, dq);
        const double *a = R + 4 * i;
        double *o = out + 4 * i;
        o[3] = a[3] * dq[3] - a[0] * dq[0] - a[1] * dq[1] - a[2] * dq[2];
        o[0] = a[3] * dq[0] + a[0] * dq[3] + a[1] * dq[2] - a[2] * dq[1];
        o[1] = a[3] * dq[1] + a[1] * dq[3] + a[2] * dq[0] - a[0] * dq[2];
        o[2] = a[3] * dq[2] + a[2] * dq[3] + a[0] * dq[1] - a[1] * dq[0];
    }
}

/* ------------------------------------------------------------------------ */
/* input producers                                                           */
/* ------------------------------------------------------------------------ */
void orc_pyramid_down_u8(const unsigned char *src, int H, int W, unsigned char *dst)
{ /* ImagePyramid.h:59-99: 2x2 box, T(0.25*sum) truncation; H,W = source size */
    const int Hl = H / 2, Wl = W / 2;
    for (int h = 0; h < Hl; ++h)
        for (int w = 0; w < Wl; ++w) {
            const float a = (float)src[(2 * h) * W + 2 * w], b = (float)src[(2 * h) * W + 2 * w + 1];
            const float c = (float)src[(2 * h + 1) * W + 2 * w], d = (float)src[(2 * h + 1) * W + 2 * w + 1];
            dst[h * Wl + w] = (unsigned char)(0.25 * (a + b + c + d));
        }
}

void orc_image_gradients_u8(const unsigned char *src, int H, int W, float *g, float *mag)
{ /* Gradient.h:16-75, C = 1 */
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t i = (size_t)y * W + x;
            if (x == 0 || y == 0 || x == W - 1 || y == H - 1) {
                g[2 * i] = 0; g[2 * i + 1] = 0;
                if (mag) mag[i] = 0;
                continue;
            }
            const float dx = (float)(0.5 * ((float)src[i + 1] - (float)src[i - 1]));
            const float dy = (float)(0.5 * ((float)src[i + W] - (float)src[i - W]));
            g[2 * i] = dx; g[2 * i + 1] = dy;
            if (mag) { float m = 0; m += sqrtf(dx * dx + dy * dy); mag[i] = m / 1; }
        }
}

/* ------------------------------------------------------------------------ */
/* synthetic data (generate_synthetic_data.cpp:127-214)                      */
/* ------------------------------------------------------------------------ */
void orc_warp_image(const unsigned char *ref, int H, int W, const double R[4], const double t[3],
                    double plane_depth, const double intr[4], unsigned char *out)
{
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            double v = 0;
            orc_pixel_intensity(ref, NULL, H, W, R, t, plane_depth, intr[0], intr[1], intr[2], intr[3],
                                (double)c, (double)r, &v, NULL);
            out[(size_t)r * W + c] = (unsigned char)v;
        }
}

static void spline_pose(int k, double t0, double dt, const double *kt, const double *kR, double t,
                        double p[3], double q[4])
{ /* Spline.h:222-281 GetPose without jacobians */
    int idx; double u;
    orc_spline_segment(t, t0, dt, &idx, &u);
    if (k == 2) { orc_c2_vec3(kt + idx * 3, u, p, NULL); orc_c2_rot3(kR + idx * 4, u, q, NULL); }
    else { orc_c4_vec3(kt + idx * 3, u, p, NULL); orc_c4_rot3(kR + idx * 4, u, q, NULL); }
}

void orc_synthesize_blur(const unsigned char *ref, int H, int W, double plane_depth,
                         const double intr[4], int k, double t0, double dt,
                         const double *knots_t, const double *knots_R,
                         double cap, double exp_t, int num_samples, unsigned char *out)
{
    const size_t n = (size_t)H * W;
    float *acc = (float *)calloc(n, sizeof(float));
    unsigned char *cur = (unsigned char *)malloc(n);
    for (int i = 0; i < num_samples; ++i) {
        const double t = cap - exp_t * 0.5 + i * exp_t / (num_samples - 1);
        double p[3], q[4];
        spline_pose(k, t0, dt, knots_t, knots_R, t, p, q);
        orc_warp_image(ref, H, W, q, p, plane_depth, intr, cur);
        for (size_t j = 0; j < n; ++j) acc[j] = acc[j] + (float)cur[j];
    }
    for (size_t j = 0; j < n; ++j) {
        /* I_internal /= num_samples; convertTo(CV_8UC1) == saturate_cast<uchar>(cvRound(v)) */
        const float v = acc[j] / (float)num_samples;
        long r = lrintf(v);
        out[j] = (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
    free(acc); free(cur);
}

/* ------------------------------------------------------------------------ */
/* LM loop over the pyramid (blur_aware_direct_tracker.cpp:544-924)          */
/* ------------------------------------------------------------------------ */
static int detect_outliers(const double *patch_blocks, int K, int E, double chi,
                           unsigned char *flags)
{ /* :639-699 (A14): only frame 0's K patches */
    double sum = 0; int cnt = 0;
    for (int i = 0; i < K; ++i) { const double c = patch_blocks[(size_t)i * E]; if (c < 1e-8) continue; sum += c; ++cnt; }
    const double mu = sum / cnt;
    double var = 0;
    for (int i = 0; i < K; ++i) { const double c = patch_blocks[(size_t)i * E]; if (c < 1e-8) continue; var += (c - mu) * (c - mu); }
    var = var / cnt;
    int n_out = 0;
    for (int i = 0; i < K; ++i) {
        const double c = patch_blocks[(size_t)i * E];
        if (fabs(c - mu) > chi * (double)sqrtf((float)var)) { flags[i] = 1; ++n_out; }
        {
            const double thr_ = chi * (double)sqrtf((float)var), m_ = fabs(fabs(c - mu) - thr_) / (thr_ > 0 ? thr_ : 1.0);
            if (m_ < g_margins[1]) g_margins[1] = m_;
        }
    }
    return n_out;
}

int orc_optimize_trajectory(const orc_track_opts *o, const orc_level *levels, int F,
                            const double *cap, const double *exp_t, double t0, double dt,
                            double *knots_t, double *knots_R, int N,
                            int *start_idx_out, double *final_cost,
                            orc_trace_rec *trace, int trace_cap)
{
    const int k = o->k, n = 6 * N;
    const int ndim = 6 * k + 1, E = ndim * (ndim + 1) / 2;
    int ntrace = 0;
    int *start_idx = (int *)malloc(sizeof(int) * F);
    for (int f = 0; f < F; ++f) { double u; orc_spline_segment(cap[f], t0, dt, &start_idx[f], &u); } /* :549-560 */
    if (start_idx_out) memcpy(start_idx_out, start_idx, sizeof(int) * F);
    double *H = (double *)malloc(sizeof(double) * (size_t)n * n);
    double *g = (double *)malloc(sizeof(double) * n);
    double *step = (double *)malloc(sizeof(double) * n);
    double *Hx = (double *)malloc(sizeof(double) * n);
    double *cand_t = (double *)malloc(sizeof(double) * 3 * N);
    double *cand_R = (double *)malloc(sizeof(double) * 4 * N);
    orc_lm lm; orc_lm_init(&lm);
    orc_tr tr; orc_tr_init(&tr, o->max_nonmono);
    double eval_cost = 0;

#define TRACE(lv_, it_, kind_, nout_, rad_, ec_, cc_, mc_, q_) do { if (trace && ntrace < trace_cap) { \
        orc_trace_rec *r_ = &trace[ntrace]; r_->level = lv_; r_->iter = it_; r_->kind = kind_; r_->num_outliers = nout_; \
        r_->radius = rad_; r_->eval_cost = ec_; r_->candidate_cost = cc_; r_->model_change = mc_; r_->quality = q_; } ++ntrace; } while (0)

    for (int li = 0; li < o->num_levels; ++li) {
        const int lv = o->num_levels - li - 1; /* A18 (:571-575) */
        const orc_level *L = &levels[lv];
        const int scale = 1 << lv;
        unsigned char *flags = (unsigned char *)calloc(L->K > 0 ? L->K : 1, 1);
        double *patch_blocks = (double *)calloc((size_t)F * L->K * E, sizeof(double));
        double *frame_blocks = (double *)calloc((size_t)F * E, sizeof(double));
        orc_problem p;
        memset(&p, 0, sizeof(p));
        p.S = L->S; p.F = F; p.K = L->K; p.P = L->P; p.k = k; p.N = N; p.H = L->H; p.W = L->W;
        p.ref_img = L->ref_img; p.ref_dIxy = L->ref_dIxy; p.cur_imgs = L->cur_imgs;
        p.kp_xy = L->kp_xy; p.kp_z = L->kp_z; p.pattern = L->pattern;
        p.outlier = flags; p.num_bad = 0;
        for (int a = 0; a < 4; ++a) p.intr[a] = o->intr[a] / scale; /* :766-770 */
        p.cap = cap; p.exp_t = exp_t; p.t0 = t0; p.dt = dt;
        p.knots_t = knots_t; p.knots_R = knots_R; p.start_idx = start_idx; p.huber_a = o->huber_k;

        double cand_cost = 0, model = 0, quality = 0;
        orc_evaluate(&p, patch_blocks, frame_blocks, &eval_cost, H, g); /* iteration 0 (:604) */
        orc_lm_reset(&lm);
        orc_tr_reset(&tr, eval_cost);
        TRACE(lv, 0, 0, 0, lm.radius, eval_cost, 0.0, 0.0, 0.0);

        int iter = 0; double abs_dec = 1e10;
        for (;;) {
            ++iter; /* :910-924 */
            if (iter > o->max_num_iterations) break;
            if (abs_dec < o->min_abs_cost_decrease) break;

            /* computeTrustRegionStep (:799-831), A15: damping accumulates in place */
            const double iradius = 1. / lm.radius;
            for (int i = 0; i < n; ++i) H[(size_t)i * n + i] += H[(size_t)i * n + i] * iradius;
            orc_solve_normal_equation(H, g, n, o->solver_type, step);
            double gx = 0; for (int i = 0; i < n; ++i) gx += g[i] * step[i];
            for (int r = 0; r < n; ++r) { double a = 0; for (int c = 0; c < n; ++c) a += H[(size_t)c * n + r] * step[c]; Hx[r] = a; }
            double xHx = 0; for (int i = 0; i < n; ++i) xHx += step[i] * Hx[i];
            model = -(gx + 0.5 * xHx);
            if (model < 0) { orc_lm_rejected(&lm); TRACE(lv, iter, 3, p.num_bad, lm.radius, eval_cost, 0.0, model, 0.0); continue; }

            /* computeCandidatePointAndEvaluateCost (:833-883) */
            orc_plus_t(knots_t, step, N, cand_t);
            orc_plus_R(knots_R, step + 3 * N, N, cand_R);
            p.knots_t = cand_t; p.knots_R = cand_R;
            orc_evaluate(&p, patch_blocks, frame_blocks, &cand_cost, NULL, NULL);
            p.knots_t = knots_t; p.knots_R = knots_R;

            abs_dec = eval_cost - cand_cost; /* A16 */
            quality = orc_tr_quality(&tr, cand_cost, model);
            if (quality > o->min_step_quality && cand_cost < eval_cost) { /* A17 */
                p.num_bad = detect_outliers(patch_blocks, L->K, E, o->max_chi_square_error, flags);
                memcpy(knots_t, cand_t, sizeof(double) * 3 * N);
                memcpy(knots_R, cand_R, sizeof(double) * 4 * N);
                orc_evaluate(&p, patch_blocks, frame_blocks, &eval_cost, H, g);
                orc_lm_accepted(&lm, quality);
                orc_tr_accepted(&tr, eval_cost, model);
                TRACE(lv, iter, 1, p.num_bad, lm.radius, eval_cost, cand_cost, model, quality);
                continue;
            }
            orc_lm_rejected(&lm);
            TRACE(lv, iter, 2, p.num_bad, lm.radius, eval_cost, cand_cost, model, quality);
        }
        free(flags); free(patch_blocks); free(frame_blocks);
    }
#undef TRACE
    if (final_cost) *final_cost = eval_cost;
    free(start_idx); free(H); free(g); free(step); free(Hx); free(cand_t); free(cand_R);
    return ntrace;
}
