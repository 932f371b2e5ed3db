/*
 * mbavo_oracle_vo.c -- CPU ORACLE (test infrastructure, NOT product code), part 2:
 * the callers either side of the hot path.
 *
 *   keyframe pre-processing   core/feature_detectors/FeatureDetectorSemiDense.cpp:16-59,
 *                             FeatureDetectorBase.cpp:49-91,
 *                             ba_tracker/blur_aware_direct_tracker.cpp:342-415 (tmpProcessKeyframe)
 *   Transformation exp / log  core/states/Transformation.cpp:164-178 -> Sophus::SE3d (third party, absent
 *                             from /root/reference, version unpinned by the repo): restated from Sophus'
 *                             published closed forms, "parity unpinned", checked against scipy expm/logm
 *   spline frame changes      core/common/Spline.h:171-219 (TransformTo, TransformByRight)
 *   keyframe decision         blur_aware_direct_tracker.cpp:205-262 (isKeyframe, first overload)
 *   trackFrame                blur_aware_direct_tracker.cpp:88-203
 *
 * The detector uses cv::KeyPoint (OpenCV is absent), so the reference's own detector cannot be compiled here:
 * the restatement is checked by a brute-force numpy property test (tests/test_oracle_frontend.py).
 * Eigen's quaternion product / rotate / normalise are restated in their published operation order.
 */
#define _GNU_SOURCE
#include "mbavo_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* Eigen::Quaterniond pieces used by Transformation / SplineSE3              */
/* ------------------------------------------------------------------------ */
static void q_normalized(const double q[4], double o[4])
{ /* Eigen normalized(): coeffs / sqrt(squaredNorm) when the norm is positive */
    const double z = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (z > 0) { const double n = sqrt(z); for (int i = 0; i < 4; ++i) o[i] = q[i] / n; }
    else memcpy(o, q, 4 * sizeof(double));
}
static void q_inverse(const double q[4], double o[4])
{ /* Eigen inverse(): conjugate / squaredNorm */
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (n2 > 0) { o[0] = -q[0] / n2; o[1] = -q[1] / n2; o[2] = -q[2] / n2; o[3] = q[3] / n2; }
    else { o[0] = o[1] = o[2] = o[3] = 0; }
}
static void q_rot(const double q[4], const double v[3], double o[3])
{ /* Eigen _transformVector: uv = 2 * (q.vec x v); v + w*uv + q.vec x uv */
    double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    o[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    o[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    o[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}

/* Transformation(R, t) normalises R (Transformation.cpp:39-45); poses are (t[3], q[4] xyzw) = 7 doubles */
static void T_make(const double q[4], const double t[3], double T[7])
{
    q_normalized(q, T + 3);
    memcpy(T, t, 3 * sizeof(double));
}
static void T_identity(double T[7]) { memset(T, 0, 7 * sizeof(double)); T[6] = 1; }
static void T_inverse(const double T[7], double o[7])
{ /* Transformation.cpp:83-90 */
    const double qc[4] = {-T[3], -T[4], -T[5], T[6]}, nt[3] = {-T[0], -T[1], -T[2]};
    double ti[3];
    q_rot(qc, nt, ti);
    T_make(qc, ti, o);
}
static void T_mul(const double A[7], const double B[7], double o[7])
{ /* Transformation.cpp:109-119 */
    double q[4], t[3];
    orc_quat_mul(A + 3, B + 3, q);
    q_rot(A + 3, B, t);
    t[0] += A[0]; t[1] += A[1]; t[2] += A[2];
    T_make(q, t, o);
}
static void T_apply(const double T[7], const double p[3], double o[3])
{ /* Transformation.cpp:92-97 */
    q_rot(T + 3, p, o);
    o[0] += T[0]; o[1] += T[1]; o[2] += T[2];
}

void orc_transform_mul(const double A[7], const double B[7], double out[7]) { T_mul(A, B, out); }
void orc_transform_inverse(const double A[7], double out[7]) { T_inverse(A, out); }

/* ------------------------------------------------------------------------ */
/* Sophus::SE3d::exp / log (tangent = [upsilon, omega])                      */
/* ------------------------------------------------------------------------ */
static void hat_sq(const double w[3], double O[9], double O2[9])
{
    const double h[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    memcpy(O, h, sizeof(h));
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double a = 0;
            for (int j = 0; j < 3; ++j) a += h[r * 3 + j] * h[j * 3 + c];
            O2[r * 3 + c] = a;
        }
}

void orc_se3_exp(const double a[6], double t[3], double q[4])
{ /* Transformation.cpp:171-177 -> Sophus::SE3d::exp */
    const double *om = a + 3;
    orc_so3_exp(om, q);
    const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    double O[9], O2[9], V[9];
    hat_sq(om, O, O2);
    if (theta < 1e-10) { /* V = so3.matrix() */
        const double x = q[0], y = q[1], z = q[2], w = q[3];
        const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                             2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                             2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
        memcpy(V, R, sizeof(R));
    } else {
        const double th2 = theta * theta;
        const double c1 = (1 - cos(theta)) / th2, c2 = (theta - sin(theta)) / (th2 * theta);
        for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * O[i] + c2 * O2[i];
    }
    for (int r = 0; r < 3; ++r) t[r] = V[r * 3] * a[0] + V[r * 3 + 1] * a[1] + V[r * 3 + 2] * a[2];
    double qn[4];
    q_normalized(q, qn); /* Transformation(T.unit_quaternion(), ...) normalises again */
    memcpy(q, qn, sizeof(qn));
}

void orc_se3_log(const double t[3], const double q_in[4], double out[6])
{ /* Transformation.cpp:164-169 -> Sophus::SE3d::log (SO3d::logAndTheta inside) */
    double q[4];
    q_normalized(q_in, q); /* the SE3d(quaternion, t) constructor normalises */
    const double sn = q[0] * q[0] + q[1] * q[1] + q[2] * q[2], w = q[3];
    double two_atan, n = sqrt(sn);
    if (sn < 1e-10 * 1e-10) {
        const double sw = w * w;
        two_atan = 2.0 / w - 2.0 * sn / (w * sw);
    } else if (fabs(w) < 1e-10) {
        two_atan = (w > 0 ? M_PI : -M_PI) / n;
    } else {
        two_atan = 2.0 * atan(n / w) / n;
    }
    const double theta = two_atan * n;
    double om[3] = {two_atan * q[0], two_atan * q[1], two_atan * q[2]};
    double O[9], O2[9], Vi[9];
    hat_sq(om, O, O2);
    double c2;
    if (fabs(theta) < 1e-10) c2 = 1.0 / 12.0;
    else { const double h = 0.5 * theta; c2 = (1 - theta * cos(h) / (2 * sin(h))) / (theta * theta); }
    for (int i = 0; i < 9; ++i) Vi[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * O[i] + c2 * O2[i];
    for (int r = 0; r < 3; ++r) out[r] = Vi[r * 3] * t[0] + Vi[r * 3 + 1] * t[1] + Vi[r * 3 + 2] * t[2];
    out[3] = om[0]; out[4] = om[1]; out[5] = om[2];
}

/* ------------------------------------------------------------------------ */
/* SplineSE3 frame changes (Spline.h:171-219)                                */
/* ------------------------------------------------------------------------ */
void orc_spline_get_pose(int k, double t0, double dt, const double *kt, const double *kR, double t,
                         double p[3], double q[4])
{ /* Spline.h:222-281 without Jacobians */
    int idx; double u;
    orc_spline_segment(t, t0, dt, &idx, &u);
    if (k == 2) { orc_c2_vec3(kt + idx * 3, u, p, NULL); orc_c2_rot3(kR + idx * 4, u, q, NULL); }
    else { orc_c4_vec3(kt + idx * 3, u, p, NULL); orc_c4_rot3(kR + idx * 4, u, q, NULL); }
}

void orc_spline_transform_by_right(double *kt, double *kR, int N, const double dq[4], const double dt[3])
{ /* Spline.h:212-219 */
    for (int i = 0; i < N; ++i) {
        double r[3], q[4];
        q_rot(kR + 4 * i, dt, r);
        kt[3 * i] = r[0] + kt[3 * i]; kt[3 * i + 1] = r[1] + kt[3 * i + 1]; kt[3 * i + 2] = r[2] + kt[3 * i + 2];
        orc_quat_mul(kR + 4 * i, dq, q);
        memcpy(kR + 4 * i, q, sizeof(q));
    }
}

void orc_spline_transform_to(int k, double t0, double dtk, double *kt, double *kR, int N, double t,
                             const double q_target[4], const double t_target[3])
{ /* Spline.h:183-200: move the spline so that its pose at time t becomes the target */
    double po[3], qo[4], qi[4], dR[4], d[3], dt[3];
    orc_spline_get_pose(k, t0, dtk, kt, kR, t, po, qo);
    q_inverse(qo, qi);
    orc_quat_mul(qi, q_target, dR);
    d[0] = t_target[0] - po[0]; d[1] = t_target[1] - po[1]; d[2] = t_target[2] - po[2];
    q_rot(qi, d, dt);
    for (int i = 0; i < N; ++i) {
        double r[3], q[4];
        q_rot(kR + 4 * i, dt, r);
        kt[3 * i] += r[0]; kt[3 * i + 1] += r[1]; kt[3 * i + 2] += r[2];
        orc_quat_mul(kR + 4 * i, dR, q);
        memcpy(kR + 4 * i, q, sizeof(q));
    }
}

/* ------------------------------------------------------------------------ */
/* semi-dense keypoints                                                      */
/* ------------------------------------------------------------------------ */
int orc_detect_semidense(const float *mag, int H, int W, int lv, int im_H0, int im_W0, int cell_H, int cell_W,
                         float thr, float *out_xy, float *out_resp, int cap)
{
    /* FeatureDetectorSemiDense.cpp:27-43: candidates = pixels with magnitude > threshold, row-major order;
     * FeatureDetectorBase.cpp:49-91: per grid cell keep the first candidate of strictly largest response */
    int n = 0;
    if (!(cell_H > 0 && cell_W > 0)) {
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w)
                if (mag[(size_t)h * W + w] > thr) {
                    if (n < cap) { out_xy[2 * n] = (float)w; out_xy[2 * n + 1] = (float)h; if (out_resp) out_resp[n] = mag[(size_t)h * W + w]; }
                    ++n;
                }
        return n;
    }
    const int scale_factor = (int)pow(2, lv);
    const int im_H_lv = im_H0 / scale_factor, im_W_lv = im_W0 / scale_factor;
    const int cell_H_lv = (int)(cell_H / pow(1.414, lv)), cell_W_lv = (int)(cell_W / pow(1.414, lv));
    const int nCellsH = im_H_lv / cell_H_lv + 1, nCellsW = im_W_lv / cell_W_lv + 1;
    const int nc = nCellsH * nCellsW;
    float *resp = (float *)calloc(nc, sizeof(float)), *xy = (float *)calloc(2 * (size_t)nc, sizeof(float));
    for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w) {
            const float m = mag[(size_t)h * W + w];
            if (!(m > thr)) continue;
            const int ci = (int)((float)h / cell_H_lv) * nCellsW + (int)((float)w / cell_W_lv);
            if (ci < 0 || ci >= nc) continue; /* the reference's .at() would throw; cannot happen for pyramid sizes */
            if (resp[ci] < m) { resp[ci] = m; xy[2 * ci] = (float)w; xy[2 * ci + 1] = (float)h; }
        }
    for (int i = 0; i < nc; ++i) {
        if (resp[i] < 1e-6) continue;
        if (n < cap) { out_xy[2 * n] = xy[2 * i]; out_xy[2 * n + 1] = xy[2 * i + 1]; if (out_resp) out_resp[n] = resp[i]; }
        ++n;
    }
    free(resp); free(xy);
    return n;
}

int orc_keypoint_depths(const float *kp_xy, int n, int lv, const float *depth_z, int H0, int W0,
                        double *out_xy, double *out_z)
{ /* blur_aware_direct_tracker.cpp:389-415: depth at the level-0 position, drop z < 1e-2 */
    const double scale = pow(2, lv);
    int K = 0;
    for (int i = 0; i < n; ++i) {
        const int x = (int)(kp_xy[2 * i] * scale + 0.5), y = (int)(kp_xy[2 * i + 1] * scale + 0.5);
        (void)H0;
        const float z = depth_z[(size_t)y * W0 + x];
        if (z < 1e-2) continue;
        out_xy[2 * K] = kp_xy[2 * i]; out_xy[2 * K + 1] = kp_xy[2 * i + 1];
        out_z[K] = z;
        ++K;
    }
    return K;
}

/* ------------------------------------------------------------------------ */
/* keyframe decision                                                         */
/* ------------------------------------------------------------------------ */
static int cam_project(const double intr[4], const double P[3], double p[2])
{ /* CameraPinhole.cpp:24-43 (no distortion model attached) */
    if (P[2] < 0) return 0;
    p[0] = intr[0] * (P[0] / (P[2] + 1e-8)) + intr[2];
    p[1] = intr[1] * (P[1] / (P[2] + 1e-8)) + intr[3];
    return 1;
}

int orc_is_keyframe(const double intr[4], const double *kp_xy, const double *kp_z, int K,
                    int k, double t0, double dtk, const double *kt, const double *kR,
                    double cap, double exp_t, double flow_mag0, double flow_mag1, double max_kernel,
                    double *avg_flow_out, double *avg_kernel_out)
{ /* blur_aware_direct_tracker.cpp:205-262.  A point that projects behind the camera leaves the reference's
   * output vector uninitialised (undefined); here it keeps the value (0,0). */
    double flow = 0, kern = 0;
    const double times[3] = {cap, cap - 0.5 * exp_t, cap + 0.5 * exp_t};
    double Tinv[3][7];
    for (int j = 0; j < 3; ++j) {
        double p[3], q[4], T[7];
        orc_spline_get_pose(k, t0, dtk, kt, kR, times[j], p, q);
        T_make(q, p, T);
        T_inverse(T, Tinv[j]);
    }
    for (int i = 0; i < K; ++i) {
        const double x = kp_xy[2 * i], y = kp_xy[2 * i + 1], z = kp_z[i];
        const double P[3] = {(x - intr[2]) / intr[0] * z, (y - intr[3]) / intr[1] * z, z}; /* CameraPinhole.cpp:79-94 */
        double Pc[3], a[2] = {0, 0}, b[2] = {0, 0}, c[2] = {0, 0};
        T_apply(Tinv[0], P, Pc); cam_project(intr, Pc, a);
        flow += (a[0] - x) * (a[0] - x) + (a[1] - y) * (a[1] - y);
        T_apply(Tinv[1], P, Pc); cam_project(intr, Pc, b);
        T_apply(Tinv[2], P, Pc); cam_project(intr, Pc, c);
        kern += (b[0] - c[0]) * (b[0] - c[0]) + (b[1] - c[1]) * (b[1] - c[1]);
    }
    const double avg_flow = sqrtf(flow / K), avg_kernel = sqrtf(kern / K);
    if (avg_flow_out) *avg_flow_out = avg_flow;
    if (avg_kernel_out) *avg_kernel_out = avg_kernel;
    if (avg_flow > flow_mag0 && avg_kernel < max_kernel) return 1;
    if (avg_flow > flow_mag1) return 1;
    return 0;
}

/* ------------------------------------------------------------------------ */
/* trackFrame                                                                */
/* ------------------------------------------------------------------------ */
#define ORC_MAX_LV 8
#define ORC_VO_TRACE_CAP 512
struct orc_vo {
    orc_vo_opts o;
    int first;
    /* keyframe */
    unsigned char *ref[ORC_MAX_LV]; float *grad[ORC_MAX_LV];
    double *kp_xy[ORC_MAX_LV], *kp_z[ORC_MAX_LV]; int K[ORC_MAX_LV];
    unsigned char *cur[ORC_MAX_LV];
    int *pattern[ORC_MAX_LV];
    /* spline + motion model */
    double kt[3 * 16], kR[4 * 16]; int N; double t0, dtk; /* max_num_ctrl_knots = 16 */
    double T_keyframe[7], T_prev_b2w[7], vel[6], prev_stamp;
    double last_cost;
    orc_trace_rec trace[ORC_VO_TRACE_CAP]; int ntrace; /* the last trackFrame's LM records (long-horizon parity runs) */
};

orc_vo *orc_vo_create(const orc_vo_opts *o)
{ /* BlurAwareDirectTracker ctor (blur_aware_direct_tracker.cpp:14-34) */
    if (o->num_levels < 1 || o->num_levels > ORC_MAX_LV) return NULL;
    orc_vo *v = (orc_vo *)calloc(1, sizeof(orc_vo));
    v->o = *o;
    v->first = 1;
    for (int l = 0; l < o->num_levels; ++l) {
        const int Hl = o->H >> l, Wl = o->W >> l, P = o->patch_size[l];
        v->ref[l] = (unsigned char *)malloc((size_t)Hl * Wl);
        v->cur[l] = (unsigned char *)malloc((size_t)Hl * Wl);
        v->grad[l] = (float *)malloc(sizeof(float) * 2 * Hl * Wl);
        v->pattern[l] = (int *)malloc(sizeof(int) * 2 * P);
        memcpy(v->pattern[l], o->pattern_xy[l], sizeof(int) * 2 * P);
    }
    v->dtk = o->dt_ctrl_knot;
    T_identity(v->T_keyframe); T_identity(v->T_prev_b2w);
    return v;
}

void orc_vo_destroy(orc_vo *v)
{
    if (!v) return;
    for (int l = 0; l < v->o.num_levels; ++l) {
        free(v->ref[l]); free(v->cur[l]); free(v->grad[l]); free(v->pattern[l]); free(v->kp_xy[l]); free(v->kp_z[l]);
    }
    free(v);
}

static void process_keyframe(orc_vo *v, const unsigned char *sharp, const float *depth_z)
{ /* tmpProcessKeyframe (blur_aware_direct_tracker.cpp:342-415) */
    const orc_vo_opts *o = &v->o;
    memcpy(v->ref[0], sharp, (size_t)o->H * o->W);
    for (int l = 1; l < o->num_levels; ++l) orc_pyramid_down_u8(v->ref[l - 1], o->H >> (l - 1), o->W >> (l - 1), v->ref[l]);
    for (int l = 0; l < o->num_levels; ++l) {
        const int Hl = o->H >> l, Wl = o->W >> l;
        float *mag = (float *)malloc(sizeof(float) * Hl * Wl);
        orc_image_gradients_u8(v->ref[l], Hl, Wl, v->grad[l], mag);
        const int cap = Hl * Wl;
        float *xy = (float *)malloc(sizeof(float) * 2 * cap);
        const int n = orc_detect_semidense(mag, Hl, Wl, l, o->H, o->W, o->grid_cell_H, o->grid_cell_W, o->score_threshold, xy, NULL, cap);
        free(v->kp_xy[l]); free(v->kp_z[l]);
        v->kp_xy[l] = (double *)malloc(sizeof(double) * 2 * (n > 0 ? n : 1));
        v->kp_z[l] = (double *)malloc(sizeof(double) * (n > 0 ? n : 1));
        v->K[l] = orc_keypoint_depths(xy, n, l, depth_z, o->H, o->W, v->kp_xy[l], v->kp_z[l]);
        free(mag); free(xy);
    }
}

int orc_vo_set_spline(orc_vo *v, double t0, double dt, int N, const double *kt, const double *kR)
{ /* getSplineTrajectory()->InsertControlKnot(...) before the first frame (the `get_num_knots() == 0` test at :99) */
    if (N < 0 || N > 16) return -1;
    v->t0 = t0; v->dtk = dt; v->N = N;
    memcpy(v->kt, kt, sizeof(double) * 3 * N);
    memcpy(v->kR, kR, sizeof(double) * 4 * N);
    return 0;
}

int orc_vo_last_trace(const orc_vo *v, orc_trace_rec *out, int cap)
{ /* records of the last trackFrame's optimizeTrajectory, in order; returns how many were copied */
    const int n = v->ntrace < cap ? v->ntrace : cap;
    if (n > 0) memcpy(out, v->trace, sizeof(orc_trace_rec) * n);
    return n;
}
void orc_vo_get_state(const orc_vo *v, orc_vo_state *s)
{
    memset(s, 0, sizeof(*s));
    s->t0 = v->t0; s->dt = v->dtk; s->N = v->N; s->is_first = v->first;
    memcpy(s->knots_t, v->kt, sizeof(v->kt)); memcpy(s->knots_R, v->kR, sizeof(v->kR));
    memcpy(s->T_keyframe, v->T_keyframe, sizeof(v->T_keyframe)); memcpy(s->T_prev_b2w, v->T_prev_b2w, sizeof(v->T_prev_b2w));
    memcpy(s->velocity, v->vel, sizeof(v->vel)); s->prev_timestamp = v->prev_stamp;
}
void orc_vo_set_state(orc_vo *v, const orc_vo_state *s)
{
    v->t0 = s->t0; v->dtk = s->dt; v->N = s->N; v->first = s->is_first;
    memcpy(v->kt, s->knots_t, sizeof(v->kt)); memcpy(v->kR, s->knots_R, sizeof(v->kR));
    memcpy(v->T_keyframe, s->T_keyframe, sizeof(v->T_keyframe)); memcpy(v->T_prev_b2w, s->T_prev_b2w, sizeof(v->T_prev_b2w));
    memcpy(v->vel, s->velocity, sizeof(v->vel)); v->prev_stamp = s->prev_timestamp;
}
void orc_vo_set_keyframe(orc_vo *v, const unsigned char *sharp, const float *depth_z) { process_keyframe(v, sharp, depth_z); }
int orc_vo_num_keypoints(const orc_vo *v, int level) { return v->K[level]; }
void orc_vo_keypoints(const orc_vo *v, int level, double *xy, double *z)
{
    memcpy(xy, v->kp_xy[level], sizeof(double) * 2 * v->K[level]);
    memcpy(z, v->kp_z[level], sizeof(double) * v->K[level]);
}
void orc_vo_spline(const orc_vo *v, double *t0, double *dt, int *N, double *kt, double *kR)
{
    *t0 = v->t0; *dt = v->dtk; *N = v->N;
    memcpy(kt, v->kt, sizeof(double) * 3 * v->N);
    memcpy(kR, v->kR, sizeof(double) * 4 * v->N);
}

int orc_vo_track_frame(orc_vo *v, const unsigned char *sharp, const float *depth_z, double sharp_cap,
                       const unsigned char *blur, double blur_cap, double blur_exp,
                       double T_out[7], orc_vo_info *info)
{ /* trackFrame (blur_aware_direct_tracker.cpp:88-203) */
    const orc_vo_opts *o = &v->o;
    if (info) memset(info, 0, sizeof(*info));
    if (v->first) {
        v->first = 0;
        process_keyframe(v, sharp, depth_z);
        v->prev_stamp = sharp_cap;
        if (v->N == 0) { /* :99-106: two identity knots whatever the spline degree */
            v->dtk = o->dt_frame;
            v->t0 = sharp_cap;
            memset(v->kt, 0, sizeof(v->kt)); memset(v->kR, 0, sizeof(v->kR));
            v->kR[3] = 1; v->kR[7] = 1;
            v->N = 2;
        }
        memcpy(T_out, v->T_keyframe, sizeof(double) * 7);
        if (info) { info->is_keyframe = 1; info->num_keypoints0 = v->K[0]; }
        return 0;
    }
    memcpy(v->cur[0], blur, (size_t)o->H * o->W);
    for (int l = 1; l < o->num_levels; ++l) orc_pyramid_down_u8(v->cur[l - 1], o->H >> (l - 1), o->W >> (l - 1), v->cur[l]);

    const double dt_frame = blur_cap - v->prev_stamp;
    for (int i = 0; i < 6; ++i) v->vel[i] *= dt_frame; /* :123-140: constant-velocity prediction */
    double dT[7];
    orc_se3_exp(v->vel, dT, dT + 3);
    v->t0 = blur_cap - 0.5 * blur_exp;
    orc_spline_transform_by_right(v->kt, v->kR, v->N, dT + 3, dT);

    /* optimizeTrajectory (:544-588) */
    orc_level lv[ORC_MAX_LV];
    const unsigned char *curp[ORC_MAX_LV][1];
    for (int l = 0; l < o->num_levels; ++l) {
        curp[l][0] = v->cur[l];
        lv[l].H = o->H >> l; lv[l].W = o->W >> l; lv[l].K = v->K[l]; lv[l].P = o->patch_size[l]; lv[l].S = o->num_virtual_poses[l];
        lv[l].ref_img = v->ref[l]; lv[l].ref_dIxy = v->grad[l]; lv[l].cur_imgs = curp[l];
        lv[l].kp_xy = v->kp_xy[l]; lv[l].kp_z = v->kp_z[l]; lv[l].pattern = v->pattern[l];
    }
    orc_track_opts to;
    memset(&to, 0, sizeof(to));
    to.num_levels = o->num_levels; to.k = o->spline_deg_k; to.max_num_iterations = o->max_num_iterations;
    to.max_nonmono = o->max_nonmono; to.solver_type = o->solver_type;
    memcpy(to.intr, o->intr, sizeof(to.intr));
    to.huber_k = o->huber_k; to.min_step_quality = o->min_step_quality;
    to.min_abs_cost_decrease = o->min_abs_cost_decrease; to.max_chi_square_error = o->max_chi_square_error;
    int start = 0;
    const int ntrace = orc_optimize_trajectory(&to, lv, 1, &blur_cap, &blur_exp, v->t0, v->dtk, v->kt, v->kR, v->N,
                                               &start, &v->last_cost, v->trace, ORC_VO_TRACE_CAP);
    v->ntrace = ntrace < ORC_VO_TRACE_CAP ? (ntrace > 0 ? ntrace : 0) : ORC_VO_TRACE_CAP;

    double af = 0, ak = 0;
    const int is_kf = orc_is_keyframe(o->intr, v->kp_xy[0], v->kp_z[0], v->K[0], o->spline_deg_k, v->t0, v->dtk, v->kt, v->kR,
                                      blur_cap, blur_exp, o->keyframe_max_flow_mag0, o->keyframe_max_flow_mag1,
                                      o->keyframe_max_blur_kernel_mag, &af, &ak);

    double p[3], q[4], T_b2w[7], Ti[7], dTn[7], lg[6];
    orc_spline_get_pose(o->spline_deg_k, v->t0, v->dtk, v->kt, v->kR, blur_cap, p, q);
    T_make(q, p, T_b2w);
    T_inverse(v->T_prev_b2w, Ti);
    T_mul(Ti, T_b2w, dTn);
    orc_se3_log(dTn, dTn + 3, lg);
    for (int i = 0; i < 6; ++i) v->vel[i] = lg[i] / dt_frame; /* :152-155 */
    memcpy(v->T_prev_b2w, T_b2w, sizeof(T_b2w));

    if (is_kf) { /* :176-188 */
        process_keyframe(v, sharp, depth_z);
        double T[7], Tk[7];
        orc_spline_get_pose(o->spline_deg_k, v->t0, v->dtk, v->kt, v->kR, blur_cap, p, q);
        T_make(q, p, T);
        T_mul(v->T_keyframe, T, Tk);
        memcpy(v->T_keyframe, Tk, sizeof(Tk));
        const double qi[4] = {0, 0, 0, 1}, ti[3] = {0, 0, 0};
        orc_spline_transform_to(o->spline_deg_k, v->t0, v->dtk, v->kt, v->kR, v->N, blur_cap, qi, ti);
        T_identity(v->T_prev_b2w);
    }
    v->prev_stamp = blur_cap;
    double T[7];
    orc_spline_get_pose(o->spline_deg_k, v->t0, v->dtk, v->kt, v->kR, blur_cap, p, q);
    T_make(q, p, T);
    T_mul(v->T_keyframe, T, T_out);
    if (info) {
        info->is_keyframe = is_kf; info->num_keypoints0 = v->K[0]; info->avg_flow = af; info->avg_kernel = ak;
        info->final_cost = v->last_cost; info->num_trace = ntrace; info->start_idx = start;
    }
    return 0;
}
