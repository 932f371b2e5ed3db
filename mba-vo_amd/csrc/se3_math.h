// se3_math.h -- SE(3) cumulative B-spline sampling with pose-to-knot Jacobians
// (host + device, header only).
//
// Same quantities as the reference's functors (core/common/SplineFunctor.h:13-365,
// Quaternion.h:61-283): pose [t | q(xyzw)] on a degree-k (k = 2 linear, 4 cubic)
// cumulative B-spline and d(pose)/d(knots) for the right-multiplicative local
// update R_j <- R_j * exp(w_j).  The reference builds the 4x3k rotation Jacobian
// with 4x4 left/right product matrices staged through global scratch; here a
// 4x3 Jacobian block is kept as three quaternion columns and every
// "matrix(q) * block" is a quaternion product, so the whole chain stays in
// registers.  Small-angle branch thresholds (Quaternion.h:77,100,166) are kept.
#ifndef MBAVO_SE3_MATH_H
#define MBAVO_SE3_MATH_H

#include "core_types.h"
#include <math.h>

namespace mbavo
{
    struct Quat
    {
        double x, y, z, w;
    };

    MBAVO_HD Quat qmul(const Quat &a, const Quat &b)
    { // Hamilton product, term order of Quaternion.h:45-51.  No FMA contraction: patch centres computed with it
      // are truncated to integer pixels (compute_hessian_gradients_cost.cu:69-70), so the last bit decides
      // which pixel is read when a centre falls on an integer
#pragma clang fp contract(off)
        Quat r;
        r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
        r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
        r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
        r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
        return r;
    }

    MBAVO_HD Quat qconj(const Quat &a) { return Quat{-a.x, -a.y, -a.z, a.w}; }

    MBAVO_HD void qrotate(const Quat &q, const double p[3], double o[3])
    { // q * (p,0) * conj(q)   (Quaternion.h:53-60)
        const Quat v = qmul(qmul(q, Quat{p[0], p[1], p[2], 0.0}), qconj(q));
        o[0] = v.x; o[1] = v.y; o[2] = v.z;
    }


#if defined(__HIP_DEVICE_COMPILE__)
#define MBAVO_POSE_FASTMATH 1
    // Device forms of the four library calls on the pose chain (round 3).  A sample's pose entries are ONE dependent chain per
    // lane -- 2.9 us of stage A on configs[1], every evaluation waits for it -- and the runtime's sqrt / division / atan / sin /
    // cos are general-purpose sequences (argument reduction for any double, special values): 31, 18, 95, 189 and 189
    // instructions.  The arguments here are a rotation's half-angle and the tangent of one: finite, moderate.  Each form
    // below is within one or two units in the last place of the correctly rounded value (the runtime's own bound for sin /
    // cos / atan); polynomial coefficients and reduction constants are those of fdlibm's k_sin.c / k_cos.c / s_atan.c /
    // e_rem_pio2.c.
    namespace fastm
    {
        __device__ __forceinline__ double rcp(double x) // 1 / x: v_rcp_f64 (2^-23) + two Newton steps; x finite, normal, non-zero
        {
            double r = __builtin_amdgcn_rcp(x);
            r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
            r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
            return r;
        }
        // s = sqrt(x), r = 1 / sqrt(x) from ONE v_rsq_f64 + two Newton steps (the chain needs both everywhere it takes a norm)
        __device__ __forceinline__ void sqrt_rsqrt(double x, double &s, double &r)
        {
            double y = __builtin_amdgcn_rsq(x);
            const double h = 0.5 * x;
            y = __builtin_fma(y, __builtin_fma(-(h * y), y, 0.5), y);
            y = __builtin_fma(y, __builtin_fma(-(h * y), y, 0.5), y);
            double t = x * y;
            t = __builtin_fma(__builtin_fma(-t, t, x), 0.5 * y, t); // one correction of the root itself
            s = t;
            r = y;
        }
        // sin and cos of a moderate argument (two-constant Cody-Waite reduction: good to |x| ~ 1e5; the chain's arguments are
        // half of c * |log(q)| <= pi / 2 -- no large-argument path, a NaN stays a NaN)
        __device__ __forceinline__ void sincos(double x, double &sn, double &cs)
        {
            const double k = rint(x * 6.36619772367581382433e-01);
            double r = __builtin_fma(-k, 1.57079632673412561417e+00, x); // exact: the constant holds 33 bits of pi/2
            r = __builtin_fma(-k, 6.07710050650619224932e-11, r);
            const double z = r * r;
            const double ps = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                              z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
            const double ks = r + (z * r) * (-1.66666666666666324348e-01 + z * ps);
            const double pc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                              z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
            const double hz = 0.5 * z, w = 1.0 - hz;
            const double kc = w + (((1.0 - w) - hz) + z * pc);
            const int q = (int)k & 3;
            sn = (q & 1) ? kc : ks;
            cs = (q & 1) ? ks : kc;
            if (q == 1 || q == 2) cs = -cs;
            if (q >= 2) sn = -sn;
        }
        // atan(num / den) for finite num, den (den != 0 where num != 0: the callers branch on |den| < 1e-10 first)
        __device__ __forceinline__ double atan_ratio(double num, double den)
        {
            const double x0 = num * rcp(den);
            const double ax = fabs(x0);
            double x = ax, hi = 0.0, lo = 0.0;
            if (ax >= 0.4375)
            {
                if (ax < 1.1875)
                {
                    if (ax < 0.6875) { x = (2.0 * ax - 1.0) * rcp(2.0 + ax); hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17; }
                    else { x = (ax - 1.0) * rcp(ax + 1.0); hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17; }
                }
                else if (ax < 2.4375) { x = (ax - 1.5) * rcp(1.0 + 1.5 * ax); hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17; }
                else { x = ax < 1e300 ? -rcp(ax) : 0.0; hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17; }
            }
            const double z = x * x, w = z * z;
            const double s1 = z * (3.33333333333329318027e-01 + w * (1.42857142725034663711e-01 + w * (9.09088713343650656196e-02 +
                              w * (6.66107313738753120669e-02 + w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
            const double s2 = w * (-1.99999999998764832476e-01 + w * (-1.11111104054623557880e-01 + w * (-7.69187620504482999495e-02 +
                              w * (-5.83357013379057348645e-02 + w * -3.65315727442169155270e-02))));
            const double r = ax < 0.4375 ? x - x * (s1 + s2) : hi - ((x * (s1 + s2) - lo) - x);
            return x0 < 0 ? -r : r;
        }
    } // namespace fastm
#endif

    // A 4x3 Jacobian block d(quaternion)/d(3-vector), stored as its columns.  Every operation below acts on the
    // columns independently, so a block can also be processed one column at a time (NC = 1: the pose-table kernel
    // gives each column of a block its own lane).
    template <int NC>
    struct JacC
    {
        Quat c[NC];
    };
    typedef JacC<3> Jac43;
    // 3x4 (log) and 4x3 (exp) Jacobians, row-major as in the reference.
    struct LogJac { double m[12]; };
    struct ExpJac { double m[12]; };

    // log map with the reference's three branches (Quaternion.h:61-157).
    template <bool WITH_J>
    MBAVO_HD void qlog(const Quat &q, double tg[3], LogJac *J)
    {
        const double x = q.x, y = q.y, z = q.z, w = q.w;
        const double sn = x * x + y * y + z * z;
        double lam, dx = 0, dy = 0, dz = 0, dw = 0;
        if (sn < 1e-20)
        {
            const double www = w * w * w;
            lam = 2. / w - 2. / 3. * sn / www;
            if (WITH_J)
            { // series derivative exactly as the reference writes it (:80-88)
                dx = 2. / w - 4. / 3. * x / www;
                dy = 2. / w - 4. / 3. * y / www;
                dz = 2. / w - 4. / 3. * z / www;
                dw = -2 / (w * w) + 2 * sn / (www * w);
            }
        }
        else
        {
#if defined(MBAVO_POSE_FASTMATH)
            double n, n_rcp;
            fastm::sqrt_rsqrt(sn, n, n_rcp);
#else
            const double n = sqrt(sn);
#endif
            if (fabs(w) < 1e-10)
            {
                const double pi = 3.14159265358979323846;
                lam = (w > 0 ? pi : -pi) / n;
                if (WITH_J)
                {
                    const double s = (w > 0 ? -lam : lam) / sn;
                    dx = s * x; dy = s * y; dz = s * z;
                }
            }
            else
            {
#if defined(__HIP_DEVICE_COMPILE__)
                // device: ONE reciprocal of n and products where the reference divides by n four times (an IEEE fp64
                // division is ~13 dependent instructions, and a pose lane walks this chain alone on its SIMD); the results
                // differ from the quotients in the last place at most
#if defined(MBAVO_POSE_FASTMATH)
                const double rn = n_rcp;
                lam = 2.0 * fastm::atan_ratio(n, w) * rn;
#else
                const double rn = 1.0 / n; // (v_rsq_f64 + Newton instead of sqrt + division: no measurable difference)
                lam = 2.0 * atan(n / w) * rn;
#endif
                if (WITH_J)
                {
                    const double dn = (2 * w - lam) * rn * rn;
                    dx = dn * x; dy = dn * y; dz = dn * z;
                    dw = -2.;
                }
#else
                lam = 2.0 * atan(n / w) / n;
                if (WITH_J)
                {
                    const double dn = (2 * w - lam) / n;
                    dx = dn * x / n; dy = dn * y / n; dz = dn * z / n;
                    dw = -2.;
                }
#endif
            }
        }
        if (WITH_J)
        {
            double *m = J->m;
            m[0] = dx * x + lam; m[1] = dy * x;        m[2] = dz * x;         m[3] = dw * x;
            m[4] = dx * y;       m[5] = dy * y + lam;  m[6] = dz * y;         m[7] = dw * y;
            m[8] = dx * z;       m[9] = dy * z;        m[10] = dz * z + lam;  m[11] = dw * z;
        }
        tg[0] = lam * x; tg[1] = lam * y; tg[2] = lam * z;
    }

    // exp map (Quaternion.h:159-233).
    template <bool WITH_J>
    MBAVO_HD Quat qexp(const double tg[3], ExpJac *J)
    {
        double im, re;
        const double th2 = tg[0] * tg[0] + tg[1] * tg[1] + tg[2] * tg[2];
        if (th2 < 1e-20)
        {
            const double th4 = th2 * th2;
            im = 0.5 - 1. / 48. * th2 + 1. / 3840. * th4;
            re = 1. - 1. / 8. * th2 + 1. / 384. * th4;
            if (WITH_J)
            {
                for (int i = 0; i < 12; ++i) J->m[i] = 0;
                J->m[0] = 0.5; J->m[4] = 0.5; J->m[8] = 0.5;
            }
        }
        else
        {
#if defined(MBAVO_POSE_FASTMATH)
            double th, rth, hs, hc;
            fastm::sqrt_rsqrt(th2, th, rth);
            fastm::sincos(0.5 * th, hs, hc);
#else
            const double th = sqrt(th2);
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#if !defined(MBAVO_POSE_FASTMATH)
            // (sincos() for the pair measured 2 us SLOWER per evaluation: its results come back through private memory)
            const double hs = sin(0.5 * th), hc = cos(0.5 * th);
            const double rth = 1.0 / th; // (device: one reciprocal instead of six divisions by theta, see qlog)
#endif
            im = hs * rth;
            re = hc;
            if (WITH_J)
            {
                const double x = tg[0], y = tg[1], z = tg[2];
                const double ux = x * rth, uy = y * rth, uz = z * rth;
                const double dim = (0.5 * re - im) * rth;
#else
            const double hs = sin(0.5 * th);
            im = hs / th;
            re = cos(0.5 * th);
            if (WITH_J)
            {
                const double x = tg[0], y = tg[1], z = tg[2];
                const double ux = x / th, uy = y / th, uz = z / th;
                const double dim = 0.5 * re / th - im / th;
#endif
                const double dre = -0.5 * hs;
                const double ax = dim * ux, ay = dim * uy, az = dim * uz;
                double *m = J->m;
                m[0] = ax * x + im; m[1] = ay * x;      m[2] = az * x;
                m[3] = ax * y;      m[4] = ay * y + im; m[5] = az * y;
                m[6] = ax * z;      m[7] = ay * z;      m[8] = az * z + im;
                m[9] = dre * ux;    m[10] = dre * uy;   m[11] = dre * uz;
            }
        }
        return Quat{im * tg[0], im * tg[1], im * tg[2], re};
    }

    // d(R * exp(w))/dw at w = 0: columns R * (e_i/2, 0)      (L(R) * [I/2; 0]); columns first .. first + NC - 1
    template <int NC>
    MBAVO_HD JacC<NC> local_param_jac_cols(const Quat &R, int first)
    {
        JacC<NC> o;
        for (int i = 0; i < NC; ++i)
        {
            const int a = first + i;
            o.c[i] = qmul(R, Quat{a == 0 ? 0.5 : 0.0, a == 1 ? 0.5 : 0.0, a == 2 ? 0.5 : 0.0, 0.0});
        }
        return o;
    }
    MBAVO_HD Jac43 local_param_jac(const Quat &R) { return local_param_jac_cols<3>(R, 0); }
    template <int NC>
    MBAVO_HD JacC<NC> lmul(const Quat &q, const JacC<NC> &M) // L(q) * M : q (x) column
    {
        JacC<NC> o;
        for (int i = 0; i < NC; ++i) o.c[i] = qmul(q, M.c[i]);
        return o;
    }
    template <int NC>
    MBAVO_HD JacC<NC> rmul(const JacC<NC> &M, const Quat &q) // Rhat(q) * M : column (x) q
    {
        JacC<NC> o;
        for (int i = 0; i < NC; ++i) o.c[i] = qmul(M.c[i], q);
        return o;
    }
    template <int NC>
    MBAVO_HD JacC<NC> conj_cols(const JacC<NC> &M) // K * M, K = diag(-1,-1,-1,1)
    {
        JacC<NC> o;
        for (int i = 0; i < NC; ++i) o.c[i] = qconj(M.c[i]);
        return o;
    }
    // dexp(4x3) * (scale * dlog(3x4) * M(4xNC))
    template <int NC>
    MBAVO_HD JacC<NC> through_log_exp(const ExpJac &de, double scale, const LogJac &dl, const JacC<NC> &M)
    {
        double t[3][NC];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < NC; ++c)
            {
                const Quat &v = M.c[c];
                double a = 0.0;
                a += dl.m[r * 4 + 0] * v.x;
                a += dl.m[r * 4 + 1] * v.y;
                a += dl.m[r * 4 + 2] * v.z;
                a += dl.m[r * 4 + 3] * v.w;
                t[r][c] = a * scale;
            }
        JacC<NC> o;
        for (int c = 0; c < NC; ++c)
        {
            double v[4];
            for (int r = 0; r < 4; ++r)
            {
                double a = 0.0;
                a += de.m[r * 3 + 0] * t[0][c];
                a += de.m[r * 3 + 1] * t[1][c];
                a += de.m[r * 3 + 2] * t[2][c];
                v[r] = a;
            }
            o.c[c] = Quat{v[0], v[1], v[2], v[3]};
        }
        return o;
    }
    template <int NC>
    MBAVO_HD JacC<NC> jadd(const JacC<NC> &a, const JacC<NC> &b)
    {
        JacC<NC> o;
        for (int i = 0; i < NC; ++i)
            o.c[i] = Quat{a.c[i].x + b.c[i].x, a.c[i].y + b.c[i].y, a.c[i].z + b.c[i].z, a.c[i].w + b.c[i].w};
        return o;
    }
    // write a 4x3 block into the row-major 4 x ncols matrix at column col0
    MBAVO_HD void store_block(double *J, int ncols, int col0, const Jac43 &B)
    {
        for (int i = 0; i < 3; ++i)
        {
            J[0 * ncols + col0 + i] = B.c[i].x;
            J[1 * ncols + col0 + i] = B.c[i].y;
            J[2 * ncols + col0 + i] = B.c[i].z;
            J[3 * ncols + col0 + i] = B.c[i].w;
        }
    }

    MBAVO_HD Quat load_quat(const double *p) { return Quat{p[0], p[1], p[2], p[3]}; }

    // segment index + normalised time (SplineFunctor.h:13-19): truncation toward zero
    MBAVO_HD void spline_segment(double t, double t0, double dt, int &idx, double &u)
    {
        const double tn = (t - t0) / dt;
        idx = (int)tn;
        u = tn - idx;
    }

    // basis weights of the translation spline; J_t = kron(coeffs, I3) (SplineFunctor.h:21-94)
    template <int KDEG>
    MBAVO_HD void trans_coeffs(double u, double c[KDEG])
    {
        if constexpr (KDEG == 2)
        {
            c[0] = 1 - u;
            c[1] = u;
        }
        else
        {
            const double uu = u * u, uuu = uu * u, s = 1. / 6.;
            c[0] = s - 0.5 * u + 0.5 * uu - s * uuu;
            c[1] = 4 * s - uu + 0.5 * uuu;
            c[2] = s + 0.5 * u + 0.5 * uu - 0.5 * uuu;
            c[3] = s * uuu;
        }
    }

    template <int KDEG>
    MBAVO_HD void spline_translation(const double *kt, const double c[KDEG], double p[3])
    {
        for (int a = 0; a < 3; ++a)
        {
            double v = c[0] * kt[a];
            for (int j = 1; j < KDEG; ++j) v = v + c[j] * kt[3 * j + a];
            p[a] = v;
        }
    }

    // rotation on the cumulative spline + 4 x 3k Jacobian (row-major) w.r.t. the
    // knots' local parameters.  JR may be nullptr when WITH_J is false.
    template <int KDEG, bool WITH_J>
    MBAVO_HD Quat spline_rotation(const double *kR, double u, double *JR)
    {
        if constexpr (KDEG == 2)
        { // SplineFunctor.h:155-217
            const Quat R0 = load_quat(kR), R1 = load_quat(kR + 4);
            const Quat R0c = qconj(R0);
            LogJac dl; ExpJac de;
            double om[3];
            qlog<WITH_J>(qmul(R0c, R1), om, &dl);
            om[0] *= u; om[1] *= u; om[2] *= u;
            const Quat A0 = qexp<WITH_J>(om, &de);
            if (WITH_J)
            {
                const Jac43 E0 = local_param_jac(R0), E1 = local_param_jac(R1);
                // R0: direct factor + through R01 = conj(R0) * R1
                Jac43 a = rmul(E0, A0);
                Jac43 b = lmul(R0, through_log_exp(de, u, dl, rmul(conj_cols(E0), R1)));
                store_block(JR, 6, 0, jadd(a, b));
                // R1: through R01 only
                Jac43 c = lmul(R0, through_log_exp(de, u, dl, lmul(R0c, E1)));
                store_block(JR, 6, 3, c);
            }
            return qmul(R0, A0);
        }
        else
        { // SplineFunctor.h:219-365
            const double uu = u * u, uuu = uu * u, s = 1. / 6.;
            const double c1 = 5 * s + 0.5 * u - 0.5 * uu + s * uuu;
            const double c2 = s + 0.5 * u + 0.5 * uu - 2 * s * uuu;
            const double c3 = s * uuu;
            const Quat R0 = load_quat(kR), R1 = load_quat(kR + 4), R2 = load_quat(kR + 8), R3 = load_quat(kR + 12);
            const Quat R0c = qconj(R0), R1c = qconj(R1), R2c = qconj(R2);
            LogJac dl01, dl12, dl23; ExpJac de0, de1, de2;
            double o01[3], o12[3], o23[3];
            qlog<WITH_J>(qmul(R0c, R1), o01, &dl01);
            qlog<WITH_J>(qmul(R1c, R2), o12, &dl12);
            qlog<WITH_J>(qmul(R2c, R3), o23, &dl23);
            for (int a = 0; a < 3; ++a) { o01[a] *= c1; o12[a] *= c2; o23[a] *= c3; }
            const Quat A0 = qexp<WITH_J>(o01, &de0);
            const Quat A1 = qexp<WITH_J>(o12, &de1);
            const Quat A2 = qexp<WITH_J>(o23, &de2);
            const Quat R0A0 = qmul(R0, A0);
            const Quat R0A0A1 = qmul(R0A0, A1);
            if (WITH_J)
            {
                const Quat A12 = qmul(A1, A2);
                const Quat A012 = qmul(qmul(A0, A1), A2);
                const Jac43 E0 = local_param_jac(R0), E1 = local_param_jac(R1);
                const Jac43 E2 = local_param_jac(R2), E3 = local_param_jac(R3);
                // knot 0
                {
                    Jac43 a = rmul(E0, A012);
                    Jac43 dA0 = through_log_exp(de0, c1, dl01, rmul(conj_cols(E0), R1));
                    Jac43 b = lmul(R0, rmul(dA0, A12));
                    store_block(JR, 12, 0, jadd(a, b));
                }
                // knot 1: through R01 (as right factor) and R12 (as conj left factor)
                {
                    Jac43 dA0 = through_log_exp(de0, c1, dl01, lmul(R0c, E1));
                    Jac43 a = lmul(R0, rmul(dA0, A12));
                    Jac43 dA1 = through_log_exp(de1, c2, dl12, rmul(conj_cols(E1), R2));
                    Jac43 b = lmul(R0A0, rmul(dA1, A2));
                    store_block(JR, 12, 3, jadd(a, b));
                }
                // knot 2
                {
                    Jac43 dA1 = through_log_exp(de1, c2, dl12, lmul(R1c, E2));
                    Jac43 a = lmul(R0A0, rmul(dA1, A2));
                    Jac43 dA2 = through_log_exp(de2, c3, dl23, rmul(conj_cols(E2), R3));
                    Jac43 b = lmul(R0A0A1, dA2);
                    store_block(JR, 12, 6, jadd(a, b));
                }
                // knot 3
                {
                    Jac43 dA2 = through_log_exp(de2, c3, dl23, lmul(R2c, E3));
                    store_block(JR, 12, 9, lmul(R0A0A1, dA2));
                }
            }
            return qmul(R0A0A1, A2);
        }
    }

    // Same spline sample as spline_rotation, but only columns col0 .. col0 + NC - 1 of the 4x3 Jacobian block of ONE
    // knot J (0 <= J < KDEG).  The pose-table kernel spreads the knots of a sample over separate waves and the
    // columns of a block over lanes (blocks and columns are independent given the shared logs / exps); with J a
    // compile-time constant only the log / exp Jacobians that knot needs are computed.  Per column the arithmetic is
    // the one of spline_rotation.
    template <int KDEG, int NC, int J>
    MBAVO_HD Quat spline_rotation_knot(const double *kR, double u, int col0, JacC<NC> &block)
    {
        if constexpr (KDEG == 2)
        {
            const Quat R0 = load_quat(kR), R1 = load_quat(kR + 4);
            const Quat R0c = qconj(R0);
            LogJac dl; ExpJac de;
            double om[3];
            qlog<true>(qmul(R0c, R1), om, &dl);
            om[0] *= u; om[1] *= u; om[2] *= u;
            const Quat A0 = qexp<true>(om, &de);
            if constexpr (J == 0)
            {
                const JacC<NC> E0 = local_param_jac_cols<NC>(R0, col0);
                block = jadd(rmul(E0, A0), lmul(R0, through_log_exp(de, u, dl, rmul(conj_cols(E0), R1))));
            }
            else
                block = lmul(R0, through_log_exp(de, u, dl, lmul(R0c, local_param_jac_cols<NC>(R1, col0))));
            return qmul(R0, A0);
        }
        else
        {
            const double uu = u * u, uuu = uu * u, s = 1. / 6.;
            const double c1 = 5 * s + 0.5 * u - 0.5 * uu + s * uuu;
            const double c2 = s + 0.5 * u + 0.5 * uu - 2 * s * uuu;
            const double c3 = s * uuu;
            const Quat R0 = load_quat(kR), R1 = load_quat(kR + 4), R2 = load_quat(kR + 8), R3 = load_quat(kR + 12);
            const Quat R0c = qconj(R0), R1c = qconj(R1), R2c = qconj(R2);
            constexpr bool J01 = J <= 1, J12 = J == 1 || J == 2, J23 = J >= 2; // which segments this knot differentiates
            LogJac dl01, dl12, dl23; ExpJac de0, de1, de2;
            double o01[3], o12[3], o23[3];
            qlog<J01>(qmul(R0c, R1), o01, &dl01);
            qlog<J12>(qmul(R1c, R2), o12, &dl12);
            qlog<J23>(qmul(R2c, R3), o23, &dl23);
            for (int a = 0; a < 3; ++a) { o01[a] *= c1; o12[a] *= c2; o23[a] *= c3; }
            const Quat A0 = qexp<J01>(o01, &de0);
            const Quat A1 = qexp<J12>(o12, &de1);
            const Quat A2 = qexp<J23>(o23, &de2);
            const Quat R0A0 = qmul(R0, A0);
            const Quat R0A0A1 = qmul(R0A0, A1);
            const Quat A12 = qmul(A1, A2);
            if constexpr (J == 0)
            {
                const JacC<NC> E0 = local_param_jac_cols<NC>(R0, col0);
                const Quat A012 = qmul(qmul(A0, A1), A2);
                const JacC<NC> dA0 = through_log_exp(de0, c1, dl01, rmul(conj_cols(E0), R1));
                block = jadd(rmul(E0, A012), lmul(R0, rmul(dA0, A12)));
            }
            else if constexpr (J == 1)
            {
                const JacC<NC> E1 = local_param_jac_cols<NC>(R1, col0);
                const JacC<NC> dA0 = through_log_exp(de0, c1, dl01, lmul(R0c, E1));
                const JacC<NC> dA1 = through_log_exp(de1, c2, dl12, rmul(conj_cols(E1), R2));
                block = jadd(lmul(R0, rmul(dA0, A12)), lmul(R0A0, rmul(dA1, A2)));
            }
            else if constexpr (J == 2)
            {
                const JacC<NC> E2 = local_param_jac_cols<NC>(R2, col0);
                const JacC<NC> dA1 = through_log_exp(de1, c2, dl12, lmul(R1c, E2));
                const JacC<NC> dA2 = through_log_exp(de2, c3, dl23, rmul(conj_cols(E2), R3));
                block = jadd(lmul(R0A0, rmul(dA1, A2)), lmul(R0A0A1, dA2));
            }
            else
            {
                const JacC<NC> dA2 = through_log_exp(de2, c3, dl23, lmul(R2c, local_param_jac_cols<NC>(R3, col0)));
                block = lmul(R0A0A1, dA2);
            }
            return qmul(R0A0A1, A2);
        }
    }

    // ---- the same spline sample in two STAGES (the pose kernels: engine.hip).  The cumulative factors of a sample,
    // A_g = exp(c_g * log(R_g^-1 R_{g+1})), are independent of each other and every Jacobian column needs them all, so
    // one lane per (sample, segment) evaluates a segment ONCE (stage A: the log, the exp and their Jacobians -- the
    // transcendental part, ~2/3 of the chain) and the (sample, knot, column) lanes pick the results up from shared memory
    // (stage B: the products).  Arithmetic per quantity is that of spline_rotation_knot, so the results are bit-identical;
    // the dependent chain per lane is one segment + the products instead of three segments + the products.
    struct SplineSeg
    {
        Quat A;    // exp(c * log(Ra^-1 Rb))
        LogJac dl; // d log / d quaternion   at Ra^-1 Rb
        ExpJac de; // d exp / d tangent      at c * log(Ra^-1 Rb)
    };
    // cumulative-spline weight of segment g at normalised time u (SplineFunctor.h:232-234; k = 2: the lerp weight u)
    template <int KDEG>
    MBAVO_HD double seg_weight(double u, int g)
    {
        if constexpr (KDEG == 2)
            return u;
        else
        {
            const double uu = u * u, uuu = uu * u, s = 1. / 6.;
            return g == 0 ? 5 * s + 0.5 * u - 0.5 * uu + s * uuu : (g == 1 ? s + 0.5 * u + 0.5 * uu - 2 * s * uuu : s * uuu);
        }
    }
    // On the device a REAL call (round 3): inlined into the fused kernels, which sit at their register budget, the shorter
    // chain of the fastm forms came back as 6 vector spills in k_fused<4, .., POSE> (+0.5 us per dense evaluation); as a call the
    // kernel keeps its 166 registers and the semi-dense single-launch kernel drops from 172 to 150 with no scalar spill
    // (configs[1] dense step 37.44 -> 36.68 us, semi-dense evaluation 18.26 -> 17.58, trackFrame 0.407 -> 0.394 ms per frame;
    // A/B in one call, profiles/r03_kfused_experiments.txt).
#if defined(__HIP_DEVICE_COMPILE__)
#define MBAVO_SEG_FN __device__ __noinline__
#else
#define MBAVO_SEG_FN MBAVO_HD
#endif
    template <bool WITH_J>
    MBAVO_SEG_FN void spline_segment_eval(const double *Ra4, const double *Rb4, double c, SplineSeg &out)
    {
        double om[3];
        qlog<WITH_J>(qmul(qconj(load_quat(Ra4)), load_quat(Rb4)), om, &out.dl);
        om[0] *= c; om[1] *= c; om[2] *= c;
        out.A = qexp<WITH_J>(om, &out.de);
    }
    // rotation of the sample from its evaluated segments (no Jacobian)
    template <int KDEG>
    MBAVO_HD Quat spline_rotation_from_segs(const double *kR, const SplineSeg *sg)
    {
        if constexpr (KDEG == 2)
            return qmul(load_quat(kR), sg[0].A);
        else
            return qmul(qmul(qmul(load_quat(kR), sg[0].A), sg[1].A), sg[2].A);
    }
    // stage B: columns col0 .. col0 + NC - 1 of knot J's 4x3 Jacobian block, from the evaluated segments
    template <int KDEG, int NC, int J>
    MBAVO_HD Quat spline_rotation_knot_from_segs(const double *kR, double u, int col0, const SplineSeg *sg, JacC<NC> &block)
    {
        if constexpr (KDEG == 2)
        {
            const Quat R0 = load_quat(kR), R1 = load_quat(kR + 4);
            const Quat R0c = qconj(R0);
            const Quat A0 = sg[0].A;
            if constexpr (J == 0)
            {
                const JacC<NC> E0 = local_param_jac_cols<NC>(R0, col0);
                block = jadd(rmul(E0, A0), lmul(R0, through_log_exp(sg[0].de, u, sg[0].dl, rmul(conj_cols(E0), R1))));
            }
            else
                block = lmul(R0, through_log_exp(sg[0].de, u, sg[0].dl, lmul(R0c, local_param_jac_cols<NC>(R1, col0))));
            return qmul(R0, A0);
        }
        else
        {
            const double c1 = seg_weight<4>(u, 0), c2 = seg_weight<4>(u, 1), c3 = seg_weight<4>(u, 2);
            const Quat R0 = load_quat(kR), R1 = load_quat(kR + 4), R2 = load_quat(kR + 8), R3 = load_quat(kR + 12);
            const Quat R0c = qconj(R0), R1c = qconj(R1), R2c = qconj(R2);
            const Quat A0 = sg[0].A, A1 = sg[1].A, A2 = sg[2].A;
            const Quat R0A0 = qmul(R0, A0);
            const Quat R0A0A1 = qmul(R0A0, A1);
            const Quat A12 = qmul(A1, A2);
            if constexpr (J == 0)
            {
                const JacC<NC> E0 = local_param_jac_cols<NC>(R0, col0);
                const Quat A012 = qmul(qmul(A0, A1), A2);
                const JacC<NC> dA0 = through_log_exp(sg[0].de, c1, sg[0].dl, rmul(conj_cols(E0), R1));
                block = jadd(rmul(E0, A012), lmul(R0, rmul(dA0, A12)));
            }
            else if constexpr (J == 1)
            {
                const JacC<NC> E1 = local_param_jac_cols<NC>(R1, col0);
                const JacC<NC> dA0 = through_log_exp(sg[0].de, c1, sg[0].dl, lmul(R0c, E1));
                const JacC<NC> dA1 = through_log_exp(sg[1].de, c2, sg[1].dl, rmul(conj_cols(E1), R2));
                block = jadd(lmul(R0, rmul(dA0, A12)), lmul(R0A0, rmul(dA1, A2)));
            }
            else if constexpr (J == 2)
            {
                const JacC<NC> E2 = local_param_jac_cols<NC>(R2, col0);
                const JacC<NC> dA1 = through_log_exp(sg[1].de, c2, sg[1].dl, lmul(R1c, E2));
                const JacC<NC> dA2 = through_log_exp(sg[2].de, c3, sg[2].dl, rmul(conj_cols(E2), R3));
                block = jadd(lmul(R0A0, rmul(dA1, A2)), lmul(R0A0A1, dA2));
            }
            else
            {
                const JacC<NC> dA2 = through_log_exp(sg[2].de, c3, sg[2].dl, lmul(R2c, local_param_jac_cols<NC>(R3, col0)));
                block = lmul(R0A0A1, dA2);
            }
            return qmul(R0A0A1, A2);
        }
    }

    // SO(3) exponential as a unit quaternion (what Spline.h:302,326 gets from
    // Sophus::SO3d::exp): series below theta^2 < 1e-20, closed form above.
    MBAVO_HD Quat so3_exp(const double om[3])
    {
        const double th2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
        double im, re;
        if (th2 < 1e-10 * 1e-10)
        {
            const double th4 = th2 * th2;
            im = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
            re = 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4;
        }
        else
        {
#if defined(MBAVO_POSE_FASTMATH) // (device: the batched LM's candidate step sits on the chain of every iteration -- the runtime's sqrt, sin, cos and division are ~430 instructions of it)
            double th, rth, hs, hc;
            fastm::sqrt_rsqrt(th2, th, rth);
            fastm::sincos(0.5 * th, hs, hc);
            im = hs * rth;
            re = hc;
#else
            const double th = sqrt(th2);
            im = sin(0.5 * th) / th;
            re = cos(0.5 * th);
#endif
        }
        return Quat{im * om[0], im * om[1], im * om[2], re};
    }
} // namespace mbavo

#endif
