// pixel_math.h -- per pixel-sample photometric warp, bilinear tap and Jacobian
// chain (host + device, header only).
//
// Computes what compute_pixel_intensity<double> + bilinear_interpolation<double>
// (ba_tracker/compute_pixel_intensity.h:25-209) and the per-sample chain of
// kernel_compute_pixel_jacobian_residual (compute_hessian_gradients_cost.cu:
// 123-153) compute, restructured for one-lane-per-pixel execution:
//   * everything that depends only on the pixel (unit ray, 1/(D+1e-8)) is hoisted
//     out of the sample loop; everything that depends only on the pose (rotation
//     matrix entries, spline weights, J_R) comes from a per-sample table;
//   * the warped point lies on the plane z = D of the keyframe, so P_z == D and
//     the reference's second division per sample disappears;
//   * the twelve dP/dq terms collapse to four dot products sharing m = C1*(dI.rho);
//   * J_t = kron(c, I3) (SplineFunctor.h:77-90), so the translation chain is
//     3k multiplies instead of a dense 3x3k product.
// The fp32 islands of the reference are kept bit-for-bit: sqrtf in the unit ray
// (:119) and float weights / float accumulation in the bilinear tap (:40-68),
// with FMA contraction disabled inside them.
#ifndef MBAVO_PIXEL_MATH_H
#define MBAVO_PIXEL_MATH_H

#include "core_types.h"
#include "se3_math.h"
#include <math.h>
#include <string.h>

// Image and table pointers arrive inside descriptor structs, so the compiler sees generic
// pointers and would emit flat_load_* (which also occupy the LDS queue).  On the device they are
// re-qualified as global-memory pointers.
#if defined(__HIP_DEVICE_COMPILE__)
#define MBAVO_GLOBAL __attribute__((address_space(1)))
#else
#define MBAVO_GLOBAL
#endif

namespace mbavo
{
    // Packed index e -> (i, j), i <= j, of the row-major upper triangle of an nd x nd matrix
    // (compute_hessian_gradients_cost.cu:217-229), in closed form: row = floor((2nd + 1 - sqrt((2nd + 1)^2 - 8e)) / 2)
    // in fp32, then one step of correction either way.  (A row-by-row search costs up to nd divergent iterations per
    // lane: 1.3 us of every workgroup's end-of-tile work in the fused kernel, measured.)  Checked against the search for
    // every e at nd = 12, 13, 24, 25, 37, 49, 96 (tests/test_host_logic.py restates it).
    MBAVO_HD void tri_decode(int e, int nd, int &i, int &j)
    {
        const float b = 2.0f * (float)nd + 1.0f;
        int r = (int)((b - sqrtf(b * b - 8.0f * (float)e)) * 0.5f);
        int start = r * nd - r * (r - 1) / 2;
        if (start > e)
        {
            --r;
            start = r * nd - r * (r - 1) / 2;
        }
        else if (e - start >= nd - r)
        {
            start += nd - r;
            ++r;
        }
        i = r;
        j = r + (e - start);
    }

    // One entry per (problem, frame, blur sample), written by the pose kernel.
    // Field order follows the sample loop's SCALAR loads (the table is read with wave-uniform addresses, 16 dwords per load
    // at most, and the kernel has ~100 SGPRs: it cannot fetch a whole entry ahead): what a sample's warp needs (R, t) is one
    // contiguous 96 bytes at the head of the entry -- two loads and two waits where t, R and c apart took three dependent
    // load-and-wait steps --, the Jacobian chain's c and A follow, and the two fields only the patch centre reads (from ONE
    // entry per frame) come last.  (Removing the table loads altogether, an ablation, takes 4 % off the dense kernel at S = 8
    // and 12 % at S = 16: most of that is the 40 doubles of c and A per sample, which no layout removes.)
    template <int KDEG>
    struct PoseEntry
    {
        double R[9];          // rotation matrix of q, row-major, reference term order
        double t[3];          // t_c2r
        double c[KDEG];       // translation spline weights (J_t = kron(c, I3))
        double A[9 * KDEG];   // 3 x 3k row-major: R * d(body-frame rotation of the pose)/d(knot local rotations), i.e. the
                              // derivative of the pose's rotation about the KEYFRAME's axes (see sample_retire)
        double rt[3];         // conj(q) applied to t: the pixel-independent half of patch_centre (read from sample S/2)
        double q[4];          // R_c2r xyzw
    };

    // The reference chains dI/dq (1x4) through J_R = dq/dw (4x3k).  I does not depend on |q| (the warped point
    // is (D - t_z) * rho / rho_z with rho = R_h(q) ray, homogeneous in q), so dI/dq is tangent to the unit sphere
    // at q and equals 2 * L3(q) * phi, where phi = dI/d(body-frame rotation) and L3(q) = first three columns of
    // the left-product matrix (Quaternion.h:239-260).  Hence dI/dq * J_R == phi * A with A = 2 * L3(q)^T * J_R:
    // a 3x3k table instead of 4x3k, and phi (below) is cheaper to form than dI/dq.
    // (The kernels go one step further and tabulate R * A: sample_retire then needs phi about the keyframe's axes, which
    // costs 9 instructions per pixel-sample instead of 18.  This function returns the body-frame A.)
    template <int KDEG>
    MBAVO_HD void tangent_jacobian(const double q[4], const double *JR /*4 x 3k*/, double *A /*3 x 3k*/)
    {
        const double x = q[0], y = q[1], z = q[2], w = q[3];
        const double L3[4][3] = {{w, -z, y}, {z, w, -x}, {-y, x, w}, {-x, -y, -z}}; // rows: qx qy qz qw
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3 * KDEG; ++c)
            {
                double v = L3[0][a] * JR[c];
                v += L3[1][a] * JR[3 * KDEG + c];
                v += L3[2][a] * JR[6 * KDEG + c];
                v += L3[3][a] * JR[9 * KDEG + c];
                A[a * 3 * KDEG + c] = 2.0 * v;
            }
    }

    // rotation matrix entries exactly as compute_pixel_intensity.h:124-126,168-177 forms them
    MBAVO_HD void rotation_entries(const double q[4], double R[9])
    {
        const double qx = q[0], qy = q[1], qz = q[2], qw = q[3];
        R[0] = qw * qw + qx * qx - qy * qy - qz * qz;
        R[1] = -2. * (qw * qz - qx * qy);
        R[2] = 2. * (qw * qy + qx * qz);
        R[3] = 2. * (qw * qz + qx * qy);
        R[4] = qw * qw - qx * qx + qy * qy - qz * qz;
        R[5] = -2. * (qw * qx - qy * qz);
        R[6] = -2. * (qw * qy - qx * qz);
        R[7] = 2. * (qw * qx + qy * qz);
        R[8] = qw * qw - qx * qx - qy * qy + qz * qz;
    }

    struct Camera
    {
        double fx, fy, cx, cy;
        int H, W;
    };

    // 1 / x for the per-sample depth scale: the runtime's IEEE division (v_div_scale x2, v_rcp, four FMAs, a multiply,
    // v_div_fmas, v_div_fixup = 11 instructions) without the range scaling and the special-case fix-up -- the same
    // Newton steps and the same final correction, so the SAME correctly rounded bits whenever x is a normal number
    // with a normal reciprocal (x = row 3 of R times a unit ray here: |x| <= 1).  x = 0 or subnormal gives NaN
    // instead of inf / a huge value; either way the sample's coordinates fail the bounds test.
    MBAVO_HD double reciprocal(double x)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        double r = __builtin_amdgcn_rcp(x);
        double e = __builtin_fma(-x, r, 1.0);
        r = __builtin_fma(r, e, r);
        e = __builtin_fma(-x, r, 1.0);
        r = __builtin_fma(r, e, r);
        e = __builtin_fma(-x, r, 1.0);
        return __builtin_fma(e, r, r);
#else
        return 1. / x;
#endif
    }

    // 1 / x to within one unit in the last place: v_rcp_f64 (2^-23) and two Newton steps, without reciprocal()'s final
    // correctly-rounding correction (two instructions).  For the per-SAMPLE depth scale only: what it feeds (the tap
    // coordinates) already differs from the reference's rounding by the fused multiply-adds of the rotation before it.
    MBAVO_HD double reciprocal_1ulp(double x)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        double r = __builtin_amdgcn_rcp(x);
        r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
        return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
#else
        return reciprocal(x);
#endif
    }

    // n / d the same way (the runtime's sequence minus v_div_scale x2 and v_div_fixup: 8 instructions instead of 11; the
    // same bits whenever d, n / d and the intermediate products are normal numbers or n is zero).  For the per-pixel
    // quotients with well-conditioned denominators only: focal lengths, the number of samples.
    MBAVO_HD double quotient(double n, double d)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        double r = __builtin_amdgcn_rcp(d);
        double e = __builtin_fma(-d, r, 1.0);
        r = __builtin_fma(r, e, r);
        e = __builtin_fma(-d, r, 1.0);
        r = __builtin_fma(r, e, r);
        const double q = n * r;
        return __builtin_fma(__builtin_fma(-d, q, n), r, q);
#else
        return n / d;
#endif
    }

    // quotient() in two halves, for a denominator shared by many quotients (the number of samples: wave-uniform, the caller
    // keeps both values in scalar registers): r = quotient_recip(d) once, then quotient_with(n, d, r) == quotient(n, d).
    MBAVO_HD double quotient_recip(double d)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        double r = __builtin_amdgcn_rcp(d);
        double e = __builtin_fma(-d, r, 1.0);
        r = __builtin_fma(r, e, r);
        e = __builtin_fma(-d, r, 1.0);
        return __builtin_fma(r, e, r);
#else
        return 1. / d;
#endif
    }
    MBAVO_HD double quotient_with(double n, double d, double r)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        const double q = n * r;
        return __builtin_fma(__builtin_fma(-d, q, n), r, q);
#else
        (void)r;
        return n / d;
#endif
    }

    // unit ray through an integer pixel; z uses the reference's fp32 sqrt (A4)
    MBAVO_HD void unit_ray(const Camera &cam, double px, double py, double ray[3])
    {
        double xh = quotient(px - cam.cx, cam.fx);
        double yh = quotient(py - cam.cy, cam.fy);
        const double zh = reciprocal((double)sqrtf((float)(1. + xh * xh + yh * yh)));
        ray[0] = xh * zh;
        ray[1] = yh * zh;
        ray[2] = zh;
    }

    // Bilinear tap of the u8 image and the interleaved float gradient image, split in two so
    // that the loads of sample s+1 can be in flight while sample s is consumed.
    // In-bounds test inclusive (0 <= x <= W-1); the 2x2 window is anchored at
    // min(floor, size-2) which is value-identical to the reference's zero-weight
    // taps at the last row/column and never reads outside the buffer (A6).
    struct __attribute__((packed, aligned(1))) UnalignedU16 { unsigned short v; };
    struct __attribute__((aligned(8))) Float4A8 { float v[4]; };
    struct __attribute__((aligned(4))) Half4A4 { unsigned short v[4]; };

    // IEEE half -> float (exact); subnormals and signed zeros included, inf/nan not needed for gradients
    MBAVO_HD float half_bits_to_float(unsigned short h)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        return (float)__builtin_bit_cast(_Float16, h);
#else
        const unsigned sign = (unsigned)(h & 0x8000u) << 16;
        const int e = (h >> 10) & 0x1f;
        const unsigned m = h & 0x3ffu;
        float v;
        if (e == 0) v = (float)m * 5.9604644775390625e-08f; // m * 2^-24
        else
        {
            unsigned bits = ((unsigned)(e + 112) << 23) | (m << 13);
            memcpy(&v, &bits, 4);
        }
        return sign ? -v : v;
#endif
    }

    struct TapLoads
    {
        float w00, w01, w10, w11;
        unsigned short r0, r1; // image rows y, y+1: bytes (x, x+1)
        float g0[4], g1[4];    // gradient rows: [dx(x) dy(x) dx(x+1) dy(x+1)]
        unsigned pk[4];        // packed keyframe (GRAD == 2): words of (x, y), (x+1, y), (x, y+1), (x+1, y+1)
        bool ok;
    };

    // Packed keyframe (mbavo_problem.grad_fp16 = 2, mbavo_pack_keyframe_u8): ONE 32-bit word per pixel holds the intensity and
    // both central differences of an 8-bit image -- bits 0-7 I, bits 8-16 2 dI/dx and bits 23-31 2 dI/dy as 9-bit two's
    // complement (|2 d| <= 255) -- so a bilinear tap with gradients is two 8-byte loads from ONE image instead of two 2-byte
    // loads from the u8 image and two 16-byte loads from the float gradient image (9 bytes per pixel in two images -> 4 in
    // one; per 8-pixel patch ~10 touched 128-byte lines instead of ~27).  Every value is recovered exactly: the fp32 blend
    // runs on the doubled differences and its result is halved, which commutes with every rounding of the blend.
    struct __attribute__((aligned(4))) Word2A4 { unsigned v[2]; };
    MBAVO_HD unsigned pack_keyframe_word(int I, int kx, int ky) { return (unsigned)I | ((unsigned)kx & 0x1ffu) << 8 | (unsigned)ky << 23; }

    // HALF_GRAD: the gradient image holds IEEE half pairs (4 B/pixel) instead of float pairs.  A compile-time
    // switch: a run-time branch around the loads makes the compiler wait for ALL outstanding loads (vmcnt(0)) at
    // the first use, which serialises the sample pipeline.
    template <bool WITH_GRAD, int HALF_GRAD = 0> // 0: float pairs, 1: IEEE half pairs, 2: packed keyframe words
    MBAVO_HD void tap_fetch(const unsigned char *__restrict__ I, const float *__restrict__ G, int H, int W,
                            double x, double y, TapLoads &t)
    {
#pragma clang fp contract(off)
        // Branch-free: an out-of-bounds (or NaN) coordinate only clears t.ok; its window is clamped into the image
        // so the loads stay legal and the garbage it produces is discarded by the caller.  Keeping the sample loop
        // free of divergent control flow lets the compiler overlap the issue of sample s+1 with the retire of s.
        t.ok = x >= 0 && x <= W - 1 && y >= 0 && y <= H - 1; // (ordered compares: false for NaN, four instructions instead of five)
#if defined(__HIP_DEVICE_COMPILE__)
        // the conversion saturates and maps NaN to 0 (v_cvt_i32_f64), so the window is clamped on the integer side
        // (one v_med3_i32 per axis) instead of zeroing the coordinates of a dropped sample first
        int xi = __double2int_rz(x), yi = __double2int_rz(y);
        // clamp to [0, W - 2] x [0, H - 2] (W, H: wave-uniform image size; the compiler only forms med3 for constants)
        asm("v_med3_i32 %0, %1, 0, %2" : "=v"(xi) : "v"(xi), "s"(W - 2));
        asm("v_med3_i32 %0, %1, 0, %2" : "=v"(yi) : "v"(yi), "s"(H - 2));
        const float dx = (float)(x - xi);
        const float dy = (float)(y - yi);
#else
        const double xs = t.ok ? x : 0.0, ys = t.ok ? y : 0.0;
        int xi = (int)xs, yi = (int)ys;
        xi = xi > W - 2 ? W - 2 : xi;
        yi = yi > H - 2 ? H - 2 : yi;
        const float dx = (float)(xs - xi);
        const float dy = (float)(ys - yi);
#endif
        const float dxdy = dx * dy;
        t.w00 = 1.0f - dx - dy + dxdy;
        t.w01 = dx - dxdy;
        t.w10 = dy - dxdy;
        t.w11 = dxdy;
        // Unsigned 32-bit byte offsets from the (wave-uniform) image bases: the loads take the base from SGPRs and a
        // 32-bit VGPR offset, with no 64-bit address arithmetic (images of up to 2^29 pixels).
#if defined(MBAVO_EXP_NO_TAPS) // timing experiment only: every tap hits the same address
        const unsigned idx = (unsigned)((xi + yi) & 1);
#else
        const unsigned idx = (unsigned)(yi * W + xi);
#endif
        const unsigned idx1 = idx + (unsigned)W;
        const MBAVO_GLOBAL unsigned char *Ig = (const MBAVO_GLOBAL unsigned char *)I;
        if (WITH_GRAD && HALF_GRAD == 2)
        { // everything a tap needs is in the packed image (cost-only passes keep to the u8 image: 2 bytes a row)
            const MBAVO_GLOBAL unsigned char *Gb = (const MBAVO_GLOBAL unsigned char *)G;
            const Word2A4 a = *(const MBAVO_GLOBAL Word2A4 *)(Gb + idx * 4u);
            const Word2A4 b = *(const MBAVO_GLOBAL Word2A4 *)(Gb + idx1 * 4u);
            t.pk[0] = a.v[0]; t.pk[1] = a.v[1]; t.pk[2] = b.v[0]; t.pk[3] = b.v[1];
            return;
        }
        t.r0 = ((const MBAVO_GLOBAL UnalignedU16 *)(Ig + idx))->v;
        t.r1 = ((const MBAVO_GLOBAL UnalignedU16 *)(Ig + idx1))->v;
        if (WITH_GRAD)
        {
            const MBAVO_GLOBAL unsigned char *Gb = (const MBAVO_GLOBAL unsigned char *)G;
            if (HALF_GRAD == 1)
            { // 8 bytes per row pair instead of 16
                const Half4A4 a = *(const MBAVO_GLOBAL Half4A4 *)(Gb + idx * 4u);
                const Half4A4 b = *(const MBAVO_GLOBAL Half4A4 *)(Gb + idx1 * 4u);
                for (int i = 0; i < 4; ++i) { t.g0[i] = half_bits_to_float(a.v[i]); t.g1[i] = half_bits_to_float(b.v[i]); }
            }
            else
            {
                const Float4A8 a = *(const MBAVO_GLOBAL Float4A8 *)(Gb + idx * 8u);
                const Float4A8 b = *(const MBAVO_GLOBAL Float4A8 *)(Gb + idx1 * 8u);
                for (int i = 0; i < 4; ++i) { t.g0[i] = a.v[i]; t.g1[i] = b.v[i]; }
            }
        }
    }

    // float weights, float accumulation, order w11*I11 + w10*I10 + w01*I01 + w00*I00 (:56-68)
    template <bool WITH_GRAD, int GRAD = 0>
    MBAVO_HD void tap_blend(const TapLoads &t, double &val, double &gx, double &gy)
    {
#pragma clang fp contract(off)
        if (WITH_GRAD && GRAD == 2)
        { // packed keyframe words: the same blend on the doubled differences, halved at the end (exact)
            const float i00 = (float)(t.pk[0] & 0xffu), i01 = (float)(t.pk[1] & 0xffu);
            const float i10 = (float)(t.pk[2] & 0xffu), i11 = (float)(t.pk[3] & 0xffu);
            float v = t.w11 * i11;
            v = v + t.w10 * i10;
            v = v + t.w01 * i01;
            v = v + t.w00 * i00;
            val = (double)v;
            // (9-bit fields: shift the field's top bit to bit 31, arithmetic shift back)
            const float x00 = (float)((int)(t.pk[0] << 15) >> 23), x01 = (float)((int)(t.pk[1] << 15) >> 23);
            const float x10 = (float)((int)(t.pk[2] << 15) >> 23), x11 = (float)((int)(t.pk[3] << 15) >> 23);
            const float y00 = (float)((int)t.pk[0] >> 23), y01 = (float)((int)t.pk[1] >> 23);
            const float y10 = (float)((int)t.pk[2] >> 23), y11 = (float)((int)t.pk[3] >> 23);
#if defined(__HIP_DEVICE_COMPILE__)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 g11 = {x11, y11}, g10 = {x10, y10}, g01 = {x01, y01}, g00 = {x00, y00};
            f32x2 ab = t.w11 * g11;
            ab = ab + t.w10 * g10;
            ab = ab + t.w01 * g01;
            ab = ab + t.w00 * g00;
            ab = 0.5f * ab;
            gx = (double)ab.x;
            gy = (double)ab.y;
#else
            float a = t.w11 * x11;
            a = a + t.w10 * x10;
            a = a + t.w01 * x01;
            a = a + t.w00 * x00;
            float b = t.w11 * y11;
            b = b + t.w10 * y10;
            b = b + t.w01 * y01;
            b = b + t.w00 * y00;
            gx = (double)(0.5f * a);
            gy = (double)(0.5f * b);
#endif
            return;
        }
        const float i00 = (float)(t.r0 & 0xff), i01 = (float)(t.r0 >> 8);
        const float i10 = (float)(t.r1 & 0xff), i11 = (float)(t.r1 >> 8);
        float v = t.w11 * i11;
        v = v + t.w10 * i10;
        v = v + t.w01 * i01;
        v = v + t.w00 * i00;
        val = (double)v;
        if (WITH_GRAD)
        {
#if defined(__HIP_DEVICE_COMPILE__)
            // the two gradient blends are the same four steps on (dx, dy) pairs that sit in adjacent registers as loaded:
            // packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32, the weight broadcast by op_sel) round each half exactly
            // like the scalar ones -- 7 instructions instead of 14
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 g11 = {t.g1[2], t.g1[3]}, g10 = {t.g1[0], t.g1[1]}, g01 = {t.g0[2], t.g0[3]}, g00 = {t.g0[0], t.g0[1]};
            f32x2 ab = t.w11 * g11;
            ab = ab + t.w10 * g10;
            ab = ab + t.w01 * g01;
            ab = ab + t.w00 * g00;
            gx = (double)ab.x;
            gy = (double)ab.y;
#else
            float a = t.w11 * t.g1[2];
            a = a + t.w10 * t.g1[0];
            a = a + t.w01 * t.g0[2];
            a = a + t.w00 * t.g0[0];
            float b = t.w11 * t.g1[3];
            b = b + t.w10 * t.g1[1];
            b = b + t.w01 * t.g0[3];
            b = b + t.w00 * t.g0[1];
            gx = (double)a;
            gy = (double)b;
#endif
        }
    }

    template <bool WITH_GRAD>
    MBAVO_HD bool bilinear_tap(const unsigned char *__restrict__ I, const float *__restrict__ G,
                               int H, int W, double x, double y, double &val, double &gx, double &gy)
    {
        TapLoads t;
        tap_fetch<WITH_GRAD>(I, G, H, W, x, y, t);
        if (!t.ok) return false;
        tap_blend<WITH_GRAD>(t, val, gx, gy);
        return true;
    }

    // One blur sample of one pixel: interpolated keyframe intensity and (WITH_J) its
    // derivative w.r.t. the sample pose, jt = dI/dt (1x3) and b = dI/dq (1x4, xyzw) --
    // the 1x7 Jacobian of compute_pixel_intensity.h:155-207.
    // Returns false when the warp leaves the keyframe image.
    template <bool WITH_J>
    MBAVO_HD bool sample_eval(const double t[3], const double q[4], const double R[9], const double ray[3],
                              double D, double iz, const Camera &cam, const unsigned char *__restrict__ I,
                              const float *__restrict__ G, double &val, double jt[3], double b[4])
    {
        const double rx = R[0] * ray[0] + R[1] * ray[1] + R[2] * ray[2];
        const double ry = R[3] * ray[0] + R[4] * ray[1] + R[5] * ray[2];
        const double rz = R[6] * ray[0] + R[7] * ray[1] + R[8] * ray[2]; // == lambda (:124-126)
        const double C1 = 1. / rz;
        const double sc = (D - t[2]) * C1;
        const double Px = sc * rx + t[0];
        const double Py = sc * ry + t[1];
        const double u = cam.fx * (Px * iz) + cam.cx;
        const double v = cam.fy * (Py * iz) + cam.cy;
        double gx = 0, gy = 0;
        if (!bilinear_tap<WITH_J>(I, G, cam.H, cam.W, u, v, val, gx, gy)) return false;
        if (WITH_J)
        {
            const double dIx = gx * iz * cam.fx;
            const double dIy = gy * iz * cam.fy;
            const double dIz = -iz * iz * (gx * Px * cam.fx + gy * Py * cam.fy);
            const double dxy = dIx * rx + dIy * ry;
            const double m = C1 * (dxy + dIz * rz);
            jt[0] = dIx;
            jt[1] = dIy;
            jt[2] = -C1 * dxy; // dIz*(1 - rz*C1) vanishes identically (:199)
            const double qx = q[0], qy = q[1], qz = q[2], qw = q[3];
            const double T0 = qx * ray[0] + qy * ray[1] + qz * ray[2];
            const double T3 = qw * ray[0] + qy * ray[2] - qz * ray[1];
            const double T2 = qw * ray[1] - qx * ray[2] + qz * ray[0];
            const double T4 = qw * ray[2] + qx * ray[1] - qy * ray[0];
            const double g2 = 2. * sc;
            b[0] = g2 * (dIx * T0 - dIy * T4 + dIz * T2 - T2 * m);
            b[1] = g2 * (dIx * T4 + dIy * T0 - dIz * T3 + T3 * m);
            b[2] = g2 * (-dIx * T2 + dIy * T3 + dIz * T0 - T0 * m);
            b[3] = g2 * (dIx * T3 + dIy * T2 + dIz * T4 - T4 * m);
        }
        return true;
    }

    // patch centre of a keypoint in the current frame at the mid-exposure pose
    // (compute_local_patches_xy.cu:19-49)
    // R_r2c * t_c2r with R_r2c = conj(R_c2r): the part of patch_centre that does not depend on the keypoint (the pose
    // kernel stores it in the table entry; 56 of patch_centre's ~130 instructions)
    MBAVO_HD void rotated_translation(const double t_c2r[3], const double q_c2r[4], double rt[3])
    {
#pragma clang fp contract(off)
        const Quat r2c = qconj(Quat{q_c2r[0], q_c2r[1], q_c2r[2], q_c2r[3]});
        qrotate(r2c, t_c2r, rt);
    }

    MBAVO_HD void patch_centre_rt(const double rt[3], const double q_c2r[4], double kx, double ky, double kz,
                                  const Camera &cam, double &ox, double &oy)
    {
#pragma clang fp contract(off)
        const double P[3] = {quotient(kz * (kx - cam.cx), cam.fx), quotient(kz * (ky - cam.cy), cam.fy), kz};
        // R_r2c = conj(R_c2r); t_r2c = -(R_r2c * t_c2r); P_c = R_r2c * P + t_r2c
        const Quat r2c = qconj(Quat{q_c2r[0], q_c2r[1], q_c2r[2], q_c2r[3]});
        double rp[3];
        qrotate(r2c, P, rp);
        const double X = rp[0] - rt[0], Y = rp[1] - rt[1], Z = rp[2] - rt[2];
        ox = X / Z * cam.fx + cam.cx;
        oy = Y / Z * cam.fy + cam.cy;
    }

    // The same centre through the rotation MATRIX, fused multiply-adds and reciprocals (~30 instructions against ~110
    // for the quaternion sandwich rounded operation by operation as the reference does it).  The two agree to ~1e-11 pixels;
    // what the callers need is the TRUNCATED coordinate (A3), which can only differ when the value lies that close to an
    // integer: a lane whose coordinate (centre + pattern offset) is farther than 1e-5 from every integer may keep this
    // value, otherwise the caller evaluates patch_centre_rt (see patch_centre_sure).  R = rotation_entries(q_c2r),
    // inv_fx = 1 / fx, inv_fy = 1 / fy.
    MBAVO_HD void patch_centre_fast(const double rt[3], const double R[9], double kx, double ky, double kz, const Camera &cam,
                                    double inv_fx, double inv_fy, double &ox, double &oy)
    {
        const double Px = kz * (kx - cam.cx) * inv_fx, Py = kz * (ky - cam.cy) * inv_fy;
        // R_r2c = R^T
        const double X = R[0] * Px + R[3] * Py + R[6] * kz - rt[0];
        const double Y = R[1] * Px + R[4] * Py + R[7] * kz - rt[1];
        const double Z = R[2] * Px + R[5] * Py + R[8] * kz - rt[2];
#if defined(__HIP_DEVICE_COMPILE__)
        double r = __builtin_amdgcn_rcp(Z);
        r = __builtin_fma(r, __builtin_fma(-Z, r, 1.0), r);
        r = __builtin_fma(r, __builtin_fma(-Z, r, 1.0), r);
#else
        const double r = 1.0 / Z;
#endif
        ox = X * r * cam.fx + cam.cx;
        oy = Y * r * cam.fy + cam.cy;
    }
    // true when truncating v (a coordinate computed to ~1e-11) is safe: v is farther than 1e-5 from every integer.
    // NaN, infinities and |v| >= 2^52 (no fraction left) give false.
    MBAVO_HD bool patch_centre_sure(double v)
    {
        const double dist = v - __builtin_rint(v);
        return (dist < 0 ? -dist : dist) > 1e-5;
    }

    MBAVO_HD void patch_centre(const double t_c2r[3], const double q_c2r[4], double kx, double ky, double kz,
                               const Camera &cam, double &ox, double &oy)
    {
        double rt[3];
        rotated_translation(t_c2r, q_c2r, rt);
        patch_centre_rt(rt, q_c2r, kx, ky, kz, cam, ox, oy);
    }

    // Huber weight sqrt(rho') and rho for residual r (compute_hessian_gradients_cost.cu:189-199);
    // the reference's fp32 square roots are kept (A11).
    MBAVO_HD void huber_weight(double r, double a, double &w, double &rho)
    {
#pragma clang fp contract(off) // rho = 2 a sqrt(x) - a^2 as written: per-pixel costs equal the oracle's bit for bit
        const double aa = a * a;
        const double x = 0.5 * r * r;
        w = 1.;
        rho = x;
        if (x > aa)
        {
            const double sx = (double)sqrtf((float)x);
            w = (double)sqrtf((float)(a / (sx + 1e-8)));
            rho = 2 * a * sx - aa;
        }
    }

    // Sum of a patch's P Huber costs in the order of the reference's shared-memory reduce() (reduction.h:13-55):
    // power-of-two P -> the stride-halving tree b[i] += b[i + s], s = P/2 .. 1; any other P -> ascending (A10: the
    // reference drops elements there, the oracle defines the plain sum).  b_m[i] = b_2m[i] + b_2m[i + m] with
    // b_P[i] = r[i], so the total is val(0, 1) of the recursion below: per-patch costs equal the oracle's bit for bit.
    template <int N, int M = 1, class PtrT = const double *>
    MBAVO_HD double patch_tree(PtrT r, int i)
    {
        if constexpr (M == N)
            return r[i];
        else
            return patch_tree<N, 2 * M, PtrT>(r, i) + patch_tree<N, 2 * M, PtrT>(r, i + M);
    }

    template <class PtrT = const double *>
    MBAVO_HD double patch_rho_sum(PtrT r, int P)
    {
        if ((P & (P - 1)) == 0)
        {
            switch (P)
            {
            case 1: return r[0];
            case 2: return patch_tree<2, 1, PtrT>(r, 0);
            case 4: return patch_tree<4, 1, PtrT>(r, 0);
            case 8: return patch_tree<8, 1, PtrT>(r, 0);
            case 16: return patch_tree<16, 1, PtrT>(r, 0);
            default: break;
            }
            // larger power of two: the same tree is the adjacent-pairs tree over the bit-reversed index sequence;
            // evaluated as a stream with one pending partial sum per level (a binary counter), levels in registers
            int logp = 0;
            while ((1 << logp) < P) ++logp;
            constexpr int MAXL = 24;
            double lv[MAXL];
#pragma unroll
            for (int l = 0; l < MAXL; ++l) lv[l] = 0.0;
            double x = 0.0;
            for (unsigned idx = 0; idx < (unsigned)P; ++idx)
            {
                unsigned rev = 0;
                for (int b = 0; b < logp; ++b) rev |= ((idx >> b) & 1u) << (logp - 1 - b);
                x = r[rev];
                bool carrying = true;
#pragma unroll
                for (int l = 0; l < MAXL; ++l)
                {
                    if (carrying)
                    {
                        if ((idx >> l) & 1u)
                            x = lv[l] + x; // the earlier (lower-index) half is the left operand, as b[i] + b[i + s]
                        else
                        {
                            lv[l] = x;
                            carrying = false;
                        }
                    }
                }
            }
            return x; // idx = P - 1 is all ones: the carry ran through every level
        }
        double acc = 0.0;
        for (int p = 0; p < P; ++p) acc += r[p];
        return acc;
    }

    // geometry of one sample up to the tap address, with the tap loads issued
    struct SampleInFlight
    {
        double C1, sc, rx, ry;
        TapLoads taps;
    };

    template <int KDEG, bool WITH_J, int HALF_GRAD = 0>
    MBAVO_HD void sample_issue(const PoseEntry<KDEG> &pe, const double ray[3], double D, double iz, const Camera &cam,
                               const unsigned char *__restrict__ I, const float *__restrict__ G, SampleInFlight &f)
    {
        const double *R = pe.R;
        f.rx = R[0] * ray[0] + R[1] * ray[1] + R[2] * ray[2];
        f.ry = R[3] * ray[0] + R[4] * ray[1] + R[5] * ray[2];
        const double rz = R[6] * ray[0] + R[7] * ray[1] + R[8] * ray[2];
        f.C1 = reciprocal_1ulp(rz);
        f.sc = (D - pe.t[2]) * f.C1;
        double Px, Py;
        { // product and sum as written (as the oracle rounds them): the translation is a scalar operand of the add,
          // where the fused form first copies it into the accumulator's VGPRs (two moves each)
#pragma clang fp contract(off)
            Px = f.sc * f.rx + pe.t[0];
            Py = f.sc * f.ry + pe.t[1];
        }
        // iz * f is per pixel (the retire half needs it anyway): one fused instruction per coordinate
        const double u = (iz * cam.fx) * Px + cam.cx;
        const double v = (iz * cam.fy) * Py + cam.cy;
        tap_fetch<WITH_J, HALF_GRAD>(I, G, cam.H, cam.W, u, v, f.taps);
    }

    // FIRST: the pixel's first sample SETS isum and Jrow instead of adding to them (no zero-initialisation of the 6k
    // accumulators per pixel).
    template <int KDEG, bool WITH_J, bool FIRST = false, int GRAD = 0>
    MBAVO_HD void sample_retire(const PoseEntry<KDEG> &pe, const SampleInFlight &f, const double ray[3], double D, double iz,
                                const Camera &cam, double &isum, double *Jrow)
    {
        double val, gx = 0, gy = 0;
        tap_blend<WITH_J, GRAD>(f.taps, val, gx, gy);
        isum = FIRST ? val : isum + val;
        if (WITH_J)
        {
            const double *R = pe.R;
            const double rx = f.rx, ry = f.ry;
            // dI/dt (compute_pixel_intensity.h:197-199; the dI/dP_z term of [2] cancels identically)
            const double dIx = gx * (iz * cam.fx); // iz * f is per pixel: hoisted out of the sample loop by the compiler
            const double dIy = gy * (iz * cam.fy);
            const double jt[3] = {dIx, dIy, -f.C1 * (dIx * rx + dIy * ry)};
            // phi = dI/d(rotation about the KEYFRAME's axes) = sc * (R ray) x dI/dt.  The derivative w.r.t. the body
            // rotation is sc * ray x (R^T dI/dt) = R^T phi (a rotation carries a cross product along); its product with
            // A = d(body rotation)/d(knots) is phi^T (R A), and R A is what the table holds: the nine-term transposed
            // product per pixel-sample became one per table column.  sc * (R ray)_z = D - t_z, so no third ray entry is
            // kept: 9 instructions instead of 18.  (The reference forms the same derivative as twelve dP/dq terms,
            // :179-206.)
            (void)R;
            const double dz = D - pe.t[2], sj = f.sc * jt[2];
            const double phi[3] = {ry * sj - dz * jt[1], dz * jt[0] - rx * sj, f.sc * (rx * jt[1] - ry * jt[0])};
#pragma unroll
            for (int j = 0; j < KDEG; ++j)
            {
#pragma unroll
                for (int c = 0; c < 3; ++c) Jrow[3 * j + c] = FIRST ? pe.c[j] * jt[c] : Jrow[3 * j + c] + pe.c[j] * jt[c];
            }
#pragma unroll
            for (int cidx = 0; cidx < 3 * KDEG; ++cidx)
            { // three FMAs into the accumulator (one instruction less per entry than forming the sample's term first)
                double a = FIRST ? phi[0] * pe.A[cidx] : Jrow[3 * KDEG + cidx] + phi[0] * pe.A[cidx];
                a += phi[1] * pe.A[3 * KDEG + cidx];
                a += phi[2] * pe.A[6 * KDEG + cidx];
                Jrow[3 * KDEG + cidx] = a;
            }
        }
    }

    // Residual and 1 x 6k Jacobian of one pixel over its S blur samples.
    // A pixel is valid iff its integer location and all S warps are in bounds.  Returns true for a valid
    // pixel, with the residual, Jrow = the SUM of the samples' Jacobian rows and inv_S = 1 / float(S) (the mean is
    // inv_S * Jrow: the caller folds the factor into the Huber weight it scales the row with anyway).  For an invalid
    // pixel: false, residual = 0, and Jrow UNDEFINED (possibly non-finite: it must not be used, not even times zero).
    // `table` points at the S entries of this pixel's frame.  The sample loop is software pipelined: the tap loads
    // of sample s+1 are issued before sample s is consumed.
    // (pixel_row_sum: the sum of the S interpolated intensities and the current image's pixel; pixel_row below forms the
    // residual from them)
    template <int KDEG, bool WITH_J, int HALF_GRAD = 0>
    MBAVO_HD bool pixel_row_sum(const PoseEntry<KDEG> *__restrict__ table, int S, const Camera &cam,
                                const unsigned char *__restrict__ I_ref, const float *__restrict__ G_ref,
                                const unsigned char *__restrict__ I_cur, double centre_x, double centre_y,
                                double depth, int dx, int dy, double &isum_out, double *Jrow, double &cur_out)
    {
        const int px = (int)(centre_x + dx); // truncation, A3
        const int py = (int)(centre_y + dy);
        if (px < 0 || px > cam.W - 1 || py < 0 || py > cam.H - 1) return false;
        const double cur = (double)((const MBAVO_GLOBAL unsigned char *)I_cur)[py * cam.W + px];
        double ray[3];
        unit_ray(cam, (double)px, (double)py, ray);
        const double iz = reciprocal(depth + 1e-8); // P_z == plane depth, A7
        double isum;
        bool ok;
        // Samples are processed in pairs: both samples' tap loads are issued, then both are retired, so the loads of
        // the second overlap the arithmetic of the first and vice versa.  No load is left outstanding across the loop
        // back-edge: the compiler's s_waitcnt insertion cannot count loop-carried loads and falls back to vmcnt(0),
        // which would serialise every sample (measured on the ping-pong-across-iterations variant).
        SampleInFlight fa, fb;
#define MBAVO_TAB(i) table[i]
        int s;
#ifndef MBAVO_COST_GROUP
#define MBAVO_COST_GROUP 4
#endif
        if constexpr (!WITH_J && MBAVO_COST_GROUP > 2)
        { // COST-ONLY passes (round 5): a sample is ~60 instructions between its taps and there is no Jacobian state to keep, so the taps
          // of FOUR samples are in flight at once (7 registers each: four blend weights, two rows, the flag) -- the exposed memory
          // latencies of a round are S / 4 instead of S / 2.  Same arithmetic per sample, intensities added in sample order.
            constexpr int G = MBAVO_COST_GROUP;
            SampleInFlight f[G];
            ok = true;
            isum = 0.0;
            bool first = true;
            for (s = 0; s + G - 1 < S; s += G)
            {
#pragma unroll
                for (int j = 0; j < G; ++j) sample_issue<KDEG, false, HALF_GRAD>(MBAVO_TAB(s + j), ray, depth, iz, cam, I_ref, G_ref, f[j]);
#pragma unroll
                for (int j = 0; j < G; ++j)
                {
                    ok = ok && f[j].taps.ok;
                    if (first && j == 0) sample_retire<KDEG, false, true, HALF_GRAD>(MBAVO_TAB(s), f[0], ray, depth, iz, cam, isum, Jrow);
                    else sample_retire<KDEG, false, false, HALF_GRAD>(MBAVO_TAB(s + j), f[j], ray, depth, iz, cam, isum, Jrow);
                }
                first = false;
            }
            for (; s + 1 < S; s += 2)
            { // the remainder (S < 4 or S not a multiple of the group) in PAIRS, as the with-Jacobian path below issues them: two
              // samples' taps in flight (one at a time serialised S = 2, 3 and the tails); the intensities are still
              // added in sample order
                sample_issue<KDEG, false, HALF_GRAD>(MBAVO_TAB(s), ray, depth, iz, cam, I_ref, G_ref, fa);
                sample_issue<KDEG, false, HALF_GRAD>(MBAVO_TAB(s + 1), ray, depth, iz, cam, I_ref, G_ref, fb);
                ok = ok && fa.taps.ok && fb.taps.ok;
                if (first) sample_retire<KDEG, false, true, HALF_GRAD>(MBAVO_TAB(s), fa, ray, depth, iz, cam, isum, Jrow);
                else sample_retire<KDEG, false, false, HALF_GRAD>(MBAVO_TAB(s), fa, ray, depth, iz, cam, isum, Jrow);
                sample_retire<KDEG, false, false, HALF_GRAD>(MBAVO_TAB(s + 1), fb, ray, depth, iz, cam, isum, Jrow);
                first = false;
            }
            if (s < S)
            { // odd S: the last sample alone
                sample_issue<KDEG, false, HALF_GRAD>(MBAVO_TAB(s), ray, depth, iz, cam, I_ref, G_ref, fa);
                ok = ok && fa.taps.ok;
                if (first) sample_retire<KDEG, false, true, HALF_GRAD>(MBAVO_TAB(s), fa, ray, depth, iz, cam, isum, Jrow);
                else sample_retire<KDEG, false, false, HALF_GRAD>(MBAVO_TAB(s), fa, ray, depth, iz, cam, isum, Jrow);
            }
        }
        else if (S >= 2)
        { // the first pair sets the accumulators
            sample_issue<KDEG, WITH_J, HALF_GRAD>(MBAVO_TAB(0), ray, depth, iz, cam, I_ref, G_ref, fa);
            sample_issue<KDEG, WITH_J, HALF_GRAD>(MBAVO_TAB(1), ray, depth, iz, cam, I_ref, G_ref, fb);
            ok = fa.taps.ok && fb.taps.ok;
            sample_retire<KDEG, WITH_J, true, HALF_GRAD>(MBAVO_TAB(0), fa, ray, depth, iz, cam, isum, Jrow);
            sample_retire<KDEG, WITH_J, false, HALF_GRAD>(MBAVO_TAB(1), fb, ray, depth, iz, cam, isum, Jrow);
            for (s = 2; s + 1 < S; s += 2)
            {
                sample_issue<KDEG, WITH_J, HALF_GRAD>(MBAVO_TAB(s), ray, depth, iz, cam, I_ref, G_ref, fa);
                sample_issue<KDEG, WITH_J, HALF_GRAD>(MBAVO_TAB(s + 1), ray, depth, iz, cam, I_ref, G_ref, fb);
                ok = ok && fa.taps.ok && fb.taps.ok;
                sample_retire<KDEG, WITH_J, false, HALF_GRAD>(MBAVO_TAB(s), fa, ray, depth, iz, cam, isum, Jrow);
                sample_retire<KDEG, WITH_J, false, HALF_GRAD>(MBAVO_TAB(s + 1), fb, ray, depth, iz, cam, isum, Jrow);
            }
            if (s < S)
            { // odd S
                sample_issue<KDEG, WITH_J, HALF_GRAD>(MBAVO_TAB(s), ray, depth, iz, cam, I_ref, G_ref, fa);
                ok = ok && fa.taps.ok;
                sample_retire<KDEG, WITH_J, false, HALF_GRAD>(MBAVO_TAB(s), fa, ray, depth, iz, cam, isum, Jrow);
            }
        }
        else
        { // the sharp case S = 1
            sample_issue<KDEG, WITH_J, HALF_GRAD>(MBAVO_TAB(0), ray, depth, iz, cam, I_ref, G_ref, fa);
            ok = fa.taps.ok;
            sample_retire<KDEG, WITH_J, true, HALF_GRAD>(MBAVO_TAB(0), fa, ray, depth, iz, cam, isum, Jrow);
        }
        if (!ok) return false;
        isum_out = isum;
        cur_out = cur;
        return true;
    }

    template <int KDEG, bool WITH_J, int HALF_GRAD = 0>
    MBAVO_HD bool pixel_row(const PoseEntry<KDEG> *__restrict__ table, int S, const Camera &cam,
                            const unsigned char *__restrict__ I_ref, const float *__restrict__ G_ref,
                            const unsigned char *__restrict__ I_cur, double centre_x, double centre_y,
                            double depth, int dx, int dy, double &residual, double *Jrow, double &inv_S)
    {
        double isum, cur;
        residual = 0.0;
        if (!pixel_row_sum<KDEG, WITH_J, HALF_GRAD>(table, S, cam, I_ref, G_ref, I_cur, centre_x, centre_y, depth, dx, dy, isum, Jrow, cur))
            return false;
        const double fS = (double)(float)S; // A8
        residual = quotient(isum, fS) - cur;
        inv_S = 1.0 / fS;
        return true;
    }

    // The same with the sample count's constants handed in: fS = (double)(float)S, rS = quotient_recip(fS), both kept in
    // scalar registers by the caller (formed per call they are hoisted into VGPR pairs that stay live through the whole
    // kernel).  The mean's factor 1 / fS is the caller's business.
    template <int KDEG, bool WITH_J, int HALF_GRAD = 0>
    MBAVO_HD bool pixel_row(const PoseEntry<KDEG> *__restrict__ table, int S, const Camera &cam,
                            const unsigned char *__restrict__ I_ref, const float *__restrict__ G_ref,
                            const unsigned char *__restrict__ I_cur, double centre_x, double centre_y,
                            double depth, int dx, int dy, double &residual, double *Jrow, double fS, double rS)
    {
        double isum, cur;
        residual = 0.0;
        if (!pixel_row_sum<KDEG, WITH_J, HALF_GRAD>(table, S, cam, I_ref, G_ref, I_cur, centre_x, centre_y, depth, dx, dy, isum, Jrow, cur))
            return false;
        residual = quotient_with(isum, fS, rS) - cur;
        return true;
    }
} // namespace mbavo

#endif
