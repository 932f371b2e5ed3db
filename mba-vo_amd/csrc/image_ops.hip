// image_ops.hip -- keyframe input producers on device: 2x2 box pyramid level and
// central-difference gradient image (core/measurements/ImagePyramid.h:59-99,
// core/image_proc/Gradient.h:16-75).  Pure streaming kernels (HBM-bound).
#include "../../include/mbavo.h"
#include "host_math.h"
#include "pixel_math.h"
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <vector>

namespace mbavo
{
    // dst(h,w) = uchar(0.25f * (a + b + c + d)), truncation toward zero (ImagePyramid.h:86-91);
    // the four uint8 sum exactly in float and 0.25 scaling is exact, so this is (a+b+c+d) >> 2.
    __global__ void k_pyr_down(const unsigned char *__restrict__ src, int W, int Hl, int Wl,
                               unsigned char *__restrict__ dst)
    {
        const int w = blockIdx.x * blockDim.x + threadIdx.x, h = blockIdx.y;
        if (w >= Wl || h >= Hl) return;
        const unsigned char *r0 = src + (size_t)(2 * h) * W + 2 * w;
        const unsigned char *r1 = r0 + W;
        const int s = (int)r0[0] + (int)r0[1] + (int)r1[0] + (int)r1[1];
        dst[(size_t)h * Wl + w] = (unsigned char)(s >> 2);
    }

    // interleaved [dx, dy] = 0.5 * (right - left), 0.5 * (bottom - top); zero on the 1-px border
    __global__ void k_gradients(const unsigned char *__restrict__ src, int H, int W, float2 *__restrict__ g)
    {
        const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
        if (x >= W || y >= H) return;
        const size_t i = (size_t)y * W + x;
        float2 v = make_float2(0.f, 0.f);
        if (!(x == 0 || y == 0 || x == W - 1 || y == H - 1))
        {
            v.x = 0.5f * ((float)src[i + 1] - (float)src[i - 1]);
            v.y = 0.5f * ((float)src[i + W] - (float)src[i - W]);
        }
        g[i] = v;
    }
    // the same differences stored as half pairs: every value is a multiple of 0.5 in [-127.5, 127.5] -> exact
    __global__ void k_gradients_half(const unsigned char *__restrict__ src, int H, int W, __half2 *__restrict__ g)
    {
        const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
        if (x >= W || y >= H) return;
        const size_t i = (size_t)y * W + x;
        float dx = 0.f, dy = 0.f;
        if (!(x == 0 || y == 0 || x == W - 1 || y == H - 1))
        {
            dx = 0.5f * ((float)src[i + 1] - (float)src[i - 1]);
            dy = 0.5f * ((float)src[i + W] - (float)src[i - W]);
        }
        g[i] = __floats2half2_rn(dx, dy);
    }
    // intensity and both doubled differences in one word per pixel (pixel_math.h: pack_keyframe_word)
    __global__ void k_pack_keyframe(const unsigned char *__restrict__ src, int H, int W, unsigned *__restrict__ out)
    {
        const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
        if (x >= W || y >= H) return;
        const size_t i = (size_t)y * W + x;
        int kx = 0, ky = 0;
        if (!(x == 0 || y == 0 || x == W - 1 || y == H - 1))
        {
            kx = (int)src[i + 1] - (int)src[i - 1];
            ky = (int)src[i + W] - (int)src[i - W];
        }
        out[i] = pack_keyframe_word((int)src[i], kx, ky);
    }
    // Synthetic motion blur (generate_synthetic_data.cpp:127-214): every output pixel is warped into the sharp
    // image through each of the n sampled poses, every warp is truncated to 8 bits as warp_image() stores it, the n
    // images are averaged in float and rounded to nearest even like cv::Mat::convertTo(CV_8U).
    // The warp follows compute_pixel_intensity.h:113-144 operation by operation, without FMA contraction, so the
    // generator is bit-identical to the CPU one (a 1e-13 difference in the tap coordinate can flip an 8-bit
    // truncation); this is input synthesis, not the hot path, so the extra flops do not matter.
    __global__ void k_synth_blur(const unsigned char *__restrict__ ref, int H, int W, double D, double fx, double fy,
                                 double cx, double cy, const double *__restrict__ poses /*n x 7*/, int n,
                                 unsigned char *__restrict__ out)
    {
#pragma clang fp contract(off)
        const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
        if (c >= W || r >= H) return;
        double xh = ((double)c - cx) / fx;
        double yh = ((double)r - cy) / fy;
        const double zh = 1. / (double)sqrtf((float)(1. + xh * xh + yh * yh));
        xh *= zh;
        yh *= zh;
        float acc = 0.f;
        for (int i = 0; i < n; ++i)
        {
            const double *p = poses + 7 * i;
            const double x = p[0], y = p[1], z = p[2], qx = p[3], qy = p[4], qz = p[5], qw = p[6];
            const double lambda = 2. * xh * (qx * qz - qw * qy) + 2. * yh * (qx * qw + qy * qz) +
                                  zh * (qw * qw - qx * qx - qy * qy + qz * qz);
            const double s = (D - z) / lambda;
            const double pc[3] = {s * xh, s * yh, s * zh};
            double pr[3];
            qrotate(Quat{qx, qy, qz, qw}, pc, pr);
            const double Px = pr[0] + x, Py = pr[1] + y, Pz = pr[2] + z;
            const double iz = 1. / (Pz + 1e-8);
            const double u = fx * (Px * iz) + cx, v = fy * (Py * iz) + cy;
            double val = 0.0, gx, gy;
            if (!bilinear_tap<false>(ref, nullptr, H, W, u, v, val, gx, gy)) val = 0.0;
            acc = acc + (float)(unsigned char)val;
        }
        const float m = acc / (float)n;
        int q = (int)rintf(m);
        q = q < 0 ? 0 : (q > 255 ? 255 : q);
        out[(size_t)r * W + c] = (unsigned char)q;
    }
} // namespace mbavo

extern "C" int mbavo_synthesize_blur(const unsigned char *d_ref, int H, int W, double plane_depth, const double intr[4],
                                     int k, double t0, double dt, const double *h_knots_t, const double *h_knots_R, int N,
                                     double cap, double exp_t, int num_samples, unsigned char *d_out, void *stream)
{
    if (!d_ref || !d_out || !intr || !h_knots_t || !h_knots_R || (k != 2 && k != 4) || num_samples < 2 || H < 2 || W < 2)
        return MBAVO_E_ARG;
    SLAM::Core::SplineSE3 spline(t0, dt);
    spline.setSplineDegK(k);
    for (int i = 0; i < N; ++i) spline.InsertControlKnot(h_knots_R + 4 * i, h_knots_t + 3 * i);
    std::vector<double> poses((size_t)num_samples * 7);
    for (int i = 0; i < num_samples; ++i)
    { // sample times of synthesize_motion_blurred_img (:201): cap - exp/2 + i*exp/(n-1)
        const double t = cap - exp_t * 0.5 + i * exp_t / (num_samples - 1);
        if (!spline.GetPose(t, &poses[7 * i + 3], &poses[7 * i])) return MBAVO_E_RANGE;
    }
    double *d_poses = nullptr;
    hipError_t e = hipMalloc((void **)&d_poses, poses.size() * sizeof(double));
    if (e != hipSuccess) return (int)e;
    e = hipMemcpyAsync(d_poses, poses.data(), poses.size() * sizeof(double), hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess)
    {
        hipLaunchKernelGGL(mbavo::k_synth_blur, dim3((W + 127) / 128, H), dim3(128), 0, (hipStream_t)stream, d_ref, H, W,
                           plane_depth, intr[0], intr[1], intr[2], intr[3], d_poses, num_samples, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(d_poses);
    return (int)e;
}

extern "C" int mbavo_image_gradients_u8_half(const unsigned char *d_src, int H, int W, void *d_dIxy_half, void *stream)
{
    if (!d_src || !d_dIxy_half || H < 1 || W < 1) return MBAVO_E_ARG;
    hipLaunchKernelGGL(mbavo::k_gradients_half, dim3((W + 255) / 256, H), dim3(256), 0, (hipStream_t)stream, d_src, H, W,
                       (__half2 *)d_dIxy_half);
    return (int)hipGetLastError();
}

extern "C" int mbavo_pack_keyframe_u8(const unsigned char *d_src, int H, int W, void *d_packed, void *stream)
{
    if (!d_src || !d_packed || H < 1 || W < 1) return MBAVO_E_ARG;
    hipLaunchKernelGGL(mbavo::k_pack_keyframe, dim3((W + 255) / 256, H), dim3(256), 0, (hipStream_t)stream, d_src, H, W, (unsigned *)d_packed);
    return (int)hipGetLastError();
}

extern "C" int mbavo_pyramid_down_u8(const unsigned char *d_src, int H, int W, unsigned char *d_dst, void *stream)
{
    if (!d_src || !d_dst || H < 2 || W < 2) return MBAVO_E_ARG;
    const int Hl = H / 2, Wl = W / 2;
    hipLaunchKernelGGL(mbavo::k_pyr_down, dim3((Wl + 255) / 256, Hl), dim3(256), 0, (hipStream_t)stream, d_src, W, Hl, Wl, d_dst);
    return (int)hipGetLastError();
}

extern "C" int mbavo_image_gradients_u8(const unsigned char *d_src, int H, int W, float *d_dIxy, void *stream)
{
    if (!d_src || !d_dIxy || H < 1 || W < 1) return MBAVO_E_ARG;
    hipLaunchKernelGGL(mbavo::k_gradients, dim3((W + 255) / 256, H), dim3(256), 0, (hipStream_t)stream, d_src, H, W, (float2 *)d_dIxy);
    return (int)hipGetLastError();
}
