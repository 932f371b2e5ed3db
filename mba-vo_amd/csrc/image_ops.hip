// image_ops.hip -- keyframe input producers on device: 2x2 box pyramid level and
// central-difference gradient image (core/measurements/ImagePyramid.h:59-99,
// core/image_proc/Gradient.h:16-75).  Pure streaming kernels (HBM-bound).
#include "../../include/mbavo.h"
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

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
} // namespace mbavo

extern "C" int mbavo_image_gradients_u8_half(const unsigned char *d_src, int H, int W, void *d_dIxy_half, void *stream)
{
    if (!d_src || !d_dIxy_half || H < 1 || W < 1) return MBAVO_E_ARG;
    hipLaunchKernelGGL(mbavo::k_gradients_half, dim3((W + 255) / 256, H), dim3(256), 0, (hipStream_t)stream, d_src, H, W,
                       (__half2 *)d_dIxy_half);
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
