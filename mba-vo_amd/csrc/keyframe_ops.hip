// keyframe_ops.hip -- keyframe pre-processing on device: gradient magnitude, semi-dense
// keypoint selection and the depth lookup of BlurAwareDirectTracker::tmpProcessKeyframe, so that a new keyframe
// needs one H2D of the image (+ depth map) and one small D2H of the keypoint count instead of host loops over
// every pixel of every level.
//
//   gradient magnitude   core/image_proc/Gradient.h:56-71       sqrt(dx^2 + dy^2) of the central differences
//   candidates           FeatureDetectorSemiDense.cpp:27-43      magnitude > score_threshold, row-major order
//   grid selection       FeatureDetectorBase.cpp:49-91           per cell the first pixel of strictly largest response
//   depth lookup         blur_aware_direct_tracker.cpp:389-415   z at the level-0 position, drop z < 1e-2
//
// Integer / index work: results (positions, order, count) are bit-identical to the CPU restatement.
// The magnitude is never materialised for the detector: it is recomputed from the u8 image (3 loads per pixel,
// L2-resident), which is cheaper than writing and re-reading a float image.  One wave per grid cell; the ordered
// compaction over <= a few thousand cells is a single-block scan.  HBM-bound streaming work, ~1 byte per pixel.
#include "../../include/mbavo.h"
#include "engine.h"
#include "vo_frontend.h"
#include <cmath>
#include <cstring>
#include <hip/hip_runtime.h>

namespace mbavo
{
    __device__ __forceinline__ float gradient_magnitude(const unsigned char *__restrict__ src, int H, int W, int x, int y)
    {
        if (x == 0 || y == 0 || x == W - 1 || y == H - 1) return 0.f;
        const size_t i = (size_t)y * W + x;
        const float dx = 0.5f * ((float)src[i + 1] - (float)src[i - 1]);
        const float dy = 0.5f * ((float)src[i + W] - (float)src[i - W]);
        // dx, dy are multiples of 0.5 in [-127.5, 127.5]: the sum of squares is exact in fp32 whatever the
        // contraction; the reference takes the double sqrt of that float and rounds to float, which equals the
        // correctly rounded float sqrt (53 >= 2*24 + 2 bits).  sqrtf is the IEEE one here (hipcc's default
        // -fhip-fp32-correctly-rounded-divide-sqrt); __fsqrt_rn maps to the 1-ulp native instruction.
        return sqrtf(dx * dx + dy * dy);
    }

    __global__ void k_grad_mag(const unsigned char *__restrict__ src, int H, int W, float *__restrict__ mag)
    {
        const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
        if (x >= W || y >= H) return;
        mag[(size_t)y * W + x] = gradient_magnitude(src, H, W, x, y);
    }

    // level-0 depth of a level-`lv` pixel (blur_aware_direct_tracker.cpp:398-400): int(x * 2^lv + 0.5)
    __device__ __forceinline__ bool depth_of(const float *__restrict__ depth, int W0, double scale, int x, int y, float &z)
    {
        const int x0 = (int)((float)x * scale + 0.5), y0 = (int)((float)y * scale + 0.5);
        z = depth[(size_t)y0 * W0 + x0];
        return !((double)z < 1e-2);
    }

    // one wave per grid cell
    __device__ __forceinline__ void detect_cell(const unsigned char *__restrict__ src, int H, int W, int cell_h, int cell_w, int cells_w,
                                                float thr, const float *__restrict__ depth, int W0, double scale, int ci, int lane,
                                                CellPick *__restrict__ picks)
    {
        const int y0 = (ci / cells_w) * cell_h, x0 = (ci % cells_w) * cell_w;
        float best = 0.f; // cv::KeyPoint() has response 0: a pixel must beat it strictly
        int best_idx = 0x7fffffff;
        const int n = cell_h * cell_w;
        for (int i = lane; i < n; i += 64)
        {
            const int y = y0 + i / cell_w, x = x0 + i % cell_w;
            if (y >= H || x >= W) continue;
            const float m = gradient_magnitude(src, H, W, x, y);
            if (m > thr && best < m) { best = m; best_idx = y * W + x; } // per lane the scan order is increasing
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
        {
            const float om = __shfl_xor(best, off);
            const int oi = __shfl_xor(best_idx, off);
            if (om > best || (om == best && oi < best_idx)) { best = om; best_idx = oi; }
        }
        if (lane == 0)
        {
            CellPick p;
            p.keep = 0; p.x = 0; p.y = 0; p.z = 0.f;
            if (!(best < 1e-6)) // FeatureDetectorBase.cpp:82-85
            {
                p.y = best_idx / W; p.x = best_idx - p.y * W;
                p.keep = depth == nullptr ? 1 : (depth_of(depth, W0, scale, p.x, p.y, p.z) ? 1 : 0); // (null: the caller tests the depth)
            }
            picks[ci] = p;
        }
    }
    __global__ __launch_bounds__(64) void k_detect_cells(const unsigned char *__restrict__ src, int H, int W, int cell_h,
                                                         int cell_w, int cells_w, float thr,
                                                         const float *__restrict__ depth, int W0, double scale,
                                                         CellPick *__restrict__ picks)
    {
        detect_cell(src, H, W, cell_h, cell_w, cells_w, thr, depth, W0, scale, (int)blockIdx.x, (int)threadIdx.x, picks);
    }

    // ---- a keyframe's levels in ONE launch each (round 3): the per-level kernels are latency-bound (4-10 us each whatever the
    // level's size), and a keyframe ran 3 of them per level back to back -- 11 launches at four levels, ~100 us of the 150 us a
    // keyframe cost.  The levels' parameters travel by value; a workgroup finds its level from the prefix sums.
    struct PyramidLevels
    {
        const unsigned char *img[8];
        float2 *grad[8];
        int H[8], W[8];
        int row0[9];                         // gradients: first grid row of every level
        int ch[8], cw[8], cells_w[8], cell0[9]; // grid selection: cell size, cells per row, first cell of every level
        double scale[8];
        int n;
    };
    // 2 x 2 box with truncation (ImagePyramid.h:59-99), up to three levels below `src` in one launch: a workgroup takes a
    // 32 x 32 tile of the source down to 16 x 16, 8 x 8 and 4 x 4 through LDS -- the same integer operations per level
    __global__ __launch_bounds__(256) void k_pyr_down_multi(const unsigned char *__restrict__ src, int Hs, int Ws, unsigned char *__restrict__ d1,
                                                            unsigned char *__restrict__ d2, unsigned char *__restrict__ d3, int n)
    {
        __shared__ int t1[16][17], t2[8][9];
        const int tid = threadIdx.x;
        const int H1 = Hs / 2, W1 = Ws / 2, H2 = H1 / 2, W2 = W1 / 2, H3 = H2 / 2, W3 = W2 / 2;
        {
            const int ty = tid >> 4, tx = tid & 15, h = blockIdx.y * 16 + ty, w = blockIdx.x * 16 + tx;
            int v = 0;
            if (h < H1 && w < W1)
            {
                const unsigned char *r0 = src + (size_t)(2 * h) * Ws + 2 * w, *r1 = r0 + Ws;
                v = ((int)r0[0] + (int)r0[1] + (int)r1[0] + (int)r1[1]) >> 2;
                d1[(size_t)h * W1 + w] = (unsigned char)v;
            }
            t1[ty][tx] = v;
        }
        if (n < 2) return;
        __syncthreads();
        if (tid < 64)
        {
            const int ty = tid >> 3, tx = tid & 7, h = blockIdx.y * 8 + ty, w = blockIdx.x * 8 + tx;
            const int v = (t1[2 * ty][2 * tx] + t1[2 * ty][2 * tx + 1] + t1[2 * ty + 1][2 * tx] + t1[2 * ty + 1][2 * tx + 1]) >> 2;
            if (h < H2 && w < W2) d2[(size_t)h * W2 + w] = (unsigned char)v; // (its four sources are inside level 1 whenever it is inside level 2)
            t2[ty][tx] = v;
        }
        if (n < 3) return;
        __syncthreads();
        if (tid < 16)
        {
            const int ty = tid >> 2, tx = tid & 3, h = blockIdx.y * 4 + ty, w = blockIdx.x * 4 + tx;
            const int v = (t2[2 * ty][2 * tx] + t2[2 * ty][2 * tx + 1] + t2[2 * ty + 1][2 * tx] + t2[2 * ty + 1][2 * tx + 1]) >> 2;
            if (h < H3 && w < W3) d3[(size_t)h * W3 + w] = (unsigned char)v;
        }
    }
    // interleaved [dx, dy] central differences of every level (image_ops.hip: k_gradients)
    __global__ __launch_bounds__(256) void k_gradients_multi(const PyramidLevels lv)
    {
        int l = 0;
        while (l + 1 < lv.n && (int)blockIdx.y >= lv.row0[l + 1]) ++l;
        const int H = lv.H[l], W = lv.W[l], y = (int)blockIdx.y - lv.row0[l], x = blockIdx.x * blockDim.x + threadIdx.x;
        if (x >= W || y >= H) return;
        const unsigned char *src = lv.img[l];
        const size_t i = (size_t)y * W + x;
        float2 v = make_float2(0.f, 0.f);
        if (!(x == 0 || y == 0 || x == W - 1 || y == H - 1))
        {
            v.x = 0.5f * ((float)src[i + 1] - (float)src[i - 1]);
            v.y = 0.5f * ((float)src[i + W] - (float)src[i - W]);
        }
        lv.grad[l][i] = v;
    }
    // grid selection of every level: one wave per cell, the cells of all levels in one grid (picks in level order)
    __global__ __launch_bounds__(64) void k_detect_cells_multi(const PyramidLevels lv, float thr, int W0, CellPick *__restrict__ picks)
    {
        int l = 0;
        while (l + 1 < lv.n && (int)blockIdx.x >= lv.cell0[l + 1]) ++l;
        detect_cell(lv.img[l], lv.H[l], lv.W[l], lv.ch[l], lv.cw[l], lv.cells_w[l], thr, nullptr, W0, lv.scale[l],
                    (int)blockIdx.x - lv.cell0[l], (int)threadIdx.x, picks + lv.cell0[l]);
    }

    // ordered compaction of the kept cells: single block, chunked exclusive scan
    __global__ __launch_bounds__(256) void k_compact_cells(const CellPick *__restrict__ picks, int n, double *__restrict__ kp_xy,
                                                           double *__restrict__ kp_z, int cap, int *__restrict__ count)
    {
        __shared__ int sm[256];
        __shared__ int base;
        if (threadIdx.x == 0) base = 0;
        __syncthreads();
        for (int c0 = 0; c0 < n; c0 += 256)
        {
            const int i = c0 + threadIdx.x;
            CellPick p;
            p.keep = 0;
            if (i < n) p = picks[i];
            sm[threadIdx.x] = p.keep;
            __syncthreads();
            for (int d = 1; d < 256; d <<= 1)
            {
                const int v = threadIdx.x >= d ? sm[threadIdx.x - d] : 0;
                __syncthreads();
                sm[threadIdx.x] += v;
                __syncthreads();
            }
            const int pos = base + sm[threadIdx.x] - p.keep;
            if (p.keep && pos < cap)
            {
                kp_xy[2 * pos] = (double)p.x; kp_xy[2 * pos + 1] = (double)p.y;
                kp_z[pos] = (double)p.z;
            }
            __syncthreads();
            if (threadIdx.x == 255) base += sm[255];
            __syncthreads();
        }
        if (threadIdx.x == 0) *count = base;
    }

    // ---- no grid selection: every candidate, in row-major order.  Rows are the segments of the ordered compaction.
    __device__ __forceinline__ bool row_candidate(const unsigned char *__restrict__ src, int H, int W, float thr,
                                                  const float *__restrict__ depth, int W0, double scale, int x, int y, float &z)
    {
        if (x >= W) return false;
        const float m = gradient_magnitude(src, H, W, x, y);
        if (!(m > thr)) return false;
        return depth_of(depth, W0, scale, x, y, z);
    }

    __global__ __launch_bounds__(64) void k_rows_count(const unsigned char *__restrict__ src, int H, int W, float thr,
                                                       const float *__restrict__ depth, int W0, double scale,
                                                       int *__restrict__ row_count)
    {
        const int y = blockIdx.x, lane = threadIdx.x;
        int n = 0;
        for (int x0 = 0; x0 < W; x0 += 64)
        {
            float z;
            n += __popcll(__ballot(row_candidate(src, H, W, thr, depth, W0, scale, x0 + lane, y, z)));
        }
        if (lane == 0) row_count[y] = n;
    }

    __global__ __launch_bounds__(256) void k_rows_scan(int *__restrict__ row_count, int H, int *__restrict__ count)
    { // in-place exclusive scan over the rows
        __shared__ int sm[256];
        __shared__ int base;
        if (threadIdx.x == 0) base = 0;
        __syncthreads();
        for (int c0 = 0; c0 < H; c0 += 256)
        {
            const int i = c0 + threadIdx.x;
            const int v0 = i < H ? row_count[i] : 0;
            sm[threadIdx.x] = v0;
            __syncthreads();
            for (int d = 1; d < 256; d <<= 1)
            {
                const int v = threadIdx.x >= d ? sm[threadIdx.x - d] : 0;
                __syncthreads();
                sm[threadIdx.x] += v;
                __syncthreads();
            }
            if (i < H) row_count[i] = base + sm[threadIdx.x] - v0;
            __syncthreads();
            if (threadIdx.x == 255) base += sm[255];
            __syncthreads();
        }
        if (threadIdx.x == 0) *count = base;
    }

    __global__ __launch_bounds__(64) void k_rows_write(const unsigned char *__restrict__ src, int H, int W, float thr,
                                                       const float *__restrict__ depth, int W0, double scale,
                                                       const int *__restrict__ row_off, double *__restrict__ kp_xy,
                                                       double *__restrict__ kp_z, int cap)
    {
        const int y = blockIdx.x, lane = threadIdx.x;
        int pos = row_off[y];
        for (int x0 = 0; x0 < W; x0 += 64)
        {
            float z = 0.f;
            const bool c = row_candidate(src, H, W, thr, depth, W0, scale, x0 + lane, y, z);
            const unsigned long long b = __ballot(c);
            const int mine = pos + __popcll(b & ((1ull << lane) - 1ull));
            if (c && mine < cap)
            {
                kp_xy[2 * mine] = (double)(x0 + lane); kp_xy[2 * mine + 1] = (double)y;
                kp_z[mine] = (double)z;
            }
            pos += __popcll(b);
        }
    }

    int detect_semidense(Engine &eng, const unsigned char *d_img, int H, int W, int level, int im_H0, int im_W0, int cell_H,
                         int cell_W, float thr, const float *d_depth_z, double *d_kp_xy, double *d_kp_z, int cap, int *h_count)
    {
        if (!d_img || !d_depth_z || !d_kp_xy || !d_kp_z || !h_count || H < 1 || W < 1 || level < 0 || level > 30 || cap < 0)
            return MBAVO_E_ARG;
        hipStream_t st = eng.stream();
        const double scale = std::pow(2, level);
        int *d_count = (int *)eng.named_scratch(8, sizeof(int));
        if (!d_count) return MBAVO_E_ARG;
        hipError_t e;
        if (cell_H > 0 && cell_W > 0)
        { // FeatureDetectorBase.cpp:56-64
            const int sf = (int)std::pow(2, level);
            const int Hl = im_H0 / sf, Wl = im_W0 / sf;
            const int ch = (int)(cell_H / std::pow(1.414, level)), cw = (int)(cell_W / std::pow(1.414, level));
            if (ch < 1 || cw < 1) return MBAVO_E_ARG; // the reference divides by zero here
            const int cells_h = Hl / ch + 1, cells_w = Wl / cw + 1;
            if ((H - 1) / ch >= cells_h || (W - 1) / cw >= cells_w) return MBAVO_E_RANGE; // std::vector::at would throw
            const int nc = cells_h * cells_w;
            CellPick *picks = (CellPick *)eng.named_scratch(9, sizeof(CellPick) * nc);
            if (!picks) return MBAVO_E_ARG;
            hipLaunchKernelGGL(k_detect_cells, dim3(nc), dim3(64), 0, st, d_img, H, W, ch, cw, cells_w, thr, d_depth_z, im_W0, scale, picks);
            hipLaunchKernelGGL(k_compact_cells, dim3(1), dim3(256), 0, st, picks, nc, d_kp_xy, d_kp_z, cap, d_count);
        }
        else
        {
            int *rows = (int *)eng.named_scratch(9, sizeof(int) * H);
            if (!rows) return MBAVO_E_ARG;
            hipLaunchKernelGGL(k_rows_count, dim3(H), dim3(64), 0, st, d_img, H, W, thr, d_depth_z, im_W0, scale, rows);
            hipLaunchKernelGGL(k_rows_scan, dim3(1), dim3(256), 0, st, rows, H, d_count);
            hipLaunchKernelGGL(k_rows_write, dim3(H), dim3(64), 0, st, d_img, H, W, thr, d_depth_z, im_W0, scale, rows, d_kp_xy, d_kp_z, cap);
        }
        if ((e = hipGetLastError()) != hipSuccess) return (int)e;
        if ((e = hipMemcpyAsync(h_count, d_count, sizeof(int), hipMemcpyDeviceToHost, st)) != hipSuccess) return (int)e;
        return (int)hipStreamSynchronize(st);
    }

    int detect_cells_enqueue(Engine &eng, const unsigned char *d_img, int H, int W, int level, int im_H0, int im_W0, int cell_H,
                             int cell_W, float thr, CellPick *d_picks, int *num_cells)
    {
        if (!d_img || !d_picks || !num_cells || H < 1 || W < 1 || level < 0 || level > 30 || cell_H < 1 || cell_W < 1) return MBAVO_E_ARG;
        // FeatureDetectorBase.cpp:56-64 (as in detect_semidense)
        const int sf = (int)std::pow(2, level);
        const int Hl = im_H0 / sf, Wl = im_W0 / sf;
        const int ch = (int)(cell_H / std::pow(1.414, level)), cw = (int)(cell_W / std::pow(1.414, level));
        if (ch < 1 || cw < 1) return MBAVO_E_ARG;
        const int cells_h = Hl / ch + 1, cells_w = Wl / cw + 1;
        if ((H - 1) / ch >= cells_h || (W - 1) / cw >= cells_w) return MBAVO_E_RANGE;
        const int nc = cells_h * cells_w;
        hipLaunchKernelGGL(k_detect_cells, dim3(nc), dim3(64), 0, eng.stream(), d_img, H, W, ch, cw, cells_w, thr, (const float *)nullptr,
                           im_W0, std::pow(2, level), d_picks);
        *num_cells = nc;
        return (int)hipGetLastError();
    }
    int pyramid_enqueue(Engine &eng, unsigned char *const *d_levels, int H0, int W0, int L, hipStream_t on)
    { // levels 1 .. L-1 from level 0, three per launch (`on`: another stream than the engine's, or null)
        if (!d_levels || L < 1 || L > 8 || H0 < 1 || W0 < 1) return MBAVO_E_ARG;
        hipStream_t st_ = on ? on : eng.stream();
        for (int l = 0; l + 1 < L; l += 3)
        {
            const int n = L - 1 - l < 3 ? L - 1 - l : 3, Hs = H0 >> l, Ws = W0 >> l;
            if (Hs < 2 || Ws < 2) return MBAVO_E_ARG;
            hipLaunchKernelGGL(k_pyr_down_multi, dim3((Ws / 2 + 15) / 16, (Hs / 2 + 15) / 16), dim3(256), 0, st_, d_levels[l], Hs, Ws,
                               d_levels[l + 1], n >= 2 ? d_levels[l + 2] : nullptr, n >= 3 ? d_levels[l + 3] : nullptr, n);
        }
        return (int)hipGetLastError();
    }

    int keyframe_levels_enqueue(Engine &eng, unsigned char *const *d_levels, float *const *d_grads, int H0, int W0, int L, int cell_H, int cell_W,
                                float thr, CellPick *d_picks, int *cells_per_level, hipStream_t on)
    {
        if (!d_levels || !d_grads || L < 1 || L > 8) return MBAVO_E_ARG;
        hipStream_t st_ = on ? on : eng.stream();
        int rc = pyramid_enqueue(eng, d_levels, H0, W0, L, on);
        if (rc != 0) return rc;
        PyramidLevels lv;
        memset(&lv, 0, sizeof(lv));
        lv.n = L;
        const bool grid = d_picks != nullptr;
        for (int l = 0; l < L; ++l)
        {
            lv.img[l] = d_levels[l]; lv.grad[l] = (float2 *)d_grads[l];
            lv.H[l] = H0 >> l; lv.W[l] = W0 >> l;
            lv.row0[l + 1] = lv.row0[l] + lv.H[l];
            lv.scale[l] = std::pow(2, l);
            if (grid)
            { // FeatureDetectorBase.cpp:56-64 (as in detect_semidense)
                const int sf = (int)std::pow(2, l);
                const int Hl = H0 / sf, Wl = W0 / sf;
                const int ch = (int)(cell_H / std::pow(1.414, l)), cw = (int)(cell_W / std::pow(1.414, l));
                if (ch < 1 || cw < 1) return MBAVO_E_ARG;
                const int cells_h = Hl / ch + 1, cells_w = Wl / cw + 1;
                if ((lv.H[l] - 1) / ch >= cells_h || (lv.W[l] - 1) / cw >= cells_w) return MBAVO_E_RANGE;
                lv.ch[l] = ch; lv.cw[l] = cw; lv.cells_w[l] = cells_w;
                lv.cell0[l + 1] = lv.cell0[l] + cells_h * cells_w;
                if (cells_per_level) cells_per_level[l] = cells_h * cells_w;
            }
        }
        hipLaunchKernelGGL(k_gradients_multi, dim3((W0 + 255) / 256, lv.row0[L]), dim3(256), 0, st_, lv);
        if (grid) hipLaunchKernelGGL(k_detect_cells_multi, dim3(lv.cell0[L]), dim3(64), 0, st_, lv, thr, W0, d_picks);
        return (int)hipGetLastError();
    }
} // namespace mbavo

extern "C" int mbavo_gradient_magnitude_u8(const unsigned char *d_src, int H, int W, float *d_mag, void *stream)
{
    if (!d_src || !d_mag || H < 1 || W < 1) return MBAVO_E_ARG;
    hipLaunchKernelGGL(mbavo::k_grad_mag, dim3((W + 255) / 256, H), dim3(256), 0, (hipStream_t)stream, d_src, H, W, d_mag);
    return (int)hipGetLastError();
}
