// ba_tracker.hip -- the reference's launcher-by-launcher API on gfx950, plus
// CudaSharedStorages management and evaluate_cost_hessian_gradient on the fused engine.
// Reference counterparts are cited at each function.
#include "ba_tracker.h"
#include "engine.h"
#include "host_math.h"
#include "pixel_math.h"
#include "se3_math.h"

#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <mutex>

namespace SLAM
{
    namespace VO
    {
        using namespace mbavo;

#define HIP_OR_DIE(expr)                                                                       \
    do                                                                                         \
    {                                                                                          \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
        {                                                                                      \
            fprintf(stderr, "ba_tracker: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), \
                    __FILE__, __LINE__);                                                       \
            abort();                                                                           \
        }                                                                                      \
    } while (0)

        static void sync_and_check(const char *what)
        {
            hipError_t e = hipGetLastError();
            if (e == hipSuccess) e = hipDeviceSynchronize();
            if (e != hipSuccess)
            {
                fprintf(stderr, "ba_tracker: %s failed: %s\n", what, hipGetErrorString(e));
                abort();
            }
        }

        // ---------------------------------------------------------------- stage 1
        // compute_virtual_camera_poses.cu:9-110: one lane per (frame, sample)
        template <int KD>
        __global__ void k_api_poses(int S, int F, const double *__restrict__ cap, const double *__restrict__ exp_t,
                                    double t0, double dt, const double *__restrict__ kt, const double *__restrict__ kR,
                                    double *__restrict__ poses, double *__restrict__ J_t, double *__restrict__ J_R)
        {
            const int v = blockIdx.x * blockDim.x + threadIdx.x;
            if (v >= S * F) return;
            const int f = v / S, s = v - f * S;
            const double t = cap[f] - exp_t[f] * 0.5 + s * exp_t[f] / (S - 1 + 1e-8);
            int idx;
            double u;
            spline_segment(t, t0, dt, idx, u);
            double c[KD], p[3], JR[12 * KD];
            trans_coeffs<KD>(u, c);
            spline_translation<KD>(kt + 3 * idx, c, p);
            Quat q;
            if (J_R != nullptr) q = spline_rotation<KD, true>(kR + 4 * idx, u, JR);
            else q = spline_rotation<KD, false>(kR + 4 * idx, u, nullptr);
            double *o = poses + (size_t)v * 7;
            o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
            o[3] = q.x; o[4] = q.y; o[5] = q.z; o[6] = q.w;
            if (J_t != nullptr)
            { // dense 3 x 3k = kron(c, I3)  (SplineFunctor.h:30-41,74-91)
                double *jt = J_t + (size_t)v * 9 * KD;
                for (int i = 0; i < 9 * KD; ++i) jt[i] = 0.0;
                for (int a = 0; a < 3; ++a)
                    for (int j = 0; j < KD; ++j) jt[a * 3 * KD + 3 * j + a] = c[j];
            }
            if (J_R != nullptr)
            {
                double *jr = J_R + (size_t)v * 12 * KD;
                for (int i = 0; i < 12 * KD; ++i) jr[i] = JR[i];
            }
        }

        void compute_virtual_camera_poses(const int S, const int F, const double *img_cap_time,
                                          const double *img_exp_time, const int spline_deg_k,
                                          const double spline_start_time, const double spline_sample_interval,
                                          const double *knots_t, const double *knots_R, double *sampled_virtual_poses,
                                          double *J_t, double *J_R, double *, double *, double *, double *)
        {
            const int n = S * F;
            if (n <= 0) return;
            if (spline_deg_k == 2)
                hipLaunchKernelGGL(k_api_poses<2>, dim3((n + 63) / 64), dim3(64), 0, 0, S, F, img_cap_time, img_exp_time,
                                   spline_start_time, spline_sample_interval, knots_t, knots_R, sampled_virtual_poses, J_t, J_R);
            else if (spline_deg_k == 4)
                hipLaunchKernelGGL(k_api_poses<4>, dim3((n + 63) / 64), dim3(64), 0, 0, S, F, img_cap_time, img_exp_time,
                                   spline_start_time, spline_sample_interval, knots_t, knots_R, sampled_virtual_poses, J_t, J_R);
            else
            {
                fprintf(stderr, "ba_tracker: unsupported spline degree %d (2 or 4)\n", spline_deg_k);
                abort();
            }
            sync_and_check("compute_virtual_camera_poses");
        }

        // ---------------------------------------------------------------- stage 2
        // compute_local_patches_xy.cu:9-50
        __global__ void k_api_patches(int S, int F, const double *__restrict__ poses,
                                      const Core::Vector2d *__restrict__ kps, const double *__restrict__ kz, int K,
                                      Camera cam, Core::Vector2d *__restrict__ out)
        {
            const int g = blockIdx.x * blockDim.x + threadIdx.x;
            if (g >= F * K) return;
            const int f = g / K, i = g - f * K;
            const double *pose = poses + (size_t)(f * S + S / 2) * 7;
            double x, y;
            patch_centre(pose, pose + 3, kps[i].values[0], kps[i].values[1], kz[i], cam, x, y);
            out[g].nDim = 2;
            out[g].values[0] = x;
            out[g].values[1] = y;
        }

        static Camera make_camera(const Core::VectorX<double, 4> &intr, const Core::VectorX<int, 2> &hw)
        {
            Camera c;
            c.fx = intr.values[0]; c.fy = intr.values[1]; c.cx = intr.values[2]; c.cy = intr.values[3];
            c.H = hw.values[0]; c.W = hw.values[1];
            return c;
        }

        void compute_local_patches_xy(const int S, const int F, const double *virtual_cam_poses,
                                      const Core::Vector2d *sparse_keypoints, const double *sparse_keypoints_z,
                                      const int K, const Core::VectorX<double, 4> &intrinsics,
                                      const Core::VectorX<int, 2> &im_HW, Core::Vector2d *local_patches_xy)
        {
            const int n = F * K;
            if (n <= 0) return;
            hipLaunchKernelGGL(k_api_patches, dim3((n + 255) / 256), dim3(256), 0, 0, S, F, virtual_cam_poses,
                               sparse_keypoints, sparse_keypoints_z, K, make_camera(intrinsics, im_HW), local_patches_xy);
            sync_and_check("compute_local_patches_xy");
        }

        // ---------------------------------------------------------------- stage 3
        // compute_hessian_gradients_cost.cu:23-156, one lane per pixel, S samples in registers,
        // dense 1x3 * J_t and 1x4 * J_R products against the caller's Jacobian buffers.
        template <int KD, bool WITH_J>
        __global__ void k_api_pixel(const unsigned char *__restrict__ I_ref, const float *__restrict__ G_ref,
                                    const unsigned char *const *__restrict__ I_cur, int S, int F,
                                    const double *__restrict__ poses, const double *__restrict__ J_t,
                                    const double *__restrict__ J_R, const Core::Vector2d *__restrict__ centres,
                                    const double *__restrict__ kz, int K, const int *__restrict__ pattern, int P,
                                    Camera cam, double *__restrict__ residuals, double *__restrict__ jac)
        {
            const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
            if (g >= (long long)F * K * P) return;
            const int f = (int)(g / ((long long)K * P));
            const int rem = (int)(g - (long long)f * K * P);
            const int kp = rem / P, pp = rem - kp * P;
            double Jrow[WITH_J ? 6 * KD : 1];
            if (WITH_J)
                for (int i = 0; i < 6 * KD; ++i) Jrow[i] = 0.0;
            double res = 0.0;
            const Core::Vector2d c = centres[(size_t)f * K + kp];
            const int px = (int)(c.values[0] + pattern[2 * pp]);
            const int py = (int)(c.values[1] + pattern[2 * pp + 1]);
            bool ok = !(px < 0 || px > cam.W - 1 || py < 0 || py > cam.H - 1);
            if (ok)
            {
                double ray[3];
                unit_ray(cam, (double)px, (double)py, ray);
                const double D = kz[kp];
                const double iz = 1. / (D + 1e-8);
                double isum = 0.0;
                for (int s = 0; s < S && ok; ++s)
                {
                    const int v = f * S + s;
                    const double *pose = poses + (size_t)v * 7;
                    double R[9], val, jt[3], b[4];
                    rotation_entries(pose + 3, R);
                    ok = sample_eval<WITH_J>(pose, pose + 3, R, ray, D, iz, cam, I_ref, G_ref, val, jt, b);
                    if (!ok) break;
                    isum += val;
                    if (WITH_J)
                    {
                        const double *A = J_t + (size_t)v * 9 * KD;
                        const double *Bm = J_R + (size_t)v * 12 * KD;
                        for (int col = 0; col < 3 * KD; ++col)
                        {
                            double a = jt[0] * A[col];
                            a += jt[1] * A[3 * KD + col];
                            a += jt[2] * A[6 * KD + col];
                            Jrow[col] += a;
                            double r = b[0] * Bm[col];
                            r += b[1] * Bm[3 * KD + col];
                            r += b[2] * Bm[6 * KD + col];
                            r += b[3] * Bm[9 * KD + col];
                            Jrow[3 * KD + col] += r;
                        }
                    }
                }
                if (ok)
                {
                    const double fS = (double)(float)S;
                    res = isum / fS - (double)I_cur[f][py * cam.W + px];
                    if (WITH_J)
                        for (int i = 0; i < 6 * KD; ++i) Jrow[i] = Jrow[i] / fS;
                }
            }
            residuals[g] = ok ? res : 0.0;
            if (WITH_J)
                for (int i = 0; i < 6 * KD; ++i) jac[g * 6 * KD + i] = ok ? Jrow[i] : 0.0;
        }

        void compute_pixel_jacobian_residual(const unsigned char *I_ref, const float *dIxy_ref,
                                             unsigned char const *const *I_cur_imgs, const int S, const int F,
                                             const double *poses, const int spline_deg_k, const double *J_t,
                                             const double *J_R, const Core::Vector2d *local_patches_XY,
                                             const double *keypoints_z, const int K, const int *pattern, const int P,
                                             const Core::VectorX<double, 4> &intrinsics,
                                             const Core::VectorX<int, 2> &im_size_HW, FLOAT *,
                                             double *pixel_residuals, double *pixel_jacobians_tR)
        {
            const long long n = (long long)F * K * P;
            if (n <= 0) return;
            const Camera cam = make_camera(intrinsics, im_size_HW);
            const dim3 grid((unsigned)((n + 127) / 128)), block(128);
            const bool wj = pixel_jacobians_tR != nullptr;
#define MBAVO_PIX(KD, WJ)                                                                                           \
    hipLaunchKernelGGL((k_api_pixel<KD, WJ>), grid, block, 0, 0, I_ref, dIxy_ref, I_cur_imgs, S, F, poses, J_t, J_R, \
                       local_patches_XY, keypoints_z, K, pattern, P, cam, pixel_residuals, pixel_jacobians_tR)
            if (spline_deg_k == 4) { if (wj) MBAVO_PIX(4, true); else MBAVO_PIX(4, false); }
            else if (spline_deg_k == 2) { if (wj) MBAVO_PIX(2, true); else MBAVO_PIX(2, false); }
            else { fprintf(stderr, "ba_tracker: unsupported spline degree %d\n", spline_deg_k); abort(); }
#undef MBAVO_PIX
            sync_and_check("compute_pixel_jacobian_residual");
        }

        // ---------------------------------------------------------------- stage 4
        // products of two entries of the patch's weighted rows, indexed by pixel (rounded products, summed in the
        // reference's reduce() order by patch_rho_sum)
        struct RowProducts
        {
            const double *rows;
            int stride, i, j;
            __device__ double operator[](int p) const
            {
#pragma clang fp contract(off)
                return rows[p * stride + i] * rows[p * stride + j];
            }
        };

        // compute_hessian_gradients_cost.cu:165-239: one wave per patch, the weighted rows
        // of its P pixels in LDS, lane e owns packed entries e, e+64, ...
        __global__ void k_api_patch(int P, int k, const double *__restrict__ residuals, const double *__restrict__ jac,
                                    double huber_a, double inv, double *__restrict__ blocks)
        {
            extern __shared__ __attribute__((aligned(16))) double sm[];
            const int ndim = 6 * k + 1, E = ndim * (ndim + 1) / 2;
            const int stride = ndim + 1; // +1 pad: lanes walk different rows without bank conflicts
            double *rows = sm;                 // [P][stride]
            double *rho = sm + (size_t)P * stride; // [P]
            const size_t patch = blockIdx.x;
            for (int p = threadIdx.x; p < P; p += blockDim.x)
            {
                const double r = residuals[patch * P + p];
                double w, rh;
                huber_weight(r, huber_a, w, rh);
                rho[p] = rh;
                rows[p * stride] = w * r;
                if (jac)
                    for (int i = 0; i < 6 * k; ++i) rows[p * stride + 1 + i] = w * jac[(patch * P + p) * 6 * k + i];
            }
            __syncthreads();
            double *out = blocks + patch * E;
            if (jac)
            {
                for (int e = 1 + threadIdx.x; e < E; e += blockDim.x)
                {
                    int i, j;
                    tri_decode(e, ndim, i, j);
                    out[e] = patch_rho_sum(RowProducts{rows, stride, i, j}, P) * inv; // reduction.h order
                }
            }
            if (threadIdx.x == 0)
            { // slot 0 := sum(rho) * inv, overwriting sum((w r)^2) (:232-238)
                out[0] = patch_rho_sum((const double *)rho, P) * inv;
            }
        }

        void compute_patch_cost_gradient_hessian(const int F, const int K, const int P, const int spline_deg_k,
                                                 const double *pixel_residuals, const double *pixel_jacobians,
                                                 const double huber_a, const double inv_num_residuals,
                                                 double *patch_cost_gradient_hessian)
        {
            const long long n = (long long)F * K;
            if (n <= 0 || P <= 0) return;
            const int ndim = 6 * spline_deg_k + 1;
            const size_t lds = ((size_t)P * (ndim + 1) + P) * sizeof(double);
            hipLaunchKernelGGL(k_api_patch, dim3((unsigned)n), dim3(64), lds, 0, P, spline_deg_k, pixel_residuals,
                               pixel_jacobians, huber_a, inv_num_residuals, patch_cost_gradient_hessian);
            sync_and_check("compute_patch_cost_gradient_hessian");
        }

        // ---------------------------------------------------------------- stage 5
        // compute_hessian_gradients_cost.cu:247-283: 256 strided lanes then a fixed tree
        __global__ void k_api_frame(int K, int E, const double *__restrict__ blocks,
                                    const unsigned char *__restrict__ flags, double *__restrict__ out)
        {
            __shared__ double sm[256];
            const int f = blockIdx.x, e = blockIdx.y;
            double s = 0.0;
            for (int i = threadIdx.x; i < K; i += 256)
            {
                if (flags != nullptr && flags[i] == 1) continue;
                s += blocks[((size_t)f * K + i) * E + e];
            }
            sm[threadIdx.x] = s;
            __syncthreads();
            for (int h = 128; h >= 1; h >>= 1)
            {
                if ((int)threadIdx.x < h) sm[threadIdx.x] += sm[threadIdx.x + h];
                __syncthreads();
            }
            if (threadIdx.x == 0) out[(size_t)f * E + e] = sm[0];
        }

        void compute_frame_cost_gradient_hessian(const int F, const int K, const int spline_deg_k,
                                                 const double *patch_cost_gradient_hessian,
                                                 const bool eval_gradient_hessian,
                                                 const unsigned char *keypoints_outlier_flags,
                                                 double *frame_cost_gradient_hessian)
        {
            if (F <= 0) return;
            const int ndim = 6 * spline_deg_k + 1, E = ndim * (ndim + 1) / 2;
            hipLaunchKernelGGL(k_api_frame, dim3(F, eval_gradient_hessian ? E : 1), dim3(256), 0, 0, K, E,
                               patch_cost_gradient_hessian, keypoints_outlier_flags, frame_cost_gradient_hessian);
            sync_and_check("compute_frame_cost_gradient_hessian");
        }

        // ---------------------------------------------------------------- stage 6
        // merge_hessian_gradient_cost.cpp:8-87
        void merge_hessian_gradient_cost(const int F, const int spline_deg_k, const double *frame_blocks_gpu,
                                         const int *ctrl_knot_start_indices, const int N, double *total_cost,
                                         double *H, double *g)
        {
            const int ndim = 6 * spline_deg_k + 1, E = ndim * (ndim + 1) / 2;
            std::vector<double> host((size_t)F * E);
            HIP_OR_DIE(hipMemcpy(host.data(), frame_blocks_gpu, sizeof(double) * host.size(), hipMemcpyDeviceToHost));
            merge_blocks_host(F, spline_deg_k, host.data(), ctrl_knot_start_indices, N, total_cost, H, g);
        }

        void solve_normal_equation(const double *A, const double *b, const int n, const int SolverType, double *x)
        {
            if (solve_normal_equation_host(A, b, n, SolverType, x) < 0)
            {
                fprintf(stderr, "ba_tracker: Solver is not implemented...\n");
                abort();
            }
        }

        // ---------------------------------------------------------------- storages
        namespace
        {
            std::mutex g_reg_mutex;
            std::map<const void *, Engine *> g_engines; // keyed by storages.cuda_frame_cost_gradient_hessian_tR
            std::map<const void *, int> g_formats;      // keyframe format of the storages' evaluations (set_keyframe_format; default 0)

            template <class T>
            void dev_alloc(T *&p, size_t count)
            {
                HIP_OR_DIE(hipMalloc((void **)&p, sizeof(T) * (count > 0 ? count : 1)));
            }
        } // namespace

        // spline_update_step.cpp:9-58.  Same fields; the per-sample scratch the reference
        // needs (cuda_vir_pixel_*: F*K*P*S*6k doubles, 6.3 GB at its defaults) is never
        // touched by this implementation and is allocated at token size.
        void initialize_shared_cuda_storages(const int max_num_frames, const int max_S, const int max_K,
                                             const int max_P, const int max_N, const int spline_deg_k,
                                             CudaSharedStorages &st)
        {
            const size_t nposes = (size_t)max_num_frames * max_S;
            const size_t npatch = (size_t)max_num_frames * max_K;
            const size_t npix = npatch * max_P;
            const int k = spline_deg_k;
            int nelems = k * 6 + 1;
            nelems = (1 + nelems) * nelems / 2;
            dev_alloc(st.cuda_img_cap_time, max_num_frames);
            dev_alloc(st.cuda_img_exp_time, max_num_frames);
            dev_alloc(st.cuda_keypoint_depth_z, max_K);
            dev_alloc(st.cuda_local_patch_pattern_xy, (size_t)max_P * 2);
            dev_alloc(st.cuda_cur_images, max_num_frames);
            dev_alloc(st.cuda_keypoint_xy, max_K);
            dev_alloc(st.cuda_keypoints_outlier_flags, max_K);
            HIP_OR_DIE(hipMemset(st.cuda_keypoints_outlier_flags, 0, max_K > 0 ? max_K : 1));
            dev_alloc(st.cuda_spline_ctrl_knots_data_t, (size_t)max_N * 3);
            dev_alloc(st.cuda_spline_ctrl_knots_data_R, (size_t)max_N * 4);
            dev_alloc(st.cuda_sampled_virtual_poses, nposes * 7);
            dev_alloc(st.cuda_J_virtual_pose_t_to_knots_t, nposes * 9 * k);
            dev_alloc(st.cuda_J_virtual_pose_R_to_knots_R, nposes * 12 * k);
            dev_alloc(st.cuda_jacobian_log_exp, 1);
            dev_alloc(st.cuda_temp_X_4x4, 1);
            dev_alloc(st.cuda_temp_Y_4x4, 1);
            dev_alloc(st.cuda_temp_Z_4x4, 1);
            dev_alloc(st.cuda_local_patches_XY, npatch);
            dev_alloc(st.cuda_pixel_residuals, npix);
            dev_alloc(st.cuda_pixel_jacobians_tR, npix * k * 6);
            dev_alloc(st.cuda_vir_pixel_to_ctrl_knots_tR, 1);
            dev_alloc(st.cuda_vir_pixel_residual, 1);
            dev_alloc(st.cuda_patch_cost_gradient_hessian_tR, npatch * nelems);
            dev_alloc(st.cuda_frame_cost_gradient_hessian_tR, (size_t)max_num_frames * nelems);
            st.num_bad_keypoints = 0;
            int dev = 0;
            HIP_OR_DIE(hipGetDevice(&dev));
            std::lock_guard<std::mutex> lk(g_reg_mutex);
            g_engines[st.cuda_frame_cost_gradient_hessian_tR] = new Engine(dev);
        }

        void free_shared_cuda_storages(CudaSharedStorages &st)
        {
            {
                std::lock_guard<std::mutex> lk(g_reg_mutex);
                auto it = g_engines.find(st.cuda_frame_cost_gradient_hessian_tR);
                if (it != g_engines.end()) { delete it->second; g_engines.erase(it); }
                g_formats.erase(st.cuda_frame_cost_gradient_hessian_tR);
            }
            void *ptrs[] = {st.cuda_img_cap_time, st.cuda_img_exp_time, st.cuda_keypoint_depth_z,
                            st.cuda_local_patch_pattern_xy, st.cuda_cur_images, st.cuda_keypoint_xy,
                            st.cuda_keypoints_outlier_flags, st.cuda_spline_ctrl_knots_data_t,
                            st.cuda_spline_ctrl_knots_data_R, st.cuda_sampled_virtual_poses,
                            st.cuda_J_virtual_pose_t_to_knots_t, st.cuda_J_virtual_pose_R_to_knots_R,
                            st.cuda_jacobian_log_exp, st.cuda_temp_X_4x4, st.cuda_temp_Y_4x4, st.cuda_temp_Z_4x4,
                            st.cuda_local_patches_XY, st.cuda_vir_pixel_to_ctrl_knots_tR, st.cuda_vir_pixel_residual,
                            st.cuda_pixel_residuals, st.cuda_pixel_jacobians_tR,
                            st.cuda_patch_cost_gradient_hessian_tR, st.cuda_frame_cost_gradient_hessian_tR};
            for (void *p : ptrs)
                if (p) HIP_OR_DIE(hipFree(p));
            st = CudaSharedStorages();
        }

        // ---------------------------------------------------------------- evaluate
        // spline_update_step.cpp:97-349 on the fused engine: three back-to-back launches,
        // one device sync, one D2H of F*E doubles, host scatter-add.
        void evaluate_cost_hessian_gradient(const int S, const int F, const unsigned char *cuda_ref_img,
                                            const float *cuda_dIxy_ref, const int K, const int P,
                                            const Core::VectorX<double, 4> &intrinsics,
                                            const Core::VectorX<int, 2> &im_size_HW, const int spline_deg_k,
                                            const double spline_start_time, const double spline_sample_dt,
                                            const int *cpu_ctrl_knot_start_indices, const int N,
                                            const CudaSharedStorages &st, const double huber_a, double *total_costs,
                                            double *cpu_hessian_tR, double *cpu_gradient_tR)
        {
            Engine *eng = nullptr;
            int format = 0;
            {
                std::lock_guard<std::mutex> lk(g_reg_mutex);
                auto it = g_engines.find(st.cuda_frame_cost_gradient_hessian_tR);
                if (it != g_engines.end()) eng = it->second;
                auto fi = g_formats.find(st.cuda_frame_cost_gradient_hessian_tR);
                if (fi != g_formats.end()) format = fi->second;
            }
            if (!eng)
            {
                fprintf(stderr, "ba_tracker: storages were not created by initialize_shared_cuda_storages\n");
                abort();
            }
            mbavo_problem p;
            memset(&p, 0, sizeof(p));
            p.S = S; p.F = F; p.K = K; p.P = P; p.N = N;
            p.H = im_size_HW.values[0]; p.W = im_size_HW.values[1];
            p.d_ref_img = cuda_ref_img; p.d_ref_dIxy = cuda_dIxy_ref;
            p.d_cur_imgs = (const unsigned char *const *)st.cuda_cur_images;
            p.d_kp_xy = &st.cuda_keypoint_xy->values[0]; p.kp_stride = 3; // Vector2d: 24-byte stride
            p.d_kp_z = st.cuda_keypoint_depth_z;
            p.d_pattern = st.cuda_local_patch_pattern_xy;
            p.d_outlier = st.cuda_keypoints_outlier_flags;
            p.num_bad = st.num_bad_keypoints;
            for (int i = 0; i < 4; ++i) p.intrinsics[i] = intrinsics.values[i];
            p.d_cap_time = st.cuda_img_cap_time; p.d_exp_time = st.cuda_img_exp_time;
            p.t0 = spline_start_time; p.dt = spline_sample_dt;
            p.d_knots_t = st.cuda_spline_ctrl_knots_data_t; p.d_knots_R = st.cuda_spline_ctrl_knots_data_R;
            p.h_start_idx = cpu_ctrl_knot_start_indices;
            p.huber_a = huber_a;
            p.grad_fp16 = format;
            const bool with_h = cpu_hessian_tR != nullptr;
            int rc = eng->evaluate(1, &p, spline_deg_k, with_h, st.cuda_frame_cost_gradient_hessian_tR, nullptr, nullptr,
                                   st.cuda_patch_cost_gradient_hessian_tR);
            if (rc != 0)
            {
                fprintf(stderr, "ba_tracker: evaluate_cost_hessian_gradient failed (%d)\n", rc);
                abort();
            }
            HIP_OR_DIE(hipStreamSynchronize(eng->stream()));
            merge_hessian_gradient_cost(F, spline_deg_k, st.cuda_frame_cost_gradient_hessian_tR,
                                        cpu_ctrl_knot_start_indices, N, total_costs, cpu_hessian_tR, cpu_gradient_tR);
        }

        void pack_keyframe(const unsigned char *cuda_img, const int H, const int W, unsigned int *cuda_packed)
        {
            const int rc = mbavo_pack_keyframe_u8(cuda_img, H, W, cuda_packed, nullptr);
            if (rc != 0) { fprintf(stderr, "ba_tracker: pack_keyframe failed (%d)\n", rc); abort(); }
            HIP_OR_DIE(hipStreamSynchronize(nullptr));
        }

        void set_keyframe_format(const CudaSharedStorages &st, const int format)
        {
            if (format < 0 || format > 2) { fprintf(stderr, "ba_tracker: keyframe format %d (0 float pairs, 1 half pairs, 2 packed words)\n", format); abort(); }
            std::lock_guard<std::mutex> lk(g_reg_mutex);
            g_formats[st.cuda_frame_cost_gradient_hessian_tR] = format;
        }
    } // namespace VO
} // namespace SLAM
