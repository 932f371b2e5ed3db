// ba_tracker_compat.h -- the reference's HOST-CALLABLE helpers under their own names and signatures (SURVEY.md 8b:
// "plus host-callable ..."), so that code written against the reference's headers -- its module harness first of all
// -- compiles against this library:
//
//   SLAM::Core::SplineSegmentStartKnotIdxAndNormalizedU   core/common/SplineFunctor.h:13-19
//   SLAM::Core::C2/C4SplineVec3Functor                    core/common/SplineFunctor.h:21-94   (3 x 3k Jacobian, row-major)
//   SLAM::Core::C2/C4SplineRot3Functor                    core/common/SplineFunctor.h:155-365 (4 x 3k Jacobian, row-major)
//   SLAM::VO::bilinear_interpolation<T>                   ba_tracker/compute_pixel_intensity.h:25-72
//   SLAM::VO::compute_pixel_intensity<T>                  ba_tracker/compute_pixel_intensity.h:91-209 (1 x 7 Jacobian)
//   SLAM::Core::Image<T>                                  core/measurements/Image.h:10-136 (uploadToGpu -> hipMalloc / hipMemcpy)
//
// All of them are thin wrappers over se3_math.h / pixel_math.h, the code the kernels run: same results as the device
// path, callable on host and device.  The scratch arguments of the rotation functors (log/exp Jacobian and X/Y/Z
// work areas) are accepted and ignored.
#ifndef MBAVO_BA_TRACKER_COMPAT_H
#define MBAVO_BA_TRACKER_COMPAT_H

#include "core_types.h"
#include "pixel_math.h"
#include "se3_math.h"

#include <cstring>
#include <hip/hip_runtime.h>

namespace SLAM
{
    namespace Core
    {
        MBAVO_HD void SplineSegmentStartKnotIdxAndNormalizedU(double t, double ctrlKnot_t0, double ctrlKnotSampFreq,
                                                              int &start_indx, double &u)
        {
            mbavo::spline_segment(t, ctrlKnot_t0, ctrlKnotSampFreq, start_indx, u);
        }

        template <int KD>
        MBAVO_HD Vector3d spline_vec3_functor(const double *data_knots, double u, double *jacobian)
        {
            double c[KD], p[3];
            mbavo::trans_coeffs<KD>(u, c);
            mbavo::spline_translation<KD>(data_knots, c, p);
            if (jacobian)
            { // J = kron(coefficients, I3): 3 x 3k row-major
                for (int i = 0; i < 9 * KD; ++i) jacobian[i] = 0.0;
                for (int a = 0; a < 3; ++a)
                    for (int j = 0; j < KD; ++j) jacobian[a * 3 * KD + 3 * j + a] = c[j];
            }
            return Vector3d(p[0], p[1], p[2]);
        }
        MBAVO_HD Vector3d C2SplineVec3Functor(const double *data_knots, double u, double *jacobian = nullptr)
        {
            return spline_vec3_functor<2>(data_knots, u, jacobian);
        }
        MBAVO_HD Vector3d C4SplineVec3Functor(const double *data_knots, double u, double *jacobian = nullptr)
        {
            return spline_vec3_functor<4>(data_knots, u, jacobian);
        }

        MBAVO_HD Quaterniond C2SplineRot3Functor(const double *data_knots, double u, double *jacobian_4x6 = nullptr,
                                                 double * /*jacobian_log_exp_2x12*/ = nullptr, double * /*X_4x4*/ = nullptr,
                                                 double * /*Y_4x4*/ = nullptr, double * /*Z_4x4*/ = nullptr)
        {
            const mbavo::Quat q = jacobian_4x6 ? mbavo::spline_rotation<2, true>(data_knots, u, jacobian_4x6)
                                               : mbavo::spline_rotation<2, false>(data_knots, u, nullptr);
            return Quaterniond(q.x, q.y, q.z, q.w);
        }
        MBAVO_HD Quaterniond C4SplineRot3Functor(const double *data_knots, double u, double *jacobian_4x12 = nullptr,
                                                 double * /*jacobian_log_exp_6x12*/ = nullptr, double * /*X_4x4*/ = nullptr,
                                                 double * /*Y_4x4*/ = nullptr, double * /*Z_4x4*/ = nullptr)
        {
            const mbavo::Quat q = jacobian_4x12 ? mbavo::spline_rotation<4, true>(data_knots, u, jacobian_4x12)
                                                : mbavo::spline_rotation<4, false>(data_knots, u, nullptr);
            return Quaterniond(q.x, q.y, q.z, q.w);
        }

        // Host image with an optional device copy (Image.h:10-136); the device side is HIP memory.
        template <typename T>
        class Image
        {
        public:
            Image() : m_data_cpu(nullptr), m_data_gpu(nullptr), m_nHeight(0), m_nWidth(0), m_nChannels(0) {}
            Image(size_t H, size_t W, size_t C) : m_data_cpu(nullptr), m_data_gpu(nullptr), m_nHeight(0), m_nWidth(0), m_nChannels(0)
            {
                allocate(H, W, C);
            }
            ~Image() { this->free(); }
            Image(const Image &) = delete;
            Image &operator=(const Image &) = delete;

            T *getData() { return m_data_cpu; }
            T *getGpuData() { return m_data_gpu; }
            T *getData(size_t r, size_t c) { return m_data_cpu + (r * m_nWidth + c) * m_nChannels; }
            size_t nHeight() { return m_nHeight; }
            size_t nWidth() { return m_nWidth; }
            size_t nChannels() { return m_nChannels; }

            void copyFrom(T *dataptr, size_t H, size_t W, size_t C)
            {
                if (H != m_nHeight || W != m_nWidth || C != m_nChannels)
                {
                    this->free();
                    allocate(H, W, C);
                }
                std::memcpy(m_data_cpu, dataptr, H * W * C * sizeof(T));
            }
            void free()
            {
                delete[] m_data_cpu;
                m_data_cpu = nullptr;
                if (m_data_gpu) (void)hipFree(m_data_gpu);
                m_data_gpu = nullptr;
                m_nHeight = m_nWidth = m_nChannels = 0;
            }
            void uploadToGpu()
            { // Image.h:125-136: uploaded once; a second call is a no-op
                if (m_data_cpu == nullptr || m_data_gpu != nullptr) return;
                const size_t bytes = sizeof(T) * m_nHeight * m_nWidth * m_nChannels;
                if (hipMalloc((void **)&m_data_gpu, bytes) != hipSuccess) { m_data_gpu = nullptr; return; }
                (void)hipMemcpy(m_data_gpu, m_data_cpu, bytes, hipMemcpyHostToDevice);
            }

        private:
            void allocate(size_t H, size_t W, size_t C)
            {
                m_data_cpu = new T[H * W * C]();
                m_nHeight = H; m_nWidth = W; m_nChannels = C;
            }
            T *m_data_cpu, *m_data_gpu;
            size_t m_nHeight, m_nWidth, m_nChannels;
        };
    } // namespace Core

    namespace VO
    {
        // I_and_dI = (intensity, dI/dx, dI/dy) at the sub-pixel position P2d; false outside [0, W-1] x [0, H-1]
        template <typename T>
        MBAVO_HD bool bilinear_interpolation(const unsigned char *I, const float *dIxy, const int im_H, const int im_W,
                                             const Core::VectorX<T, 2> &P2d, Core::Vector3d &I_and_dI)
        {
            double v = 0, gx = 0, gy = 0;
            bool ok;
            if (dIxy)
                ok = mbavo::bilinear_tap<true>(I, dIxy, im_H, im_W, (double)P2d.values[0], (double)P2d.values[1], v, gx, gy);
            else
                ok = mbavo::bilinear_tap<false>(I, dIxy, im_H, im_W, (double)P2d.values[0], (double)P2d.values[1], v, gx, gy);
            if (!ok) return false;
            I_and_dI.values[0] = v; I_and_dI.values[1] = gx; I_and_dI.values[2] = gy;
            return true;
        }

        // intensity of the keyframe seen through pixel cur_xy of a camera at (R_c2r xyzw, t_c2r) for a fronto-parallel
        // patch at plane_depth, and its 1 x 7 Jacobian [dI/dt (3) | dI/dq (4, xyzw)]
        template <typename T>
        MBAVO_HD bool compute_pixel_intensity(const unsigned char *I_ref, const float *dIxy_ref, const int I_H, const int I_W,
                                              const T *R_c2r, const T *t_c2r, const T plane_depth, const T fx, const T fy,
                                              const T cx, const T cy, const Core::VectorX<T, 2> &cur_xy, T *intensity,
                                              T *jacobian = nullptr)
        {
            mbavo::Camera cam;
            cam.fx = fx; cam.fy = fy; cam.cx = cx; cam.cy = cy; cam.H = I_H; cam.W = I_W;
            const double q[4] = {(double)R_c2r[0], (double)R_c2r[1], (double)R_c2r[2], (double)R_c2r[3]};
            const double t[3] = {(double)t_c2r[0], (double)t_c2r[1], (double)t_c2r[2]};
            double R[9], ray[3], val = 0, jt[3], b[4];
            mbavo::rotation_entries(q, R);
            mbavo::unit_ray(cam, (double)cur_xy.values[0], (double)cur_xy.values[1], ray);
            const double D = (double)plane_depth, iz = 1.0 / (D + 1e-8);
            bool ok;
            if (jacobian)
                ok = mbavo::sample_eval<true>(t, q, R, ray, D, iz, cam, I_ref, dIxy_ref, val, jt, b);
            else
                ok = mbavo::sample_eval<false>(t, q, R, ray, D, iz, cam, I_ref, dIxy_ref, val, jt, b);
            if (!ok) return false;
            *intensity = (T)val;
            if (jacobian)
            {
                for (int i = 0; i < 3; ++i) jacobian[i] = (T)jt[i];
                for (int i = 0; i < 4; ++i) jacobian[3 + i] = (T)b[i];
            }
            return true;
        }
    } // namespace VO
} // namespace SLAM

#endif
