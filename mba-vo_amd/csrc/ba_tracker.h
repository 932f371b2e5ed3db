// ba_tracker.h -- the reference's ba_tracker free-function API (namespace SLAM::VO),
// implemented MI355X-native.  Signatures, argument meaning, pointer ownership and
// the synchronous-at-return behaviour are those of the reference headers:
//   compute_virtual_camera_poses        ba_tracker/compute_virtual_camera_poses.h:18-33
//   compute_local_patches_xy            ba_tracker/compute_local_patches_xy.h:10-18
//   compute_pixel_jacobian_residual     ba_tracker/compute_hessian_gradients_cost.h:11-29
//   compute_patch_cost_gradient_hessian ba_tracker/compute_hessian_gradients_cost.h:52-60
//   compute_frame_cost_gradient_hessian ba_tracker/compute_hessian_gradients_cost.h:62-68
//   merge_hessian_gradient_cost         ba_tracker/merge_hessian_gradient_cost.h:8-15
//   solve_normal_equation               ba_tracker/solve_normal_equation.h:10-35
//   CudaSharedStorages, initialize/free_shared_cuda_storages, evaluate_cost_hessian_gradient
//                                       ba_tracker/spline_update_step.h:18-87
// Differences, all stricter: every HIP call is checked (failure prints and aborts,
// the reference checks nothing); solve_normal_equation works on flat column-major
// buffers because Eigen is not a dependency; the five launchers stay available one
// by one, but evaluate_cost_hessian_gradient runs the fused engine (engine.h)
// instead of five launches + five device syncs.
#ifndef MBAVO_BA_TRACKER_H
#define MBAVO_BA_TRACKER_H

#include "core_types.h"

namespace SLAM
{
    namespace VO
    {
        struct CudaSharedStorages
        {
            double *cuda_img_cap_time = nullptr;
            double *cuda_img_exp_time = nullptr;
            double *cuda_keypoint_depth_z = nullptr;
            Core::Vector2d *cuda_keypoint_xy = nullptr;
            unsigned char *cuda_keypoints_outlier_flags = nullptr;
            int num_bad_keypoints = 0;

            unsigned char **cuda_cur_images = nullptr;
            int *cuda_local_patch_pattern_xy = nullptr;

            double *cuda_spline_ctrl_knots_data_t = nullptr;
            double *cuda_spline_ctrl_knots_data_R = nullptr;

            double *cuda_sampled_virtual_poses = nullptr;
            double *cuda_J_virtual_pose_t_to_knots_t = nullptr;
            double *cuda_J_virtual_pose_R_to_knots_R = nullptr;
            double *cuda_jacobian_log_exp = nullptr;
            double *cuda_temp_X_4x4 = nullptr;
            double *cuda_temp_Y_4x4 = nullptr;
            double *cuda_temp_Z_4x4 = nullptr;

            Core::Vector2d *cuda_local_patches_XY = nullptr;

            double *cuda_pixel_residuals = nullptr;
            double *cuda_pixel_jacobians_tR = nullptr;
            FLOAT *cuda_vir_pixel_to_ctrl_knots_tR = nullptr;
            FLOAT *cuda_vir_pixel_residual = nullptr;

            double *cuda_patch_cost_gradient_hessian_tR = nullptr;
            double *cuda_frame_cost_gradient_hessian_tR = nullptr;
        };

        void initialize_shared_cuda_storages(const int max_num_frames,
                                             const int max_num_virtual_poses_per_frame,
                                             const int max_num_keypoints,
                                             const int max_patch_size,
                                             const int max_num_ctrl_knots,
                                             const int spline_deg_k,
                                             CudaSharedStorages &storages);

        void free_shared_cuda_storages(CudaSharedStorages &storages);

        void compute_virtual_camera_poses(const int n_vir_poses_per_frame,
                                          const int n_frames,
                                          const double *img_cap_time,
                                          const double *img_exp_time,
                                          const int spline_deg_k,
                                          const double spline_start_time,
                                          const double spline_sample_interval,
                                          const double *spline_ctrl_knots_data_t,
                                          const double *spline_ctrl_knots_data_R,
                                          double *sampled_virtual_poses,
                                          double *jacobian_virtual_pose_t_to_ctrl_knots = nullptr,
                                          double *jacobian_virtual_pose_R_to_ctrl_knots = nullptr,
                                          double *jacobian_log_exp = nullptr,
                                          double *temp_X_4x4 = nullptr,
                                          double *temp_Y_4x4 = nullptr,
                                          double *temp_Z_4x4 = nullptr);

        void compute_local_patches_xy(const int num_virtual_poses_per_frame,
                                      const int num_frames,
                                      const double *virtual_cam_poses,
                                      const Core::Vector2d *sparse_keypoints,
                                      const double *sparse_keypoints_z,
                                      const int num_keypoints,
                                      const Core::VectorX<double, 4> &intrinsics,
                                      const Core::VectorX<int, 2> &im_HW,
                                      Core::Vector2d *local_patches_xy);

        void compute_pixel_jacobian_residual(const unsigned char *I_ref,
                                             const float *dIxy_ref,
                                             unsigned char const *const *I_cur_imgs,
                                             const int num_vir_poses_per_frame,
                                             const int num_frames,
                                             const double *sampled_virtual_poses,
                                             const int spline_deg_k,
                                             const double *jacobian_virtual_pose_t_to_ctrl_knots,
                                             const double *jacobian_virtual_pose_R_to_ctrl_knots,
                                             const Core::Vector2d *local_patches_XY,
                                             const double *keypoints_z,
                                             const int num_keypoints,
                                             const int *local_patch_pattern_xy,
                                             const int patch_size,
                                             const Core::VectorX<double, 4> &intrinsics,
                                             const Core::VectorX<int, 2> &im_size_HW,
                                             FLOAT *jacobian_pixel_to_ctrl_knots_tR,
                                             double *pixel_residuals,
                                             double *pixel_jacobians_tR = nullptr);

        void compute_patch_cost_gradient_hessian(const int num_frames,
                                                 const int num_keypoints,
                                                 const int patch_size,
                                                 const int spline_deg_k,
                                                 const double *pixel_residuals,
                                                 const double *pixel_jacobians,
                                                 const double huber_a,
                                                 const double inv_num_residuals,
                                                 double *patch_cost_gradient_hessian);

        void compute_frame_cost_gradient_hessian(const int num_frames,
                                                 const int num_keypoints,
                                                 const int spline_deg_k,
                                                 const double *patch_cost_gradient_hessian,
                                                 const bool eval_gradient_hessian,
                                                 const unsigned char *keypoints_outlier_flags,
                                                 double *frame_cost_gradient_hessian);

        void merge_hessian_gradient_cost(const int num_frames,
                                         const int spline_deg_k,
                                         const double *frame_cost_gradient_hessian_gpu,
                                         const int *ctrl_knot_start_indices,
                                         const int num_ctrl_knots,
                                         double *total_cost,
                                         double *ctrl_knot_H_cpu = nullptr,
                                         double *ctrl_knot_g_cpu = nullptr);

        // x = -A^+ b (SolverType 0, Jacobi SVD, minimum-norm) or -A^-1 b (1, LDLT);
        // A is n x n column-major.
        void solve_normal_equation(const double *A, const double *b, const int n, const int SolverType, double *x);

        void evaluate_cost_hessian_gradient(const int n_vir_poses_per_frame,
                                            const int n_frames,
                                            const unsigned char *cuda_ref_img,
                                            const float *cuda_dIxy_ref,
                                            const int num_keypoints,
                                            const int patch_size,
                                            const Core::VectorX<double, 4> &intrinsics,
                                            const Core::VectorX<int, 2> &im_size_HW,
                                            const int spline_deg_k,
                                            const double spline_start_time,
                                            const double spline_sample_dt,
                                            const int *cpu_ctrl_knot_start_indices,
                                            const int num_ctrl_knots,
                                            const CudaSharedStorages &storages,
                                            const double huber_a,
                                            double *total_costs,
                                            double *cpu_hessian_tR,
                                            double *cpu_gradient_tR);

        // ---- extension (not in the reference): the PACKED keyframe, one 32-bit word per pixel holding the intensity and both
        // central differences of Gradient.h:16-75 (include/mbavo.h: mbavo_pack_keyframe_u8; exact for 8-bit images, 4 instead of
        // 9 bytes per pixel).  A caller with many keyframes in flight packs each once,
        //     pack_keyframe(cuda_ref_img, H, W, cuda_packed);          set_keyframe_format(storages, 2);
        // and hands cuda_packed to evaluate_cost_hessian_gradient in cuda_dIxy_ref's place (cuda_ref_img stays valid: cost-only
        // evaluations tap it).  Format 0 (the default) is the reference's float [dx, dy] image.
        void pack_keyframe(const unsigned char *cuda_img, const int H, const int W, unsigned int *cuda_packed);
        void set_keyframe_format(const CudaSharedStorages &storages, const int format);
    } // namespace VO
} // namespace SLAM

#endif
