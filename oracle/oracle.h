/*
 * oracle.h — CPU oracle for the Vulcan fusion + raycast hot path.
 *
 * TEST INFRASTRUCTURE ONLY. This is a scalar C restatement of the reference's
 * CUDA kernels, written loop-iteration-for-thread, float32, with FP
 * contraction disabled (-ffp-contract=off). Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it. Nothing under vulcan_amd/ may
 * include, link or import anything from this directory.
 *
 * Pinning: the reference's kernels cannot be built here (CUDA, OpenCV), so the
 * oracle is pinned by the reference's own closed-form test cases, restated in
 * tests/test_oracle_*.py with the reference's tolerances:
 * integrator_test.cu:82-221, volume_test.cpp:100-556, tracer_test.cu:22-588,
 * depth_tracker_test.cu:12-126, and the layout / hash / projection known
 * answers in tests/golden/reference_kats.json.
 *
 * All pointers are HOST pointers. The PODs are those of include/vk.h.
 */
#ifndef ORACLE_H_
#define ORACLE_H_

#include "../include/vk.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Which racing writer wins an allocation request slot (volume.cu:200,237).
 * SERIAL: threads run one after another in raster order, last writer wins —
 *         one legal outcome of the reference's race.
 * MAXKEY: the request with the largest packed (pad=type,z,y,x) key wins — the
 *         rule the HIP path implements with a 64-bit atomic max. */
enum { ORC_POLICY_SERIAL = 0, ORC_POLICY_MAXKEY = 1 };

int  orc_version(void);
void orc_set_threads(int n);      /* OpenMP threads for the order-independent loops */
int  orc_get_threads(void);

/* volume.cu */
void orc_volume_initialize(const vk_volume* v);
void orc_volume_reset_block_visibility(const vk_volume* v);
void orc_volume_create_allocation_requests(const vk_volume* v, const float* depth,
    int width, int height, const vk_projection* projection,
    const vk_transform* Twd, int policy);
void orc_volume_handle_allocation_requests(const vk_volume* v);
void orc_volume_update_block_visibility(const vk_volume* v, int width, int height,
    const vk_projection* projection, const vk_transform* Tdw);
void orc_volume_set_view(const vk_volume* v, const vk_frame* frame, int policy);

/* *_integrator.cu */
void orc_integrate_depth(const vk_volume* v, const vk_integrator* p, const vk_frame* f);
void orc_integrate_color(const vk_volume* v, const vk_integrator* p, const vk_frame* f);
void orc_light_compute_frame_mask(const vk_frame* f, float depth_threshold, float* mask);
void orc_integrate_light_color(const vk_volume* v, const vk_integrator* p,
    const vk_light* light, const float* mask, const vk_frame* f);

/* tracer.cu, frame.cu */
void orc_trace_compute_patches(const int32_t* indices, const vk_hash_entry* entries,
    const vk_transform* Tcw, const vk_projection* projection, float block_length,
    float min_depth, float max_depth, int block_count, int image_width,
    int image_height, int bounds_width, int bounds_height, vk_patch* patches,
    int patch_capacity, int32_t* patch_count);
void orc_trace_reset_bounds(float* bounds, int count);
void orc_trace_compute_bounds(const vk_patch* patches, float* bounds,
    int bounds_width, int patch_count);
/* steps (optional, may be NULL): per-pixel number of march iterations */
void orc_trace_compute_points(const vk_hash_entry* entries, const vk_voxel* voxels,
    const float* bounds, int block_count, float block_length, float voxel_length,
    float trunc_length, const vk_transform* Twc, const vk_projection* projection,
    float* depths, float* colors, int image_width, int image_height,
    int bounds_width, int bounds_height, int32_t* steps);
void orc_frame_compute_normals(const float* depths, const vk_projection* projection,
    float* normals, int image_width, int image_height);
void orc_frame_filter_depths(int image_width, int image_height, const float* src,
    float* dst);

/* image.cu */
void orc_image_downsample(int src_w, int src_h, const float* src, float* dst, int nearest);
void orc_color_image_downsample(int src_w, int src_h, const float* src, float* dst, int nearest);

/* depth_tracker.cu, tracker.cpp, depth_tracker.cpp */
void orc_icp_compute_residuals(const vk_icp_view* keyframe, const vk_transform* Twm,
    const vk_icp_view* frame, const vk_transform* Twc, float* residuals);
void orc_icp_compute_jacobian(const vk_icp_view* keyframe, const vk_transform* Twm,
    const vk_icp_view* frame, const vk_transform* Twc, int translation_enabled,
    float* jacobian);
/* hessian: double[21] packed lower triangle, gradient: double[6]; per-pixel
 * terms are float32 as in the reference, the sums are accumulated in double
 * (the reference's atomic float sums have no defined order). */
void orc_icp_compute_system(const vk_icp_view* keyframe, const vk_transform* Twm,
    const vk_icp_view* frame, const vk_transform* Twc, int translation_enabled,
    double* hessian, double* gradient);
/* tracker.cpp:124-163 + depth_tracker.cpp:22-86; returns ||x||. */
/* the 6x6 (or 3x3) solve of tracker.cpp:127,153-159 twice: LDL^T without pivoting (what the device
 * runs) and with Eigen's diagonal pivoting (its published algorithm); A: n*n row-major symmetric */
void orc_ldlt_solve(int n, const float* A, const float* b, float* x);
void orc_ldlt_solve_pivoted(int n, const float* A, const float* b, float* x);
float orc_icp_solve_update(const float* hessian_packed, const float* gradient,
    int translation_enabled, vk_transform* Twc, float* update);

/* known-answer hooks (tests/test_oracle_kats.py) */
/* colour tracker (oracle_color_tracker.c) */
void orc_color_image_convert(int total, const float* src, float* dst);
void orc_image_gradients(int width, int height, const float* src, float* gx, float* gy);
void orc_color_tracker_compute_residuals(const vk_color_view* keyframe, const vk_color_view* frame,
    const vk_transform* Tcm, float* residuals);
void orc_color_tracker_compute_jacobian(const vk_color_view* keyframe, const vk_color_view* frame,
    const vk_transform* Tcm, int translation_enabled, float* jacobian);
void orc_color_tracker_compute_system(const vk_color_view* keyframe, const vk_color_view* frame,
    const vk_transform* Tcm, int translation_enabled, double* hessian, double* gradient);
float orc_color_tracker_solve_update(const float* hessian_packed, const float* gradient,
    int translation_enabled, const vk_transform* frame_Tcd, const vk_transform* keyframe_Twc,
    vk_color_pose* pose, float* update_out);
void orc_color_tracker_tcm(const vk_transform* frame_Tcd, const vk_transform* keyframe_Twc,
    vk_color_pose* pose);

/* light tracker (oracle_color_tracker.c) */
void orc_light_tracker_compute_residuals(const vk_color_view* keyframe, const vk_color_view* frame,
    const vk_light_terms* terms, const vk_transform* Tcm, float* residuals);
void orc_light_tracker_compute_jacobian(const vk_color_view* keyframe, const vk_color_view* frame,
    const vk_light_terms* terms, const vk_transform* Tcm, int translation_enabled, float* jacobian);
void orc_light_tracker_compute_system(const vk_color_view* keyframe, const vk_color_view* frame,
    const vk_light_terms* terms, const vk_transform* Tcm, int translation_enabled, double* hessian,
    double* gradient);

/* detector (oracle_detect.c). PARITY UNPINNED: the reference holds no test or
 * vector for Detector (tests/detector_test.cu is empty). */
void orc_detect(const vk_detector* detector, const float* points, int count,
    float* inliers, vk_detect_state* state);

/* mesh extraction (oracle_extract.c). PARITY UNPINNED: upstream's extractor is unfinished
 * and has no test (tests/extractor_test.cpp is empty). */
int orc_extract_mesh(const vk_volume* v, int all_allocated, int interpolate, float* points, int point_capacity,
    int32_t* faces, int face_capacity, int32_t* counts);
int orc_mc_triangles(int state, signed char* edges15);

uint32_t orc_kat_hash(int bx, int by, int bz, uint32_t K);
void orc_kat_project(const vk_projection* k, float x, float y, float z, float* uv);
void orc_kat_unproject(const vk_projection* k, float u, float v, float d, float* xyz);
void orc_kat_sizes(int* out);

#ifdef __cplusplus
}
#endif
#endif
