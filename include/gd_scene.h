/*
 * gd_scene.h -- C-ABI of the scene-side kernels either side of the rasterizer,
 * exported by libgd_raster.so.  Plain device pointers, caller's HIP stream, no torch types.
 *
 *   gd_scene_dist2        <- distCUDA2 / SimpleKNN::knn
 *                            (Garment_3DGS/gaussiansplatting/submodules/simple-knn/simple_knn.cu:63-220,
 *                             spatial.cu:14-25): mean squared distance to the 3 nearest neighbours, used once by
 *                            GaussianModel.create_from_pcd (scene/gaussian_model.py:135-136) for the initial scales.
 *   gd_scene_adam_step    <- torch.optim.Adam(l, lr=0.0, eps=1e-15).step() over the six per-attribute groups
 *                            (scene/gaussian_model.py:156-167): ONE launch over a flat fp32 parameter buffer with a
 *                            per-group learning rate (the flat gradient buffer is also what the view-sharded
 *                            all-reduce sends, garmentdreamer_amd/dist.py).
 *   gd_scene_activate_*   <- get_features / get_opacity / get_scaling / get_rotation (scene/gaussian_model.py:95-115)
 *                            and their autograd nodes: one launch forward, one backward.
 *   gd_scene_densify_stats <- on_before_optimizer_step + add_densification_stats
 *                            (Garment_3DGS/threestudio/systems/GaussianDreamer.py:268-279,
 *                             scene/gaussian_model.py:415-419): max_radii2D / xyz_gradient_accum / denom update.
 *
 *   gd_scene_densify_plan / _apply <- GaussianModel.densify_and_prune = densify_and_clone + densify_and_split +
 *                            prune_points with their optimizer-state surgery (scene/gaussian_model.py:283-413): the whole
 *                            event as a classification pass and ONE sweep that writes the new flat parameter /
 *                            exp_avg / exp_avg_sq buffers (csrc/raster_densify.hip).
 *
 * Return 0 on success, negative on error (gd_scene_last_error(); gd_scene_densify_last_error() for the last two).
 */
#ifndef GD_SCENE_H_INCLUDED
#define GD_SCENE_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GD_SCENE_MAX_GROUPS 8
#define GD_SCENE_KNN_BOX 1024 /* BOX_SIZE of simple_knn.cu */

/* bytes of device scratch gd_scene_dist2 needs for P points */
size_t gd_scene_dist2_scratch_bytes(int P);

/* points: float [P][3]; mean_dists: float [P] out; scratch: gd_scene_dist2_scratch_bytes(P) bytes.
 * Same algorithm and quirks as the reference: bounding box reduced with an initial value of 0 (so it always
 * contains the origin), 30-bit Morton codes of the truncated normalised coordinates, stable sort, boxes of 1024
 * consecutive codes, 3 best squared distances averaged as (b0 + b1 + b2) / 3.  P < 4 follows the reference
 * too (missing neighbours stay FLT_MAX). */
int gd_scene_dist2(void* stream, int P, const float* points, float* mean_dists, void* scratch);

/* One Adam step over a flat parameter buffer split into ngroups consecutive ranges [group_end[g-1], group_end[g])
 * (elements), each with its own learning rate; torch.optim.Adam semantics (no amsgrad, no weight decay):
 *   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps),  t = step >= 1
 * Scalars are doubles, as torch's Python-side hyper-parameters are (1 - beta and lr / (1 - beta1^t) are formed in
 * double before the fp32 kernel sees them). */
int gd_scene_adam_step(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                       int ngroups, const int64_t* group_end, const double* lr, double beta1, double beta2, double eps,
                       int step);

/* vis = radii > 0:  max_radii2D = vis ? max(max_radii2D, radii) : max_radii2D;
 *                   xyz_gradient_accum += vis ? |viewspace_grad.xy| : 0;   denom += vis ? 1 : 0.
 * radii: int32 [P]; viewspace_grad: float [P][3]; the three accumulators: float [P]. */
int gd_scene_densify_stats(void* stream, int P, const int* radii, const float* viewspace_grad, float* max_radii2D,
                           float* xyz_gradient_accum, float* denom);

/* Parameter activations of GaussianModel (get_features / get_opacity / get_scaling / get_rotation,
 * scene/gaussian_model.py:95-115) in one launch: shs = cat(f_dc, f_rest) [P][M][3], opacity = sigmoid,
 * scales = exp, rotations = q / max(|q|, 1e-12) (torch.nn.functional.normalize).  M = (sh_degree + 1)^2;
 * f_rest may be NULL when M == 1. */
int gd_scene_activate_forward(void* stream, int P, int M, const float* f_dc, const float* f_rest, const float* opacity_raw,
                              const float* scaling_raw, const float* rotation_raw, float* shs, float* opacity,
                              float* scales, float* rotations);

/* Its input gradients, ACCUMULATED into g_* (the flat gradient buffer of the scene, zeroed once per step):
 * opacity / scales are the activated outputs of the forward, d_* the gradients w.r.t. them (any may be NULL). */
int gd_scene_activate_backward(void* stream, int P, int M, const float* opacity, const float* scales,
                               const float* rotation_raw, const float* d_shs, const float* d_opacity,
                               const float* d_scales, const float* d_rotations, float* g_f_dc, float* g_f_rest,
                               float* g_opacity, float* g_scaling, float* g_rotation);

/* ---- densify_and_prune (scene/gaussian_model.py:398-413) --------------------------------------------------------
 * With g = xyz_gradient_accum / denom (NaN -> 0), s = max exp(scaling_raw), o = sigmoid(opacity_raw):
 *   clone  sqrt(g g) >= grad_threshold and s <= dense_extent   (dense_extent = percent_dense * scene_extent)
 *   split  g >= grad_threshold and s > dense_extent            (two children, N = 2; the original is removed)
 *   prune  o < min_opacity or (max_world_scale >= 0 and s > max_world_scale), evaluated on the new point set
 *          (max_world_scale = 0.1 * extent when the caller passes a max_screen_size, negative otherwise; the reference's
 *          max_radii2D > max_screen_size test never fires: densification_postfix has zeroed max_radii2D by then).
 * plan: classifies the P points, leaves the plan in `scratch` (gd_scene_densify_scratch_bytes(P) bytes) and returns
 *   totals_host[4] = {originals kept, clones kept, points selected for splitting, children kept PER COPY}
 *   (one stream synchronisation; new P = totals[0] + totals[1] + 2 totals[3]).
 * apply: writes the new buffers.  The flat layout is group-major: group g holds width[g] floats per point, groups
 *   back to back (P points in the old buffers, new P in the new ones); g_xyz / g_scaling / g_rotation name the groups
 *   the split transform rewrites (widths 3 / 3 / 4).  normals: float [2 * totals[2]][3] standard-normal samples (row
 *   n * totals[2] + j belongs to copy n of the j-th selected point, the order torch.normal(mean, std) fills
 *   stds.repeat(2, 1)); children get xyz = R(q) (normal * exp(scaling)) + xyz, scaling = log(exp(scaling) * (1/1.6)),
 *   clones and children zero Adam moments.  New point order: originals, clones, first children, second children.  */
size_t gd_scene_densify_scratch_bytes(int P);
int gd_scene_densify_plan(void* stream, int P, const float* xyz_gradient_accum, const float* denom,
                          const float* opacity_raw, const float* scaling_raw, float grad_threshold, float dense_extent,
                          float min_opacity, float max_world_scale, void* scratch, uint32_t* totals_host);
int gd_scene_densify_apply(void* stream, int P, int ngroups, const int* width, int g_xyz, int g_scaling, int g_rotation,
                           const uint32_t* totals_host, const void* scratch, const float* normals, const float* flat,
                           const float* exp_avg, const float* exp_avg_sq, float* new_flat, float* new_exp_avg,
                           float* new_exp_avg_sq);
const char* gd_scene_densify_last_error(void);

const char* gd_scene_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
