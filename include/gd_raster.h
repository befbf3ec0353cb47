/*
 * gd_raster.h -- C-ABI of the MI355X (gfx950) differentiable Gaussian rasterizer.
 *
 * Drop-in boundary for GarmentDreamer's diff-gaussian-rasterization (DGR =
 * Garment_3DGS/gaussiansplatting/submodules/diff-gaussian-rasterization).  Every entry
 * point below replaces one static method of CudaRasterizer::Rasterizer
 * (DGR/cuda_rasterizer/rasterizer.h:20-91) -- the functions DGR/rasterize_points.cu binds
 * to torch -- with plain pointers, sizes and a stream: no torch types, no C++ types, no
 * exceptions across the boundary.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 / int32 memory unless stated;
 *   - a NULL pointer means "absent optional" (the reference passes torch.Tensor([]) whose
 *     data_ptr is null: DGR/cuda_rasterizer/forward.cu:205,241);
 *   - viewmatrix / projmatrix are the reference's transposed (row-vector) 4x4 matrices
 *     (DGR/cuda_rasterizer/auxiliary.h:58-97);
 *   - the library is stateless; the caller owns every buffer (rasterizer.h has only static
 *     methods).  Scratch ("geometry", "binning", "image" byte buffers) must survive from
 *     forward to backward exactly like the reference's geomBuffer/binningBuffer/imgBuffer
 *     (DGR/diff_gaussian_rasterization/__init__.py:95-97);
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); the only host
 *     synchronisation is the 4-byte read-back of num_rendered inside gd_raster_forward,
 *     mirroring DGR/cuda_rasterizer/rasterizer_impl.cu:282;
 *   - functions return >= 0 on success and a negative GD_ERR_* code on failure;
 *     gd_raster_last_error() returns a thread-local message.
 */
#ifndef GD_RASTER_H_INCLUDED
#define GD_RASTER_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GD_OK 0
#define GD_ERR_INVALID_ARG (-1)
#define GD_ERR_HIP (-2)
#define GD_ERR_ALLOC (-3)
#define GD_ERR_NON_RGB (-4)

#define GD_MAX_VIEWS 16

/* Scratch allocator callback: must return a device pointer to at least `bytes` bytes (any
 * alignment; the library re-aligns to 128 B and the byte counts below include the slack).
 * Replaces the std::function<char*(size_t)> resize lambdas of
 * DGR/rasterize_points.cu:27-33,78-80. */
typedef char* (*gd_alloc_fn)(void* user, size_t bytes);

/* Scratch sizes (bytes).  Backward re-derives every sub-buffer from (P, R, W, H, V) alone,
 * like GeometryState/ImageState/BinningState::fromChunk (rasterizer_impl.cu:155-193). */
size_t gd_raster_geom_bytes(int P, int V);
size_t gd_raster_image_bytes(int width, int height, int V);
size_t gd_raster_binning_bytes(int64_t R);
size_t gd_raster_backward_scratch_bytes(int P, int V, int64_t R);   /* R = num_rendered of the forward call */

/* Replaces CudaRasterizer::Rasterizer::forward (rasterizer.h:32-57,
 * rasterizer_impl.cu:197-339).  Returns num_rendered (the number of (Gaussian, tile)
 * instances; the Python-visible int of DGR/diff_gaussian_rasterization/__init__.py:92,96).
 * out_color[3,H,W], out_depth[1,H,W], out_alpha[1,H,W], radii[P] (int32, may be NULL). */
int gd_raster_forward(void* stream, gd_alloc_fn geom_alloc, void* geom_user, gd_alloc_fn binning_alloc,
                      void* binning_user, gd_alloc_fn image_alloc, void* image_user, int P, int D, int M,
                      const float* background, int width, int height, const float* means3D, const float* shs,
                      const float* colors_precomp, const float* opacities, const float* scales,
                      float scale_modifier, const float* rotations, const float* cov3D_precomp,
                      const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                      float tan_fovy, int prefiltered, float* out_color, float* out_depth, float* out_alpha,
                      int* radii, int debug);

/* Replaces CudaRasterizer::Rasterizer::backward (rasterizer.h:59-91,
 * rasterizer_impl.cu:343-446).  `alphas` is the forward's out_alpha (the fork reconstructs
 * T_final = 1 - alpha from it: DGR/cuda_rasterizer/backward.cu:463).  Outputs need NOT be
 * zeroed by the caller (the reference requires torch::zeros, rasterize_points.cu:155-164;
 * here every element is written).  bwd_scratch: gd_raster_backward_scratch_bytes(P, 1, R).
 * dL_dmean2D[P,3] dL_dconic[P,2,2] dL_dopacity[P] dL_dcolor[P,3] dL_ddepth[P]
 * dL_dmean3D[P,3] dL_dcov3D[P,6] dL_dsh[P,M,3] dL_dscale[P,3] dL_drot[P,4]. */
int gd_raster_backward(void* stream, int P, int D, int M, int R, const float* background, int width, int height,
                       const float* means3D, const float* shs, const float* colors_precomp, const float* alphas,
                       const float* scales, float scale_modifier, const float* rotations,
                       const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                       const float* campos, float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer,
                       char* binning_buffer, char* image_buffer, char* bwd_scratch, const float* dL_dpix,
                       const float* dL_dpix_depth, const float* dL_dalphas, float* dL_dmean2D, float* dL_dconic,
                       float* dL_dopacity, float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D,
                       float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int debug);

/* Replaces CudaRasterizer::Rasterizer::markVisible (rasterizer.h:24-30,
 * rasterizer_impl.cu:141-153).  present: P bytes (bool). */
int gd_raster_mark_visible(void* stream, int P, const float* means3D, const float* viewmatrix,
                           const float* projmatrix, uint8_t* present);

/* ---- Batched multi-view entry (no counterpart in the reference, which loops
 * over views in Python: Garment_3DGS/threestudio/systems/GaussianDreamer.py:189-191).
 * One launch set renders V <= GD_MAX_VIEWS views of the SAME Gaussians: instance keys are
 * (view*tiles + tile) << 32 | depth_bits, one sort, one host sync.  viewmatrix/projmatrix:
 * [V,4,4]; cam_pos: [V,3]; tan_fovx/tan_fovy: HOST arrays [V]; out_*: [V,C,H,W];
 * radii: [V,P].  Scratch sizes use the same V. */
int gd_raster_forward_batched(void* stream, int V, gd_alloc_fn geom_alloc, void* geom_user,
                              gd_alloc_fn binning_alloc, void* binning_user, gd_alloc_fn image_alloc,
                              void* image_user, int P, int D, int M, const float* background, int width,
                              int height, const float* means3D, const float* shs, const float* colors_precomp,
                              const float* opacities, const float* scales, float scale_modifier,
                              const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                              const float* projmatrix, const float* cam_pos, const float* tan_fovx,
                              const float* tan_fovy, int prefiltered, float* out_color, float* out_depth,
                              float* out_alpha, int* radii, int debug);

/* The same forward pass WITHOUT the host read-back of the instance count (rasterizer_impl.cu:282 reads num_rendered back to
 * size the binning buffer: one stream synchronisation per forward pass, the only one of the iteration).  The caller names a
 * CAPACITY instead: the binning buffer is allocated for `capacity` instances (gd_raster_binning_bytes(capacity)), every grid
 * is sized for it, and the kernels read the live count from count_dev[1]:
 *   count_dev[0] = num_rendered of this call, [1] = the count the kernels worked on (= [0], or 0 after an overflow),
 *   count_dev[2] = 1 if num_rendered > capacity, 2 if a tile's list is longer than the tile-bucketed binning takes (4096
 *   instances; round 6) -- either way NOTHING was binned: every view shows the background; the caller reads the flag whenever
 *   it likes -- a deferred, asynchronous copy -- and must treat the call's results as void; after a 2 it repeats the call with
 *   a NEGATIVE capacity: |capacity| instances on the radix-sort binning, which has no such limit --, [3] = capacity.
 * Returns `capacity` (>= 0) -- the value to pass as R to gd_raster_backward_batched, gd_raster_backward_scratch_bytes and
 * gd_raster_get_layout, which then use the same layout -- or a negative error.  Results for the live instances are bit-identical
 * to gd_raster_forward_batched.  No reference counterpart (the reference synchronises). */
int gd_raster_forward_batched_capacity(void* stream, int V, gd_alloc_fn geom_alloc, void* geom_user,
                                       gd_alloc_fn binning_alloc, void* binning_user, gd_alloc_fn image_alloc,
                                       void* image_user, int P, int D, int M, const float* background, int width,
                                       int height, const float* means3D, const float* shs, const float* colors_precomp,
                                       const float* opacities, const float* scales, float scale_modifier,
                                       const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                                       const float* projmatrix, const float* cam_pos, const float* tan_fovx,
                                       const float* tan_fovy, int prefiltered, float* out_color, float* out_depth,
                                       float* out_alpha, int* radii, int debug, int64_t capacity, uint32_t* count_dev);

/* Batched backward.  dL_dpix [V,3,H,W], dL_dpix_depth/dL_dalphas/alphas [V,1,H,W].
 * dL_dmean2D is PER VIEW [V,P,3] (densification statistics need it per view,
 * GaussianDreamer.py:270-276); all other outputs are summed over the V views. */
int gd_raster_backward_batched(void* stream, int V, int P, int D, int M, int R, const float* background,
                               int width, int height, const float* means3D, const float* shs,
                               const float* colors_precomp, const float* alphas, const float* scales,
                               float scale_modifier, const float* rotations, const float* cov3D_precomp,
                               const float* viewmatrix, const float* projmatrix, const float* campos,
                               const float* tan_fovx, const float* tan_fovy, const int* radii, char* geom_buffer,
                               char* binning_buffer, char* image_buffer, char* bwd_scratch, const float* dL_dpix,
                               const float* dL_dpix_depth, const float* dL_dalphas, float* dL_dmean2D,
                               float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
                               float* dL_dsh, float* dL_dscale, float* dL_drot, int debug);

/* ---- Introspection for the parity tests (tests/ compare these integer buffers bit-for-bit
 * with the oracle).  Writes byte offsets of the named sub-buffers inside the three scratch
 * chunks, relative to the 128-B-aligned chunk base the library derives from `base`. */
typedef struct gd_raster_layout {
    /* geometry chunk (per (view, Gaussian) index vp = v*P + g) */
    size_t depths, clamped, radii, means2D, cov3D, conic_opacity, rgb, tiles_touched, point_offsets, block_sums;
    /* image chunk; pair_counts[pixel] = {list entries visited by the forward pass, entries blended} */
    size_t ranges, n_contrib, pair_counts;
    /* binning chunk */
    size_t point_list, point_list_alt, keys, keys_alt, sort_hist;
} gd_raster_layout;
int gd_raster_get_layout(const char* geom_base, const char* image_base, const char* binning_base, int P, int V,
                         int width, int height, int64_t R, gd_raster_layout* out);

/* Number of radix passes / sorted key bits used for a tile grid (getHigherMsb,
 * rasterizer_impl.cu:35-50,301). */
int gd_raster_sort_bits(int width, int height, int V);
/* y[i] = the alpha blend's exp as the kernels evaluate it (device pointers): parity hook -- must equal
 * oracle/gd_oracle.c gd_expf bit for bit. */
int gd_raster_blend_exp(void* stream, const float* x, float* y, int n);

/* Test / tuning hook: which binning the forward pass uses.  1 (default; environment GD_RASTER_BUCKETS=0 starts at 0): tile-
 * bucketed -- count per tile, prefix sum (= ranges), scatter into the tile's bucket, one LDS sort per tile; falls back to the
 * radix path when a list exceeds 4096 instances.  0: the global radix sort of (tile | depth) keys (what the reference does with
 * cub, rasterizer_impl.cu:304-309) always.  -1: back to the default.  Both leave bit-identical keys / point_list / ranges. */
int gd_raster_force_binning(int mode);

/* Test hook: fills the LDS of every CU with NaN bit patterns (LDS is not cleared between workgroups), so that a kernel
 * reading a shared-memory cell it never wrote shows up in the parity tests rather than as a rare non-finite step. */
int gd_raster_poison_lds(void* stream);

/* ---- Per-kernel timing for bench.py's roofline line.  When enabled, every launch of the
 * listed kernels is bracketed by hipEvents on the launch stream; gd_raster_profile_collect()
 * waits for the recorded events and folds them into per-kernel totals.  Disabled by default
 * (zero overhead).  Kernel ids: */
#define GD_K_PREPROCESS 0
#define GD_K_SCAN 1
#define GD_K_DUPLICATE 2
#define GD_K_SORT 3
#define GD_K_RANGES 4
#define GD_K_RENDER_FWD 5
#define GD_K_RENDER_BWD 6
#define GD_K_PREPROCESS_BWD 7
#define GD_K_COUNT 8
int gd_raster_profile_enable(int on);
int gd_raster_profile_collect(void);
int gd_raster_profile_get(int kernel_id, double* total_ms, int64_t* launches);
int gd_raster_profile_reset(void);
const char* gd_raster_profile_kernel_name(int kernel_id);

const char* gd_raster_last_error(void);
const char* gd_raster_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* GD_RASTER_H_INCLUDED */
