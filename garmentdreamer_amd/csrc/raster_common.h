// raster_common.h -- shared declarations of the gfx950 Gaussian rasterizer kernels.
// Internal to garmentdreamer_amd/csrc; the public boundary is include/gd_raster.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/gd_raster.h"

namespace gd {

constexpr int kTile = 16;        // BLOCK_X = BLOCK_Y = 16 is part of the parity contract
constexpr int kTilePix = 256;    // (DGR/cuda_rasterizer/config.h:16-17)
constexpr int kGaussBlock = 256; // Gaussians per workgroup in per-Gaussian kernels

// Per-view scalars passed by value as a kernel argument (no device round trip).
struct ViewScalars {
    int V;
    float tan_fovx[GD_MAX_VIEWS];
    float tan_fovy[GD_MAX_VIEWS];
    float focal_x[GD_MAX_VIEWS];
    float focal_y[GD_MAX_VIEWS];
};

// Scratch carved from the caller's byte buffers; index vp = view * P + gaussian.
struct GeomState {
    uint8_t* clamped;         // [VP*3]
    int* radii;               // [VP] (used when the caller passes radii == NULL)
    float2* means2D;          // [VP]
    float* cov3D;             // [VP*6]
    float4* conic_opacity;    // [VP]
    float4* rgbd;             // [VP] colour (SH-evaluated or precomputed) + view-space depth
    uint32_t* tiles_touched;  // [VP]
    uint32_t* point_offsets;  // [VP] inclusive scan of tiles_touched
    uint32_t* block_sums;     // [ceil(VP/256) + 1] exclusive-scanned block totals, last = R
    float* alpha_thr;         // [VP] smallest exponent with alpha >= 1/255 for this opacity (raster_blend_math.h); written for
                              // visible Gaussians only, last so that the offsets of the fields above are unchanged
};
struct ImageState {
    uint2* ranges;        // [V*tiles]
    uint32_t* n_contrib;  // [V*H*W]
    uint2* pair_counts;   // [V*H*W] {visited, blended} per pixel (work accounting for the roofline)
    uint32_t* strip_count;  // [V*tiles*4] entries of each 16x4 strip's compact list (see BinningState::clist)
    uint32_t* tile_perm;    // [V*tiles] tiles by descending list length (launch order of the blend kernels; tile_order_kernel)
    uint32_t* tile_cursor;  // [V*tiles] tile-bucketed binning: next free position of each tile's bucket (tile_scatter)
    uint32_t* bin_stats;    // [4] tile-bucketed binning: {instances counted, longest tile list, 0, 0} (tile_scan_kernel)
};
struct BinningState {
    uint32_t* point_list;      // [R] sort payload = instance SLOT (see slot_vp); after tile_ranges: the Gaussian (vp) ids
    uint32_t* point_list_alt;  // [R] ping-pong buffer of the sort; after tile_ranges: slot_of[list position]
    uint32_t* slot_vp;         // [R] Gaussian (vp) of instance slot o; the slots of Gaussian vp are
                               // point_offsets[vp] - tiles_touched[vp] ... point_offsets[vp] - 1 (duplicate_kernel)
    uint64_t* keys;            // [R] sorted keys
    uint64_t* keys_alt;        // [R]
    uint32_t* sort_hist;       // [bins * nblk + bins]
    uint4* clist;              // [4R] compact per-strip lists written by render_forward, read by the backward blend:
                               // strip s of a tile with list range [x, y) owns entries [4x + s (y - x), + (y - x));
                               // an entry = {64-bit ballot of the strip's pixels that BLENDED it (lane = 16 * 4x4
                               // block + 4 * row + column), Gaussian (vp) id, instance slot}, in list order, only
                               // for entries with a non-zero ballot: the reverse pass never repeats the
                               // contribution test and never sees the pairs nobody blended
    uint4* rowpos;             // [R] per instance slot, per strip: 1 + the clist index of the slot's entry in that strip's
                               // list (0 = the strip blended nothing of it).  Zeroed by duplicate_kernel, written by
                               // render_forward; the backward blend stores its row of sums at that index (coalesced,
                               // in list order) and instance_sum_kernel finds a Gaussian's rows through it
};

GeomState carve_geom(char* chunk, size_t VP, size_t* used);
ImageState carve_image(char* chunk, size_t tiles_total, size_t pixels_total, size_t* used);
BinningState carve_binning(char* chunk, size_t R, size_t* used);

uint32_t higher_msb(uint32_t n);
struct SortPlan { int total_bits, live_bits, passes, digit_bits; };
SortPlan plan_sort(uint32_t tiles_total);
constexpr int kSortItems = 8;                      // keys per thread
constexpr int kSortTile = 256 * kSortItems;        // keys per workgroup
constexpr int kSortHistBits = 10;   // widest digit any caller sorts with (the 30-bit Morton codes: 3 x 10): sizes sort_hist
constexpr int kMaxDigitBits = 9;    // wider digits cost more per pass than they save in passes (raster_binning.hip)

// getRect (DGR/cuda_rasterizer/auxiliary.h:46-56): tile rectangle of a splat, clamped to the grid.
__device__ __forceinline__ void tile_rect(float px, float py, int max_radius, uint32_t gx, uint32_t gy,
                                          uint32_t& x0, uint32_t& y0, uint32_t& x1, uint32_t& y1)
{
    x0 = min(gx, (uint32_t)max(0, (int)((px - max_radius) / kTile)));
    y0 = min(gy, (uint32_t)max(0, (int)((py - max_radius) / kTile)));
    x1 = min(gx, (uint32_t)max(0, (int)((px + max_radius + kTile - 1) / kTile)));
    y1 = min(gy, (uint32_t)max(0, (int)((py + max_radius + kTile - 1) / kTile)));
}

// ---- launchers (one per translation unit) ----
void launch_preprocess(hipStream_t s, int P, int D, int M, const float* means3D, const float* scales,
                       float scale_modifier, const float* rotations, const float* opacities, const float* shs,
                       const float* cov3D_precomp, const float* colors_precomp, const float* viewmatrix,
                       const float* projmatrix, const float* cam_pos, int W, int H, const ViewScalars& vs,
                       int* radii, GeomState g, int tiles_x, int tiles_y, bool prefiltered);
void launch_mark_visible(hipStream_t s, int P, const float* means3D, const float* viewmatrix, uint8_t* present);
void launch_preprocess_backward(hipStream_t s, int P, int D, int M, int V, const float* means3D, const int* radii,
                                const float* shs, const uint8_t* clamped, const float* scales,
                                const float* rotations, float scale_modifier, const float* cov3D,
                                size_t cov3D_view_stride, const float* viewmatrix, const float* projmatrix,
                                const float* campos, const ViewScalars& vs, const float* rows /*[4R][10] by clist index*/,
                                const uint4* rowpos /*[R]*/, const uint32_t* point_offsets,
                                const uint32_t* tiles_touched, float* acc /*[VP][10] scratch, fully written*/,
                                bool colors_precomp, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                                float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D,
                                float* dL_dsh, float* dL_dscale, float* dL_drot, float* view_partials);

void launch_scan_block_sums(hipStream_t s, uint32_t* block_sums, uint32_t nblocks, uint32_t* info = nullptr, uint32_t capacity = 0);
void launch_duplicate(hipStream_t s, int VP, int P, const int* radii, GeomState g, uint64_t* keys_out,
                      uint32_t* vals_out, uint32_t* slot_vp, uint4* rowpos, int tiles_x, int tiles_y, const uint32_t* info = nullptr);
void launch_radix_sort(hipStream_t s, BinningState b, uint32_t R, SortPlan plan, bool start_in_alt, const uint32_t* n_dev = nullptr);
// identifyTileRanges + the sort's epilogue: slot_of[s] = point_list[s] (the slot), point_list[s] = its Gaussian id
void launch_tile_ranges(hipStream_t s, const uint64_t* keys, uint32_t R, uint2* ranges, uint32_t tiles_total,
                        uint32_t* point_list, const uint32_t* slot_vp, uint32_t* slot_of, const uint32_t* n_dev = nullptr);

void launch_tile_order(hipStream_t s, const uint2* ranges, uint32_t tiles_total, uint32_t* perm);

// ---- tile-bucketed binning (round 6): counting scatter by tile, then ONE LDS sort per tile ----
constexpr uint32_t kBucketMax = 4096;   // longest tile list the per-tile LDS sort takes (32 KiB of keys); beyond: radix path
void launch_tile_count(hipStream_t s, int VP, int P, const int* radii, GeomState g, uint2* ranges /* .x counts */, uint32_t tiles_total,
                       int tiles_x, int tiles_y);
void launch_tile_scan(hipStream_t s, uint2* ranges, uint32_t tiles_total, uint32_t* cursor, uint32_t* stats, uint32_t* info);
void launch_tile_scatter(hipStream_t s, int VP, int P, const int* radii, GeomState g, uint64_t* bucket_keys, uint32_t* slot_vp,
                         uint32_t* cursor, int tiles_x, int tiles_y, const uint32_t* info);
void launch_tile_sort(hipStream_t s, const uint2* ranges, uint32_t tiles_total, const uint64_t* bucket_keys, uint64_t* keys,
                      uint32_t* point_list, uint32_t* slot_of, const uint32_t* slot_vp);
void launch_blend_exp(hipStream_t s, const float* x, float* y, int n);   // y = gd_expf(x): parity test hook
void launch_poison_lds(hipStream_t s);                                   // NaN patterns into every CU's LDS: test hook
void launch_render_forward(hipStream_t s, int V, int W, int H, int tiles_x, int tiles_y, const uint2* ranges,
                           const uint32_t* point_list, const GeomState& g, const float* bg, float* out_color,
                           float* out_depth, float* out_alpha, uint32_t* n_contrib, uint2* pair_counts,
                           const uint32_t* slot_of, uint4* clist, uint32_t* strip_count, uint32_t* rowpos, const uint32_t* tile_perm = nullptr);
void launch_render_backward(hipStream_t s, int V, int W, int H, int tiles_x, int tiles_y, const uint2* ranges,
                            const GeomState& g, const float* bg, const float* alphas, const float* dL_dpix,
                            const float* dL_dpix_depth, const float* dL_dalphas,
                            float* rows /*[4R][10]: the row of sums of clist entry i at index i*/, const uint4* clist,
                            const uint32_t* strip_count, const uint32_t* tile_perm = nullptr);


}  // namespace gd
