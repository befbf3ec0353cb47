// raster_api.hip -- C-ABI entry points (include/gd_raster.h) and host orchestration.
//
// Orchestration follows CudaRasterizer::Rasterizer::forward / backward
// (DGR/cuda_rasterizer/rasterizer_impl.cu:197-339, 343-446) with these MI355X-side changes:
//   * every launch goes to the caller's stream (the reference uses the legacy default stream);
//   * scratch lives wherever the caller's allocator puts it (the reference hard-wires
//     torch::kCUDA device 0, rasterize_points.cu:73-77);
//   * V views of the same Gaussians can be rendered by ONE launch set (batched entry): at 512^2 a
//     single view is only 1024 workgroups -- half of what 256 CUs can hold -- and each view
//     would cost its own blocking read-back of num_rendered;
//   * backward writes every output element, so the caller never pre-zeroes ten tensors.
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <vector>

#include <atomic>

#include "raster_common.h"

namespace gd {

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, const char* detail = "")
{
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

#define GD_HIP(expr)                                                          \
    do {                                                                      \
        hipError_t _e = (expr);                                               \
        if (_e != hipSuccess) return fail(GD_ERR_HIP, #expr ": %s", hipGetErrorString(_e)); \
    } while (0)

template <typename T>
void obtain(char*& chunk, T*& ptr, size_t count, size_t alignment = 128)
{
    size_t off = (reinterpret_cast<uintptr_t>(chunk) + alignment - 1) & ~(alignment - 1);
    ptr = reinterpret_cast<T*>(off);
    chunk = reinterpret_cast<char*>(ptr + count);
}

// ---- optional per-kernel event timing (gd_raster_profile_*) ----
struct Profiler {
    std::mutex mu;
    bool on = false;
    struct Rec { int kid; hipEvent_t a, b; };
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;
    double total_ms[GD_K_COUNT] = {0};
    int64_t count[GD_K_COUNT] = {0};
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
};
Profiler g_prof;

struct ProfScope {
    hipStream_t s; int kid; hipEvent_t a = nullptr, b = nullptr; bool active = false;
    ProfScope(hipStream_t s_, int kid_) : s(s_), kid(kid_)
    {
        if (!g_prof.on) return;
        std::lock_guard<std::mutex> lk(g_prof.mu);
        a = g_prof.get(); b = g_prof.get();
        if (a && b) { active = true; (void)hipEventRecord(a, s); }
    }
    ~ProfScope()
    {
        if (!active) return;
        (void)hipEventRecord(b, s);
        std::lock_guard<std::mutex> lk(g_prof.mu);
        g_prof.pending.push_back({kid, a, b});
    }
};

int check_debug(hipStream_t s, int debug, const char* where)
{
    if (!debug) return GD_OK;
    hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "[HIP ERROR] after %s: %s", where, hipGetErrorString(e));
        return GD_ERR_HIP;
    }
    return GD_OK;
}

}  // namespace

GeomState carve_geom(char* chunk, size_t VP, size_t* used)
{
    char* p = chunk;
    GeomState g;
    obtain(p, g.clamped, VP * 3);
    obtain(p, g.radii, VP);
    obtain(p, g.means2D, VP);
    obtain(p, g.cov3D, VP * 6);
    obtain(p, g.conic_opacity, VP);
    obtain(p, g.rgbd, VP);
    obtain(p, g.tiles_touched, VP);
    obtain(p, g.point_offsets, VP);
    obtain(p, g.block_sums, (VP + kGaussBlock - 1) / kGaussBlock + 1);
    obtain(p, g.alpha_thr, VP);
    if (used) *used = (size_t)(p - chunk);
    return g;
}

ImageState carve_image(char* chunk, size_t tiles_total, size_t pixels_total, size_t* used)
{
    char* p = chunk;
    ImageState im;
    obtain(p, im.ranges, tiles_total);
    obtain(p, im.n_contrib, pixels_total);
    obtain(p, im.pair_counts, pixels_total);
    obtain(p, im.strip_count, tiles_total * 4);
    obtain(p, im.tile_perm, tiles_total);
    obtain(p, im.tile_cursor, tiles_total);
    obtain(p, im.bin_stats, 4);
    if (used) *used = (size_t)(p - chunk);
    return im;
}

BinningState carve_binning(char* chunk, size_t R, size_t* used)
{
    char* p = chunk;
    BinningState b;
    obtain(p, b.point_list, R);
    obtain(p, b.point_list_alt, R);
    obtain(p, b.slot_vp, R);
    obtain(p, b.keys, R);
    obtain(p, b.keys_alt, R);
    const size_t nblk = (R + kSortTile - 1) / kSortTile;
    obtain(p, b.sort_hist, ((size_t)1 << kSortHistBits) * (nblk + 1));
    obtain(p, b.clist, 4 * R);
    obtain(p, b.rowpos, R);
    if (used) *used = (size_t)(p - chunk);
    return b;
}

namespace {

struct Dims {
    int tiles_x, tiles_y;
    uint32_t tiles_total;
    size_t pixels_total;
};
Dims make_dims(int W, int H, int V)
{
    Dims d;
    d.tiles_x = (W + kTile - 1) / kTile;
    d.tiles_y = (H + kTile - 1) / kTile;
    d.tiles_total = (uint32_t)V * d.tiles_x * d.tiles_y;
    d.pixels_total = (size_t)V * W * H;
    return d;
}

ViewScalars make_views(int V, const float* tanx, const float* tany, int W, int H)
{
    ViewScalars vs;
    memset(&vs, 0, sizeof(vs));
    vs.V = V;
    for (int v = 0; v < V; v++) {
        vs.tan_fovx[v] = tanx[v];
        vs.tan_fovy[v] = tany[v];
        vs.focal_y[v] = H / (2.0f * tany[v]);  // rasterizer_impl.cu:223-224
        vs.focal_x[v] = W / (2.0f * tanx[v]);
    }
    return vs;
}

// 1: tile-bucketed binning wherever the longest list allows it (default); 0: radix sort always (gd_raster_force_binning)
std::atomic<int> g_binning_mode{[] { const char* e = getenv("GD_RASTER_BUCKETS"); return (!e || atoi(e) != 0) ? 1 : 0; }()};

int forward_impl(hipStream_t stream, int V, gd_alloc_fn geom_alloc, void* geom_user, gd_alloc_fn binning_alloc,
                 void* binning_user, gd_alloc_fn image_alloc, void* image_user, int P, int D, int M,
                 const float* background, int W, int H, const float* means3D, const float* shs,
                 const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                 const float* projmatrix, const float* cam_pos, const float* tanx, const float* tany,
                 int prefiltered, float* out_color, float* out_depth, float* out_alpha, int* radii, int debug,
                 uint32_t capacity = 0, uint32_t* count_dev = nullptr, bool force_radix = false)
{
    // capacity > 0: the sync-free form (gd_raster_forward_batched_capacity) -- the instance count never leaves the device
    g_err[0] = 0;
    const bool sync_free = capacity > 0;
    if (sync_free && !count_dev) return fail(GD_ERR_INVALID_ARG, "%s", "the sync-free form needs the 4-word device count buffer");
    if (sync_free && capacity >= (1u << 30)) return fail(GD_ERR_INVALID_ARG, "%s", "capacity exceeds 2^30 (32-bit strip-list offsets)");
    if (V < 1 || V > GD_MAX_VIEWS) return fail(GD_ERR_INVALID_ARG, "V must be in [1, %s]", "GD_MAX_VIEWS");
    if (P < 0 || W <= 0 || H <= 0) return fail(GD_ERR_INVALID_ARG, "%s", "P, width, height must be positive");
    if (!geom_alloc || !binning_alloc || !image_alloc) return fail(GD_ERR_INVALID_ARG, "%s", "allocator callbacks are required");
    if (!out_color || !out_depth || !out_alpha) return fail(GD_ERR_INVALID_ARG, "%s", "output images are required");
    const size_t HW = (size_t)W * H;
    if (P == 0) {
        // rasterize_points.cu:83: P == 0 leaves the zero-initialised outputs untouched
        GD_HIP(hipMemsetAsync(out_color, 0, sizeof(float) * 3 * HW * V, stream));
        GD_HIP(hipMemsetAsync(out_depth, 0, sizeof(float) * HW * V, stream));
        GD_HIP(hipMemsetAsync(out_alpha, 0, sizeof(float) * HW * V, stream));
        if (sync_free) GD_HIP(hipMemsetAsync(count_dev, 0, 4 * sizeof(uint32_t), stream));
        return 0;
    }
    if (!means3D || !opacities || !background || !viewmatrix || !projmatrix || !cam_pos)
        return fail(GD_ERR_INVALID_ARG, "%s", "means3D, opacities, background, viewmatrix, projmatrix, cam_pos are required");
    if (!shs && !colors_precomp)  // NUM_CHANNELS is 3, so this is the only way to have no colour
        return fail(GD_ERR_NON_RGB, "%s", "provide SHs or precomputed colours");
    if (!cov3D_precomp && (!scales || !rotations))
        return fail(GD_ERR_INVALID_ARG, "%s", "provide scales+rotations or a precomputed 3D covariance");

    const size_t VP = (size_t)V * P;
    const Dims dm = make_dims(W, H, V);
    const ViewScalars vs = make_views(V, tanx, tany, W, H);

    char* geom_chunk = geom_alloc(geom_user, gd_raster_geom_bytes(P, V));
    char* img_chunk = image_alloc(image_user, gd_raster_image_bytes(W, H, V));
    if (!geom_chunk || !img_chunk) return fail(GD_ERR_ALLOC, "%s", "scratch allocator returned NULL");
    GeomState geom = carve_geom(geom_chunk, VP, nullptr);
    ImageState img = carve_image(img_chunk, dm.tiles_total, dm.pixels_total, nullptr);
    if (radii == nullptr) radii = geom.radii;

    { ProfScope ps(stream, GD_K_PREPROCESS);
    launch_preprocess(stream, P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, cov3D_precomp,
                      colors_precomp, viewmatrix, projmatrix, cam_pos, W, H, vs, radii, geom, dm.tiles_x, dm.tiles_y,
                      prefiltered != 0); }
    if (int e = check_debug(stream, debug, "preprocess")) return e;
    const uint32_t nblk = (uint32_t)((VP + kGaussBlock - 1) / kGaussBlock);
    { ProfScope ps(stream, GD_K_SCAN); launch_scan_block_sums(stream, geom.block_sums, nblk, sync_free ? count_dev : nullptr, capacity); }
    if (int e = check_debug(stream, debug, "scan")) return e;

    // Binning, round 6: TILE-BUCKETED by default -- count the instances of every tile, prefix-sum (= `ranges`), scatter every
    // instance into its tile's bucket, one LDS sort per tile (raster_binning.hip) -- instead of the global radix sort of all
    // (tile | depth) keys (rasterizer_impl.cu:304-309): ~5 launches and two passes over the keys instead of ~17 and eleven.
    // Lists longer than kBucketMax do not fit the per-tile sort: the synchronising form sees the longest list in the same
    // read-back as num_rendered and takes the radix path; the sync-free form flags the call (count_dev[2] = 2, nothing binned)
    // and the caller repeats it with force_radix.  GD_RASTER_BUCKETS=0 / gd_raster_force_binning(0): radix path always (A/B, tests).
    bool buckets = g_binning_mode.load(std::memory_order_relaxed) != 0 && !force_radix;
    if (buckets) {
        ProfScope ps(stream, GD_K_RANGES);
        launch_tile_count(stream, (int)VP, P, radii, geom, img.ranges, dm.tiles_total, dm.tiles_x, dm.tiles_y);
        launch_tile_scan(stream, img.ranges, dm.tiles_total, img.tile_cursor, img.bin_stats, sync_free ? count_dev : nullptr);
    }
    if (int e = check_debug(stream, debug, "tile counts")) return e;

    uint32_t num_rendered = capacity;       // sync-free: every buffer, grid and layout below is sized for the capacity
    const uint32_t* n_dev = sync_free ? count_dev + 1 : nullptr;     // ... and the kernels read the live count here
    if (!sync_free) {
        // the one host sync of the forward pass (rasterizer_impl.cu:282)
        uint32_t longest = 0;
        GD_HIP(hipMemcpyAsync(&num_rendered, geom.block_sums + nblk, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        if (buckets) GD_HIP(hipMemcpyAsync(&longest, img.bin_stats + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        GD_HIP(hipStreamSynchronize(stream));
        // the compact per-strip lists index 4 * num_rendered cells with 32-bit arithmetic (raster_render.hip: my_base, rowpos)
        if (num_rendered >= (1u << 30)) return fail(GD_ERR_INVALID_ARG, "%s", "num_rendered exceeds 2^30 (32-bit strip-list offsets)");
        if (longest > kBucketMax) buckets = false;
    }

    char* bin_chunk = binning_alloc(binning_user, gd_raster_binning_bytes(num_rendered));
    if (!bin_chunk) return fail(GD_ERR_ALLOC, "%s", "binning allocator returned NULL");
    BinningState bin = carve_binning(bin_chunk, num_rendered, nullptr);

    if (buckets) {
        { ProfScope ps(stream, GD_K_DUPLICATE);
        launch_tile_scatter(stream, (int)VP, P, radii, geom, bin.keys_alt, bin.slot_vp, img.tile_cursor, dm.tiles_x, dm.tiles_y,
                            sync_free ? count_dev : nullptr); }
        GD_HIP(hipMemsetAsync(bin.rowpos, 0, sizeof(uint4) * (size_t)num_rendered, stream));
        if (int e = check_debug(stream, debug, "scatter")) return e;
        { ProfScope ps(stream, GD_K_SORT);
        if (num_rendered > 0)
            launch_tile_sort(stream, img.ranges, dm.tiles_total, bin.keys_alt, bin.keys, bin.point_list, bin.point_list_alt,
                             bin.slot_vp); }
        if (int e = check_debug(stream, debug, "tile sort")) return e;
    } else {
    const SortPlan plan = plan_sort(dm.tiles_total);
    const bool start_in_alt = (plan.passes & 1) != 0;
    { ProfScope ps(stream, GD_K_DUPLICATE);
    launch_duplicate(stream, (int)VP, P, radii, geom, start_in_alt ? bin.keys_alt : bin.keys,
                     start_in_alt ? bin.point_list_alt : bin.point_list, bin.slot_vp, nullptr, dm.tiles_x, dm.tiles_y,
                     sync_free ? count_dev : nullptr); }
    GD_HIP(hipMemsetAsync(bin.rowpos, 0, sizeof(uint4) * (size_t)num_rendered, stream));
    if (int e = check_debug(stream, debug, "duplicate")) return e;
    { ProfScope ps(stream, GD_K_SORT); launch_radix_sort(stream, bin, num_rendered, plan, start_in_alt, n_dev); }
    if (int e = check_debug(stream, debug, "sort")) return e;
    { ProfScope ps(stream, GD_K_RANGES); launch_tile_ranges(stream, bin.keys, num_rendered, img.ranges, dm.tiles_total, bin.point_list,
                                                          bin.slot_vp, bin.point_list_alt, n_dev); }
    if (int e = check_debug(stream, debug, "ranges")) return e;
    }
    // longest tile lists first (a counting sort of the tiles by list length, one small launch): the blend kernels' workgroups
    // last as long as their lists (0 ... 1700 entries on the benchmark scene, 45 % of the tiles empty), and in index order the
    // heavy tiles of the last view started last -- render_forward 0.364 -> 0.335 ms per 8-view launch (profiles/r05_lpt_ab.txt)
    static const bool lpt_on = [] { const char* e = getenv("GD_RASTER_LPT"); return !e || atoi(e) != 0; }();
    const uint32_t* tile_perm = nullptr;
    if (lpt_on && dm.tiles_total >= 512) {
        launch_tile_order(stream, img.ranges, dm.tiles_total, img.tile_perm);
        tile_perm = img.tile_perm;
    }
    { ProfScope ps(stream, GD_K_RENDER_FWD);
    launch_render_forward(stream, V, W, H, dm.tiles_x, dm.tiles_y, img.ranges, bin.point_list, geom, background,
                          out_color, out_depth, out_alpha, img.n_contrib, img.pair_counts, bin.point_list_alt /* slot_of */,
                          bin.clist, img.strip_count, reinterpret_cast<uint32_t*>(bin.rowpos), tile_perm); }
    if (int e = check_debug(stream, debug, "render")) return e;
    GD_HIP(hipGetLastError());
    return (int)num_rendered;
}

// The backward blend runs in the forward pass's longest-first order too, its four strips of a tile kept back to back on the
// tile's XCD (raster_render_bwd.hip): 0.222 -> 0.209 ms per 8-view launch at FETCH_SIZE 80 -> 99 MB.  (Dealing the strips of a
// tile round-robin over the XCDs gave the same time at 402 MB: they gather the same Gaussians.)  GD_RASTER_LPT_BWD=0: index order.
static bool bwd_lpt()
{
    static const bool on = [] {
        const char* a = getenv("GD_RASTER_LPT"); const char* b = getenv("GD_RASTER_LPT_BWD");
        return (!a || atoi(a) != 0) && (!b || atoi(b) != 0);
    }();
    return on;
}

int backward_impl(hipStream_t stream, int V, int P, int D, int M, int R, const float* background, int W, int H,
                  const float* means3D, const float* shs, const float* colors_precomp, const float* alphas,
                  const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                  const float* viewmatrix, const float* projmatrix, const float* campos, const float* tanx,
                  const float* tany, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
                  char* bwd_scratch, const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dalphas,
                  float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_ddepth,
                  float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int debug)
{
    g_err[0] = 0;
    if (V < 1 || V > GD_MAX_VIEWS) return fail(GD_ERR_INVALID_ARG, "V must be in [1, %s]", "GD_MAX_VIEWS");
    if (P == 0) return GD_OK;
    if (!geom_buffer || !binning_buffer || !image_buffer || !bwd_scratch)
        return fail(GD_ERR_INVALID_ARG, "%s", "geom/binning/image/backward scratch buffers are required");
    if (!dL_dpix || !dL_dpix_depth || !dL_dalphas || !alphas)
        return fail(GD_ERR_INVALID_ARG, "%s", "dL_dpix, dL_dpix_depth, dL_dalphas and alphas are required");
    if (!dL_dmean3D || !dL_dopacity) return fail(GD_ERR_INVALID_ARG, "%s", "dL_dmean3D and dL_dopacity are required");
    const size_t VP = (size_t)V * P;
    const Dims dm = make_dims(W, H, V);
    const ViewScalars vs = make_views(V, tanx, tany, W, H);
    GeomState geom = carve_geom(geom_buffer, VP, nullptr);
    BinningState bin = carve_binning(binning_buffer, (size_t)R, nullptr);
    ImageState img = carve_image(image_buffer, dm.tiles_total, dm.pixels_total, nullptr);
    if (radii == nullptr) radii = geom.radii;

    // backward scratch: one 10-float row per entry of the forward pass's compact strip lists (same indexing as
    // bin.clist, at most 4R), written once and never accumulated, and the rows of each (view, Gaussian) added up
    float* rows = nullptr;
    float* acc = nullptr;     // [VP][10] (instance_sum_kernel)
    {
        char* p = bwd_scratch;
        obtain(p, rows, (size_t)R * 40);
        obtain(p, acc, VP * 10);
    }
    if (R > 0) {
        { ProfScope ps(stream, GD_K_RENDER_BWD);
        launch_render_backward(stream, V, W, H, dm.tiles_x, dm.tiles_y, img.ranges, geom, background, alphas, dL_dpix,
                               dL_dpix_depth, dL_dalphas, rows, bin.clist, img.strip_count,
                               bwd_lpt() && dm.tiles_total >= 512 ? img.tile_perm : nullptr); }
    }
    if (int e = check_debug(stream, debug, "render backward")) return e;

    const float* cov3D = cov3D_precomp ? cov3D_precomp : geom.cov3D;
    const size_t cov_stride = cov3D_precomp ? 0 : (size_t)P * 6;
    { ProfScope ps(stream, GD_K_PREPROCESS_BWD);
    launch_preprocess_backward(stream, P, D, M, V, means3D, radii, shs, geom.clamped, scales, rotations,
                               scale_modifier, cov3D, cov_stride, viewmatrix, projmatrix, campos, vs, rows, bin.rowpos,
                               geom.point_offsets, geom.tiles_touched, acc, colors_precomp != nullptr, dL_dmean2D, dL_dconic,
                               dL_dopacity, dL_dcolor, dL_ddepth,
                               dL_dmean3D, dL_dcov3D, shs ? dL_dsh : nullptr, scales ? dL_dscale : nullptr,
                               scales ? dL_drot : nullptr, nullptr); }
    if (int e = check_debug(stream, debug, "preprocess backward")) return e;
    GD_HIP(hipGetLastError());
    return GD_OK;
}

}  // namespace

}  // namespace gd

using namespace gd;

extern "C" {

size_t gd_raster_geom_bytes(int P, int V)
{
    size_t used = 0;
    carve_geom(nullptr, (size_t)P * (size_t)(V < 1 ? 1 : V), &used);
    return used + 128;
}
size_t gd_raster_image_bytes(int width, int height, int V)
{
    size_t used = 0;
    const Dims d = make_dims(width, height, V < 1 ? 1 : V);
    carve_image(nullptr, d.tiles_total, d.pixels_total, &used);
    return used + 128;
}
size_t gd_raster_binning_bytes(int64_t R)
{
    size_t used = 0;
    carve_binning(nullptr, (size_t)(R < 0 ? 0 : R), &used);
    return used + 128;
}
size_t gd_raster_backward_scratch_bytes(int P, int V, int64_t R)
{
    const size_t r = (size_t)(R < 0 ? 0 : R);
    return r * 40 * sizeof(float) + (size_t)P * (size_t)(V < 1 ? 1 : V) * 10 * sizeof(float) + 512;
}

int gd_raster_forward(void* stream, gd_alloc_fn geom_alloc, void* geom_user, gd_alloc_fn binning_alloc,
                      void* binning_user, gd_alloc_fn image_alloc, void* image_user, int P, int D, int M,
                      const float* background, int width, int height, const float* means3D, const float* shs,
                      const float* colors_precomp, const float* opacities, const float* scales,
                      float scale_modifier, const float* rotations, const float* cov3D_precomp,
                      const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                      float tan_fovy, int prefiltered, float* out_color, float* out_depth, float* out_alpha,
                      int* radii, int debug)
{
    return forward_impl((hipStream_t)stream, 1, geom_alloc, geom_user, binning_alloc, binning_user, image_alloc,
                        image_user, P, D, M, background, width, height, means3D, shs, colors_precomp, opacities,
                        scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, &tan_fovx,
                        &tan_fovy, prefiltered, out_color, out_depth, out_alpha, radii, debug);
}

int gd_raster_forward_batched(void* stream, int V, gd_alloc_fn geom_alloc, void* geom_user,
                              gd_alloc_fn binning_alloc, void* binning_user, gd_alloc_fn image_alloc,
                              void* image_user, int P, int D, int M, const float* background, int width,
                              int height, const float* means3D, const float* shs, const float* colors_precomp,
                              const float* opacities, const float* scales, float scale_modifier,
                              const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                              const float* projmatrix, const float* cam_pos, const float* tan_fovx,
                              const float* tan_fovy, int prefiltered, float* out_color, float* out_depth,
                              float* out_alpha, int* radii, int debug)
{
    if (!tan_fovx || !tan_fovy) return fail(GD_ERR_INVALID_ARG, "%s", "tan_fovx / tan_fovy host arrays are required");
    return forward_impl((hipStream_t)stream, V, geom_alloc, geom_user, binning_alloc, binning_user, image_alloc,
                        image_user, P, D, M, background, width, height, means3D, shs, colors_precomp, opacities,
                        scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx,
                        tan_fovy, prefiltered, out_color, out_depth, out_alpha, radii, debug);
}

int gd_raster_forward_batched_capacity(void* stream, int V, gd_alloc_fn geom_alloc, void* geom_user,
                                       gd_alloc_fn binning_alloc, void* binning_user, gd_alloc_fn image_alloc,
                                       void* image_user, int P, int D, int M, const float* background, int width,
                                       int height, const float* means3D, const float* shs, const float* colors_precomp,
                                       const float* opacities, const float* scales, float scale_modifier,
                                       const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                                       const float* projmatrix, const float* cam_pos, const float* tan_fovx,
                                       const float* tan_fovy, int prefiltered, float* out_color, float* out_depth,
                                       float* out_alpha, int* radii, int debug, int64_t capacity, uint32_t* count_dev)
{
    if (!tan_fovx || !tan_fovy) return fail(GD_ERR_INVALID_ARG, "%s", "tan_fovx / tan_fovy host arrays are required");
    const bool force_radix = capacity < 0;        // -capacity: the same call on the radix-sort binning (see the header)
    if (force_radix) capacity = -capacity;
    if (capacity <= 0 || capacity >= (1ll << 30)) return fail(GD_ERR_INVALID_ARG, "%s", "|capacity| must be in [1, 2^30)");
    return forward_impl((hipStream_t)stream, V, geom_alloc, geom_user, binning_alloc, binning_user, image_alloc,
                        image_user, P, D, M, background, width, height, means3D, shs, colors_precomp, opacities,
                        scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx,
                        tan_fovy, prefiltered, out_color, out_depth, out_alpha, radii, debug, (uint32_t)capacity, count_dev,
                        force_radix);
}

int gd_raster_backward(void* stream, int P, int D, int M, int R, const float* background, int width, int height,
                       const float* means3D, const float* shs, const float* colors_precomp, const float* alphas,
                       const float* scales, float scale_modifier, const float* rotations,
                       const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                       const float* campos, float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer,
                       char* binning_buffer, char* image_buffer, char* bwd_scratch, const float* dL_dpix,
                       const float* dL_dpix_depth, const float* dL_dalphas, float* dL_dmean2D, float* dL_dconic,
                       float* dL_dopacity, float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D,
                       float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int debug)
{
    return backward_impl((hipStream_t)stream, 1, P, D, M, R, background, width, height, means3D, shs, colors_precomp,
                         alphas, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, campos,
                         &tan_fovx, &tan_fovy, radii, geom_buffer, binning_buffer, image_buffer, bwd_scratch, dL_dpix,
                         dL_dpix_depth, dL_dalphas, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_ddepth,
                         dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug);
}

int gd_raster_backward_batched(void* stream, int V, int P, int D, int M, int R, const float* background,
                               int width, int height, const float* means3D, const float* shs,
                               const float* colors_precomp, const float* alphas, const float* scales,
                               float scale_modifier, const float* rotations, const float* cov3D_precomp,
                               const float* viewmatrix, const float* projmatrix, const float* campos,
                               const float* tan_fovx, const float* tan_fovy, const int* radii, char* geom_buffer,
                               char* binning_buffer, char* image_buffer, char* bwd_scratch, const float* dL_dpix,
                               const float* dL_dpix_depth, const float* dL_dalphas, float* dL_dmean2D,
                               float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
                               float* dL_dsh, float* dL_dscale, float* dL_drot, int debug)
{
    if (!tan_fovx || !tan_fovy) return fail(GD_ERR_INVALID_ARG, "%s", "tan_fovx / tan_fovy host arrays are required");
    return backward_impl((hipStream_t)stream, V, P, D, M, R, background, width, height, means3D, shs, colors_precomp,
                         alphas, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, campos,
                         tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, image_buffer, bwd_scratch, dL_dpix,
                         dL_dpix_depth, dL_dalphas, dL_dmean2D, nullptr, dL_dopacity, dL_dcolor, nullptr, dL_dmean3D,
                         dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug);
}

int gd_raster_mark_visible(void* stream, int P, const float* means3D, const float* viewmatrix,
                           const float* projmatrix, uint8_t* present)
{
    (void)projmatrix;  // in_frustum only tests view-space z (auxiliary.h:153)
    g_err[0] = 0;
    if (P == 0) return GD_OK;
    if (!means3D || !viewmatrix || !present) return fail(GD_ERR_INVALID_ARG, "%s", "means3D, viewmatrix, present are required");
    launch_mark_visible((hipStream_t)stream, P, means3D, viewmatrix, present);
    GD_HIP(hipGetLastError());
    return GD_OK;
}

int gd_raster_get_layout(const char* geom_base, const char* image_base, const char* binning_base, int P, int V,
                         int width, int height, int64_t R, gd_raster_layout* out)
{
    if (!out) return GD_ERR_INVALID_ARG;
    const size_t VP = (size_t)P * V;
    const Dims d = make_dims(width, height, V);
    GeomState g = carve_geom(const_cast<char*>(geom_base), VP, nullptr);
    ImageState im = carve_image(const_cast<char*>(image_base), d.tiles_total, d.pixels_total, nullptr);
    BinningState b = carve_binning(const_cast<char*>(binning_base), (size_t)R, nullptr);
#define OFF(base, p) ((size_t)(reinterpret_cast<const char*>(p) - (base)))
    out->depths = OFF(geom_base, g.rgbd) + 12;  // depth is the .w of rgbd (stride 16 B)
    out->clamped = OFF(geom_base, g.clamped);
    out->radii = OFF(geom_base, g.radii);
    out->means2D = OFF(geom_base, g.means2D);
    out->cov3D = OFF(geom_base, g.cov3D);
    out->conic_opacity = OFF(geom_base, g.conic_opacity);
    out->rgb = OFF(geom_base, g.rgbd);
    out->tiles_touched = OFF(geom_base, g.tiles_touched);
    out->point_offsets = OFF(geom_base, g.point_offsets);
    out->block_sums = OFF(geom_base, g.block_sums);
    out->ranges = OFF(image_base, im.ranges);
    out->n_contrib = OFF(image_base, im.n_contrib);
    out->pair_counts = OFF(image_base, im.pair_counts);
    out->point_list = OFF(binning_base, b.point_list);
    out->point_list_alt = OFF(binning_base, b.point_list_alt);
    out->keys = OFF(binning_base, b.keys);
    out->keys_alt = OFF(binning_base, b.keys_alt);
    out->sort_hist = OFF(binning_base, b.sort_hist);
#undef OFF
    return GD_OK;
}

int gd_raster_blend_exp(void* stream, const float* x, float* y, int n)
{
    if (!x || !y || n < 0) return GD_ERR_INVALID_ARG;
    launch_blend_exp((hipStream_t)stream, x, y, n);
    return hipGetLastError() == hipSuccess ? GD_OK : GD_ERR_HIP;
}

int gd_raster_force_binning(int mode)
{
    if (mode < -1 || mode > 1) return GD_ERR_INVALID_ARG;
    if (mode < 0) {
        const char* e = getenv("GD_RASTER_BUCKETS");
        mode = (!e || atoi(e) != 0) ? 1 : 0;
    }
    g_binning_mode.store(mode, std::memory_order_relaxed);
    return GD_OK;
}

int gd_raster_poison_lds(void* stream)
{
    launch_poison_lds((hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? GD_OK : GD_ERR_HIP;
}

int gd_raster_sort_bits(int width, int height, int V)
{
    const Dims d = make_dims(width, height, V < 1 ? 1 : V);
    return plan_sort(d.tiles_total).total_bits;
}

int gd_raster_profile_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.on = on != 0;
    return GD_OK;
}

int gd_raster_profile_collect(void)
{
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (auto& r : g_prof.pending) {
        if (hipEventSynchronize(r.b) == hipSuccess) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
                g_prof.total_ms[r.kid] += ms;
                g_prof.count[r.kid] += 1;
            }
        }
        g_prof.pool.push_back(r.a);
        g_prof.pool.push_back(r.b);
    }
    g_prof.pending.clear();
    return GD_OK;
}

int gd_raster_profile_get(int kernel_id, double* total_ms, int64_t* launches)
{
    if (kernel_id < 0 || kernel_id >= GD_K_COUNT) return GD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (total_ms) *total_ms = g_prof.total_ms[kernel_id];
    if (launches) *launches = g_prof.count[kernel_id];
    return GD_OK;
}

int gd_raster_profile_reset(void)
{
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (int i = 0; i < GD_K_COUNT; i++) { g_prof.total_ms[i] = 0; g_prof.count[i] = 0; }
    return GD_OK;
}

const char* gd_raster_profile_kernel_name(int kernel_id)
{
    static const char* names[GD_K_COUNT] = {"preprocess_kernel", "scan_block_sums_kernel", "duplicate_kernel",
                                            "radix_sort(all passes)", "tile_ranges_kernel", "render_forward_kernel",
                                            "render_backward_block_kernel", "instance_sum_kernel + preprocess_backward_kernel"};
    return (kernel_id >= 0 && kernel_id < GD_K_COUNT) ? names[kernel_id] : "";
}

const char* gd_raster_last_error(void) { return g_err; }

const char* gd_raster_build_info(void)
{
    return "garmentdreamer_amd rasterizer: gfx950 (CDNA4), wave64, 16x16 tiles, HIP " __VERSION__;
}

}  // extern "C"
