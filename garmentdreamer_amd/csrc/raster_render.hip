// raster_render.hip -- per-tile alpha blending, forward pass, for gfx950.
//
//   render_forward_kernel   <- renderCUDA forward  (DGR/cuda_rasterizer/forward.cu:261-381)
//   (the reverse pass lives in raster_render_bwd.hip)
//
// Mapping to CDNA4.  A 16x16 tile is one 256-thread workgroup = 4 wave64s; wave w owns the 16x4 pixel strip of tile
// rows 4w .. 4w+3, and inside a strip each group of 16 lanes is one 4x4 pixel BLOCK (lane = 16*block + 4*y + x): the
// 16-bit quarters of a wave ballot are then per-block masks, which is what the reverse pass is organised around.
// Per round of 256 list entries the workgroup stages xy (8 B), conic+opacity (16 B) and colour+depth (16 B) of each
// entry in LDS once -- the reference re-gathers colour and depth from global memory per contributing pair
// (forward.cu:359-361).  All lanes then read the same LDS address (broadcast, conflict free).
//
// Hand-over to the reverse pass: per strip a COMPACT list of the entries that at least one of its pixels blended,
// {ballot of those pixels (64 bit), Gaussian id, instance slot}, appended in list order + the strip's entry count.
// The reverse pass walks exactly these (pixel, entry) pairs -- it repeats neither the contribution test nor the
// culling, and never touches the ~4/5 of (strip, list position) pairs nobody blended.
//
// Blocks are mapped to tiles so that each XCD (private 4 MiB L2) works on a contiguous band of tiles: neighbouring
// tiles share most of their Gaussian lists.
//
// Numerics: every blend operation is ONE separately rounded fp32 operation in the oracle's order (no FMA
// contraction, gd_expf defined operation by operation): colour / depth / alpha / n_contrib are bit-identical to
// oracle/gd_oracle.c.
#include <stdlib.h>

#include "raster_common.h"
#include "raster_blend_math.h"


namespace gd {

namespace {

__device__ __forceinline__ uint32_t block_to_tile(uint32_t b, uint32_t n)
{
    // workgroup b is observed to run on XCD b % 8 (placement is a speed hint only)
    if ((n & 7u) == 0) {
        const uint32_t per = n >> 3;
        return (b & 7u) * per + (b >> 3);
    }
    return b;
}

// power = -0.5 (a dx^2 + c dy^2) - b dx dy in EXACTLY the reference's evaluation order, every operation rounded
// on its own (forward.cu:341 / backward.cu:528 as the oracle compiles them, no FMA contraction): the forward
// blend, the backward blend and the oracle then agree on the bits of `power`, hence on which (pixel, Gaussian)
// pairs pass the 1/255 test -- a flipped pair is a discontinuous step in the gradients, amplified by 0.5 W in
// dL/dmean2D.  t1 = (a dx) dx and bdx = b dx may be hoisted by the caller.  (-0.5 s is exact, so the final fma
// equals the separately rounded subtraction.)
__device__ __forceinline__ float power_exact(const float t1, const float bdx, const float c, const float dy)
{
#pragma clang fp contract(off)   // HIP's __fmul_rn / __fadd_rn are plain operators: without this they may be fused
    const float t2 = mul_rn(mul_rn(c, dy), dy);
    const float s = add_rn(t1, t2);
    return fmaf(-0.5f, s, -mul_rn(bdx, dy));
}

// Which of the tile's four 16x4 pixel strips can an entry reach?  Bit s is CLEAR only if the minimum of the
// conic's quadratic form  f(d) = 0.5 (a dx^2 + c dy^2) + b dx dy,  d = centre - pixel,  over the strip's
// rectangle exceeds tau (= ln(255 opacity) + margin): then every pixel of the strip has power < ln(1/(255 o))
// and fails the alpha >= 1/255 test, so skipping the entry for that strip's wave changes no result.
// f is convex, so unless the centre lies inside the rectangle the minimum sits on one of the four edges,
// where it is a clamped 1-D parabola minimum.  (The reference walks every entry of the tile for every
// pixel, forward.cu:330-358 / backward.cu:505-533; ~56 % of (strip, entry) pairs are dead on the
// benchmark scene.)  One evaluation per staged entry, by the staging thread.
__device__ __forceinline__ uint32_t strip_alive_mask(const float2 xy, const float4 co, const float tau,
                                                     const float x0, const float y0)
{
#ifdef GD_NO_STRIP_CULL   // A/B builds only (tools/, tests never define it)
    return 0xFu;
#endif
    const float a = co.x, b = co.y, c = co.z;
    const float dxa = xy.x - (x0 + 15.0f), dxb = xy.x - x0;
    const float ra = __builtin_amdgcn_rcpf(a), rc = __builtin_amdgcn_rcpf(c);
    const bool xin = dxa <= 0.0f && dxb >= 0.0f;
    const float tA = -b * dxa * rc, tB = -b * dxb * rc;          // unconstrained dy* on the two vertical edges
    const float hA = 0.5f * a * dxa * dxa, hB = 0.5f * a * dxb * dxb;
    const float bA = b * dxa, bB = b * dxb;
    uint32_t mask = 0;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const float dya = xy.y - (y0 + 4.0f * s + 3.0f), dyb = xy.y - (y0 + 4.0f * s);
        const bool yin = dya <= 0.0f && dyb >= 0.0f;
        float t = fminf(fmaxf(tA, dya), dyb);
        const float v1 = hA + t * (bA + 0.5f * c * t);
        t = fminf(fmaxf(tB, dya), dyb);
        const float v2 = hB + t * (bB + 0.5f * c * t);
        t = fminf(fmaxf(-b * dya * ra, dxa), dxb);
        const float v3 = 0.5f * c * dya * dya + t * (b * dya + 0.5f * a * t);
        t = fminf(fmaxf(-b * dyb * ra, dxa), dxb);
        const float v4 = 0.5f * c * dyb * dyb + t * (b * dyb + 0.5f * a * t);
        const float fmin_ = (xin && yin) ? 0.0f : fminf(fminf(v1, v2), fminf(v3, v4));
        if (!(fmin_ > tau)) mask |= 1u << s;   // NaN -> alive
    }
    return mask;
}

__global__ __launch_bounds__(kTilePix) void render_forward_kernel(
    int W, int H, uint32_t gx, uint32_t gy, uint32_t tiles_total, const uint2* __restrict__ ranges,
    const uint32_t* __restrict__ point_list, const float2* __restrict__ means2D,
    const float4* __restrict__ conic_opacity, const float* __restrict__ alpha_thr, const float4* __restrict__ rgbd,
    const float* __restrict__ bg_color,
    float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ out_alpha,
    uint32_t* __restrict__ n_contrib, uint2* __restrict__ pair_counts, const uint32_t* __restrict__ slot_of,
    uint4* __restrict__ clist, uint32_t* __restrict__ strip_count, uint32_t* __restrict__ rowpos,
    const uint32_t* __restrict__ tile_perm)
{
    // every blend operation is rounded on its own, like the oracle's (-ffp-contract=off): images match it bit for bit
#pragma clang fp contract(off)
    __shared__ float2 s_xy[kTilePix];
    __shared__ float4 s_co[kTilePix];
    __shared__ float4 s_fd[kTilePix];
    __shared__ float s_thr[kTilePix];   // ln(1/(255 opacity)) - margin: below it alpha < 1/255 for certain
    __shared__ uint32_t s_mask[kTilePix];  // strip_alive_mask per staged entry (0 beyond the list end)
    __shared__ __attribute__((aligned(4))) uint8_t s_idx[4][68];   // per wave: compact list of the current 64 entries it must walk

    const uint32_t tile = tile_perm ? tile_perm[blockIdx.x] : block_to_tile(blockIdx.x, tiles_total);
    const uint32_t tpv = gx * gy;
    const uint32_t view = tile / tpv;
    const uint32_t lt = tile - view * tpv;
    const uint32_t ty = lt / gx, tx = lt - ty * gx;
    const uint32_t tid = threadIdx.x;
    // lane = 16 * block + 4 * y + x inside the wave's 16x4 strip (4x4 pixel blocks side by side)
    const uint32_t px = tx * kTile + ((tid >> 2) & 12u) + (tid & 3u), py = ty * kTile + ((tid >> 6) << 2) + ((tid >> 2) & 3u);
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const size_t HW = (size_t)H * W;
    const size_t pix_id = (size_t)W * py + px;
    const float pixf_x = (float)px, pixf_y = (float)py;
    const float __attribute__((ext_vector_type(2))) pixf = {pixf_x, pixf_y};

    bool done = !inside;
    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    const int rounds = (total + kTilePix - 1) / kTilePix;
    int toDo = total;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));   // wave w owns strip w (tile rows 4w .. 4w+3)
    const float tile_x0 = (float)(tx * kTile), tile_y0 = (float)(ty * kTile);

    float T = 1.0f;
    // contributor = entries walked while the pixel was live (forward.cu:336): the whole list unless the pixel
    // saturates at entry g, then g + 1
    uint32_t contributor = (uint32_t)total, last_contributor = 0, blended = 0;
    // two-wide arithmetic where the operands sit in register pairs anyway (xy, conic a|c, colour rg|bd): v_pk_mul_f32 /
    // v_pk_add_f32 round each half exactly like the scalar instruction at ~0.6 of the issue time (the kernel is
    // VALU-issue bound); contraction stays off, so no half is ever fused
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 C01 = {0.f, 0.f}, C2D = {0.f, 0.f};
    float weight = 0.f;
    // compact list of this wave's strip: region [4 range.x + wave * total, + total) of clist, filled in list order
    const uint32_t my_base = range.x * 4u + wave * (uint32_t)total;
    uint4* const my_list = clist + my_base;
    uint32_t n_listed = 0;

    for (int i = 0; i < rounds; i++, toDo -= kTilePix) {
        if (__syncthreads_count(done) == kTilePix) break;
        const uint32_t progress = (uint32_t)i * kTilePix + tid;
        uint32_t alive = 0;
        if (range.x + progress < range.y) {
            const uint32_t id = point_list[range.x + progress];
            const float2 xy = means2D[id];
            s_xy[tid] = xy;
            const float4 c4 = conic_opacity[id];
            s_co[tid] = make_float4(c4.x, c4.z, c4.y, c4.w);     // (a, c | b, opacity): a|c pairs with dx|dy
            const float thr = alpha_thr[id];          // alpha_threshold_exact(opacity), from preprocess_kernel
            s_thr[tid] = thr;
            s_fd[tid] = rgbd[id];
            alive = strip_alive_mask(xy, c4, -thr + 1e-2f + 1e-3f * fabsf(thr), tile_x0, tile_y0);
        }
        s_mask[tid] = alive;
        __syncthreads();
        const int n = min(kTilePix, toDo);
        const uint32_t cbase = (uint32_t)i * kTilePix;
        bool wave_live = __builtin_amdgcn_ballot_w64(!done) != 0;
        for (int c = 0; wave_live && c < n; c += 64) {
            // 64 entries' strip bits -> the COMPACT list of the entries this wave must look at, in list order: lane l with
            // its bit set writes l at its rank.  The walk below then needs no scalar bit arithmetic per entry (find-first,
            // clear-lowest, 64-bit compares: the kernel issued as many SALU as VALU instructions, and a CU issues one SALU
            // instruction per cycle for all its waves) -- one broadcast LDS read hands it four indices at a time; up to three
            // sentinel indices (64) pad the last group: they blend nothing (threshold +inf) and match no lane.
            const uint64_t bits = __builtin_amdgcn_ballot_w64(((s_mask[c + lane] >> wave) & 1u) != 0);
            const uint32_t cnt = (uint32_t)__builtin_popcountll(bits);
            if ((bits >> lane) & 1ull)
                s_idx[wave][__builtin_amdgcn_mbcnt_hi((uint32_t)(bits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bits, 0u))] = (uint8_t)lane;
            if (lane < 3u) s_idx[wave][cnt + lane] = 64;
            // lane l collects, for list entry c + l, the ballot of the strip's pixels that BLEND it: the backward pass
            // walks exactly these (pixel, entry) pairs (zero = nobody, also for the entries the strip culling or an early
            // exit never looked at)
            uint32_t wb_lo = 0, wb_hi = 0;
            for (uint32_t k = 0; k < cnt; k += 4u) {
                const uint32_t packed = *(const uint32_t*)&s_idx[wave][k];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t jl = (packed >> (8 * u)) & 0xffu;
                    const int j = c + (int)(jl & 63u);
                    bool blend = false;
                    if (!done) {
                        const float2 xy = s_xy[j];
                        const float4 co = s_co[j];                       // a, c, b, opacity
                        const float thr = jl < 64u ? s_thr[j] : INFINITY;
                        const f2 d = f2{xy.x, xy.y} - pixf;               // (dx, dy)
                        const f2 t12 = (f2{co.x, co.y} * d) * d;          // ((a dx) dx, (c dy) dy), forward.cu:341's order
                        const float power = fmaf(-0.5f, add_rn(t12.x, t12.y), -mul_rn(mul_rn(co.z, d.x), d.y));
                        // alpha >= 1/255  <=>  power >= thr (exact, alpha_threshold_exact): only contributing pairs pay for
                        // the exponential
                        if (!(power > 0.0f) && !(power < thr)) {
                            const float alpha = fminf(0.99f, mul_rn(co.w, gd_expf(power)));
                            // the oracle's operations, each rounded on its own (forward.cu:349-363 without contraction)
                            const float test_T = mul_rn(T, sub_rn(1.0f, alpha));
                            if (test_T < 0.0001f) {
                                done = true;
                                contributor = cbase + (uint32_t)j + 1u;
                            } else {
                                const float4 fd = s_fd[j];
                                C01 = C01 + ((f2{fd.x, fd.y} * alpha) * T);
                                C2D = C2D + ((f2{fd.z, fd.w} * alpha) * T);
                                weight = add_rn(weight, mul_rn(alpha, T));
                                T = test_T;
                                last_contributor = cbase + (uint32_t)j + 1u;
                                blended++;
                                blend = true;
                            }
                        }
                    }
                    const uint64_t bal = __builtin_amdgcn_ballot_w64(blend);
                    const bool mine = lane == jl;
                    wb_lo = mine ? (uint32_t)bal : wb_lo;
                    wb_hi = mine ? (uint32_t)(bal >> 32) : wb_hi;
                }
                if (__builtin_amdgcn_ballot_w64(!done) == 0) {      // every pixel of the strip has saturated
                    wave_live = false;
                    break;
                }
            }
            // append the entries somebody blended (ranked by v_mbcnt: list order is kept)
            const bool nz = (wb_lo | wb_hi) != 0u;
            const uint64_t nzm = __builtin_amdgcn_ballot_w64(nz);
            if (nz) {
                const uint32_t pos = range.x + cbase + (uint32_t)c + lane;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(nzm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nzm, 0u));
                const uint32_t slot = slot_of[pos];
                my_list[n_listed + rank] = make_uint4(wb_lo, wb_hi, point_list[pos], slot);
                rowpos[(size_t)slot * 4u + wave] = my_base + n_listed + rank + 1u;   // where the backward pass puts the row
            }
            n_listed += (uint32_t)__builtin_popcountll(nzm);
        }
    }
    if (lane == 0) strip_count[tile * 4u + wave] = n_listed;
    if (inside) {
        n_contrib[(size_t)view * HW + pix_id] = last_contributor;
        pair_counts[(size_t)view * HW + pix_id] = make_uint2(contributor, blended);
        float* oc = out_color + (size_t)view * 3 * HW;
        oc[0 * HW + pix_id] = add_rn(C01.x, mul_rn(T, bg_color[0]));
        oc[1 * HW + pix_id] = add_rn(C01.y, mul_rn(T, bg_color[1]));
        oc[2 * HW + pix_id] = add_rn(C2D.x, mul_rn(T, bg_color[2]));
        out_alpha[(size_t)view * HW + pix_id] = weight;
        out_depth[(size_t)view * HW + pix_id] = C2D.y;
    }
}

__global__ void blend_exp_kernel(const float* __restrict__ x, float* __restrict__ y, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = gd_expf(x[i]);
}

}  // namespace

void launch_blend_exp(hipStream_t s, const float* x, float* y, int n)
{
    if (n > 0) hipLaunchKernelGGL(blend_exp_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, y, n);
}

void launch_render_forward(hipStream_t s, int V, int W, int H, int tiles_x, int tiles_y, const uint2* ranges,
                           const uint32_t* point_list, const GeomState& g, const float* bg, float* out_color,
                           float* out_depth, float* out_alpha, uint32_t* n_contrib, uint2* pair_counts,
                           const uint32_t* slot_of, uint4* clist, uint32_t* strip_count, uint32_t* rowpos,
                           const uint32_t* tile_perm)
{
    const uint32_t tiles_total = (uint32_t)V * tiles_x * tiles_y;
    hipLaunchKernelGGL(render_forward_kernel, dim3(tiles_total), dim3(kTilePix), 0, s, W, H, (uint32_t)tiles_x,
                       (uint32_t)tiles_y, tiles_total, ranges, point_list, g.means2D, g.conic_opacity, g.alpha_thr, g.rgbd, bg,
                       out_color, out_depth, out_alpha, n_contrib, pair_counts, slot_of, clist, strip_count, rowpos, tile_perm);
}

}  // namespace gd
