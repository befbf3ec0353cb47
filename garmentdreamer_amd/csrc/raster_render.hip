// raster_render.hip -- per-tile alpha blending (forward) and its reverse-order gradient pass
// for gfx950.
//
//   render_forward_kernel   <- renderCUDA forward  (DGR/cuda_rasterizer/forward.cu:261-381)
//   render_backward_kernel  <- renderCUDA backward (DGR/cuda_rasterizer/backward.cu:415-601)
//
// Mapping to CDNA4.  A 16x16 tile is one 256-thread workgroup = 4 wave64s, each wave owning
// 4 rows x 16 columns of pixels.  Per round of 256 list entries the workgroup stages
// xy (8 B), conic+opacity (16 B) and colour+depth (16 B) of each entry in LDS once -- the
// reference re-gathers colour and depth from global memory per contributing pair
// (forward.cu:359-361).  All lanes then read the same LDS address (broadcast, conflict free).
//
// Backward: the reference issues 10 fp32 atomicAdd per contributing (pixel, Gaussian) pair
// (backward.cu:555-598).  Here the 10 partials are first summed over the 64 lanes of a wave
// with DPP row operations (no LDS traffic), the 4 waves of the tile combine through LDS
// atomics (ds_add_f32), and each workgroup then issues at most one global atomic per
// (tile, entry, component): a 256x cut of L2 atomic traffic.  Entries no pixel of the wave
// touches are skipped before the reduction, and list entries behind every pixel's last
// contributor are never visited at all.  The accumulation target is an interleaved
// acc[vp][10] row (40 B, one or two cache lines per Gaussian) instead of five separate arrays.
//
// Blocks are mapped to tiles so that each XCD (private 4 MiB L2) works on a contiguous band
// of tiles: neighbouring tiles share most of their Gaussian lists.
//
// Numerics: same operation order as the reference per pixel; FMA contraction is allowed here
// (pixel / gradient parity is a tolerance), expf is the accurate libm form so that
// the 1/255 and 1e-4 thresholds (and therefore n_contrib) agree with the oracle.
#include <stdlib.h>

#include "raster_common.h"

#ifndef GD_BWD_ROUND
#define GD_BWD_ROUND 128
#define GD_BWD_CAP 512
#endif
#ifndef GD_ABLATE
#define GD_ABLATE 0   // 1: skip the cross-lane reduction + LDS atomics (timing ablation only; wrong results)
#endif

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

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_term(float v)
{
    return __int_as_float(
        __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, true));
}

// Sum over the 64 lanes of a wave; the total is valid in lane 63.
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v += dpp_term<0xB1, 0xf, 0xf>(v);   // quad_perm [1,0,3,2]
    v += dpp_term<0x4E, 0xf, 0xf>(v);   // quad_perm [2,3,0,1]
    v += dpp_term<0x141, 0xf, 0xf>(v);  // row_half_mirror
    v += dpp_term<0x140, 0xf, 0xf>(v);  // row_mirror      -> every lane holds its row's sum
    v += dpp_term<0x142, 0xa, 0xf>(v);  // row_bcast:15 into rows 1,3
    v += dpp_term<0x143, 0xc, 0xf>(v);  // row_bcast:31 into rows 2,3 -> lane 63 = total
    return v;
}

typedef __attribute__((ext_vector_type(2))) unsigned gd_u2;
// v_permlane32_swap: a <- [a.lo | b.lo], b <- [a.hi | b.hi]   (lo/hi = lanes 0-31 / 32-63)
__device__ __forceinline__ void swap32(float& a, float& b)
{
    gd_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
// v_permlane16_swap: a <- [a.r0, b.r0, a.r2, b.r2], b <- [a.r1, b.r1, a.r3, b.r3]   (r = rows of 16 lanes)
__device__ __forceinline__ void swap16(float& a, float& b)
{
    gd_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

// Sum TEN values over the 64 lanes of a wave with a halving tree: each step adds lane partners AND
// halves the number of live registers (the two halves of the wave / row / octet / quad carry different
// values afterwards), so the whole reduction is ~38 VALU instead of 10 x 6 DPP adds.  On return the
// 4 lanes of quad (lane >> 2) all hold the total of value  4*(lane>>4) + ((lane&4) ? 2 + ((lane>>3)&1)
// : ((lane>>3)&1))  (values >= 10 are junk).
__device__ __forceinline__ float wave_reduce10(float (&v)[10], uint32_t lane)
{
    float s[8];
    // distance 32: (v[i], v[i+8]) -> lower half: value i, upper half: value i+8
    swap32(v[0], v[8]); s[0] = v[0] + v[8];
    swap32(v[1], v[9]); s[1] = v[1] + v[9];
#pragma unroll
    for (int i = 2; i < 8; i++) {   // partner value is identically zero: only the lower half is meaningful
        float c = v[i];
        swap32(v[i], c);
        s[i] = v[i] + c;
    }
    // distance 16: rows become values i, i+4, i+8, i+12
#pragma unroll
    for (int i = 0; i < 4; i++) {
        swap16(s[i], s[i + 4]);
        s[i] = s[i] + s[i + 4];
    }
    // distance 8 (row_ror:8), keep t0|t1 and t2|t3 in the two octets of a row
    const bool oct1 = (lane & 8u) != 0, quad1 = (lane & 4u) != 0;
    const float a0 = s[0] + dpp_term<0x128, 0xf, 0xf>(s[0]);
    const float a1 = s[1] + dpp_term<0x128, 0xf, 0xf>(s[1]);
    const float a2 = s[2] + dpp_term<0x128, 0xf, 0xf>(s[2]);
    const float a3 = s[3] + dpp_term<0x128, 0xf, 0xf>(s[3]);
    const float u0 = oct1 ? a1 : a0, u1 = oct1 ? a3 : a2;
    // distance 4: row_shl:4 serves lanes with (lane&4)==0, row_shr:4 the others
    const float p = u0 + dpp_term<0x104, 0xf, 0xf>(u0);
    const float q = u1 + dpp_term<0x114, 0xf, 0xf>(u1);
    float z = quad1 ? q : p;
    z += dpp_term<0x4E, 0xf, 0xf>(z);   // quad_perm [2,3,0,1]
    z += dpp_term<0xB1, 0xf, 0xf>(z);   // quad_perm [1,0,3,2]
    return z;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off, 64));
    return v;
}

// power = -0.5 (a dx^2 + c dy^2) - b dx dy in EXACTLY the reference's evaluation order, every operation rounded
// on its own (forward.cu:341 / backward.cu:528 as the oracle compiles them, no FMA contraction): the forward
// blend, the backward blend and the oracle then agree on the bits of `power`, hence on which (pixel, Gaussian)
// pairs pass the 1/255 test -- a flipped pair is a discontinuous step in the gradients, amplified by 0.5 W in
// dL/dmean2D.  t1 = (a dx) dx and bdx = b dx may be hoisted by the caller.  (-0.5 s is exact, so the final fma
// equals the separately rounded subtraction.)
__device__ __forceinline__ float power_exact(const float t1, const float bdx, const float c, const float dy)
{
    const float t2 = __fmul_rn(__fmul_rn(c, dy), dy);
    const float s = __fadd_rn(t1, t2);
    return fmaf(-0.5f, s, -__fmul_rn(bdx, dy));
}

// Smallest fp32 exponent p with  !(min(0.99, o * expf(p)) < 1/255): `power >= thr` is then the forward pass's
// contribution test itself (forward.cu:346-348), decided without evaluating the exponential per pair.
// A few accurate expf per STAGED entry (once per tile and entry).
__device__ __forceinline__ float alpha_threshold_exact(const float o)
{
    const float k = 1.0f / 255.0f;
    float t = -logf(255.0f * o);
    if (!(o > 0.0f) || !isfinite(t)) return INFINITY;     // opacity 0 (or NaN): nothing ever contributes
#pragma unroll 1
    for (int it = 0; it < 8; it++) {      // walk down while the next lower exponent still passes
        const float d = nextafterf(t, -INFINITY);
        if (fminf(0.99f, o * expf(d)) < k) break;
        t = d;
    }
#pragma unroll 1
    for (int it = 0; it < 16; it++) {     // walk up while this exponent fails
        if (!(fminf(0.99f, o * expf(t)) < k)) break;
        t = nextafterf(t, INFINITY);
    }
    return t;
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
    const float4* __restrict__ conic_opacity, const float4* __restrict__ rgbd, const float* __restrict__ bg_color,
    float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ out_alpha,
    uint32_t* __restrict__ n_contrib, uint2* __restrict__ pair_counts, uint64_t* __restrict__ ballots, uint32_t R)
{
    __shared__ float2 s_xy[kTilePix];
    __shared__ float4 s_co[kTilePix];
    __shared__ float4 s_fd[kTilePix];
    __shared__ float s_thr[kTilePix];   // ln(1/(255 opacity)) - margin: below it alpha < 1/255 for certain
    __shared__ uint32_t s_mask[kTilePix];  // strip_alive_mask per staged entry (0 beyond the list end)

    const uint32_t tile = block_to_tile(blockIdx.x, tiles_total);
    const uint32_t tpv = gx * gy;
    const uint32_t view = tile / tpv;
    const uint32_t lt = tile - view * tpv;
    const uint32_t ty = lt / gx, tx = lt - ty * gx;
    const uint32_t tid = threadIdx.x;
    const uint32_t px = tx * kTile + (tid & 15u), py = ty * kTile + (tid >> 4);
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const size_t HW = (size_t)H * W;
    const size_t pix_id = (size_t)W * py + px;
    const float pixf_x = (float)px, pixf_y = (float)py;

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
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, weight = 0.f, Dd = 0.f;

    for (int i = 0; i < rounds; i++, toDo -= kTilePix) {
        if (__syncthreads_count(done) == kTilePix) break;
        const uint32_t progress = (uint32_t)i * kTilePix + tid;
        uint32_t alive = 0;
        if (range.x + progress < range.y) {
            const uint32_t id = point_list[range.x + progress];
            const float2 xy = means2D[id];
            s_xy[tid] = xy;
            const float4 c4 = conic_opacity[id];
            s_co[tid] = c4;
            const float thr = alpha_threshold_exact(c4.w);
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
            // 64 entries' strip bits -> one scalar bitmap of the entries this wave must look at
            uint64_t bits = __builtin_amdgcn_ballot_w64(((s_mask[c + lane] >> wave) & 1u) != 0);
            // lane l collects, for list entry c + l, the ballot of the strip's pixels that BLEND it: the backward pass
            // walks exactly these (pixel, entry) pairs (ballots[strip][list position]; zero = nobody, also for the
            // entries the strip culling or an early exit never looked at)
            uint32_t wb_lo = 0, wb_hi = 0;
            while (bits) {
                const int jl = (int)__builtin_ctzll(bits);
                const int j = c + jl;
                bits &= bits - 1;
                bool blend = false;
                if (!done) {
                    const float2 xy = s_xy[j];
                    const float dx = xy.x - pixf_x, dy = xy.y - pixf_y;
                    const float4 co = s_co[j];
                    const float power = power_exact(__fmul_rn(__fmul_rn(co.x, dx), dx), __fmul_rn(co.y, dx), co.z, dy);
                    // alpha >= 1/255  <=>  power >= s_thr (exact, alpha_threshold_exact): only contributing pairs pay
                    // for the exponential
                    if (!(power > 0.0f) && !(power < s_thr[j])) {
                        const float alpha = fminf(0.99f, co.w * expf(power));
                        {
                            const float test_T = T * (1 - alpha);
                            if (test_T < 0.0001f) {
                                done = true;
                                contributor = cbase + (uint32_t)j + 1u;
                            } else {
                                const float4 fd = s_fd[j];
                                const float w = alpha * T;
                                C0 += fd.x * alpha * T;
                                C1 += fd.y * alpha * T;
                                C2 += fd.z * alpha * T;
                                weight += w;
                                Dd += fd.w * alpha * T;
                                T = test_T;
                                last_contributor = cbase + (uint32_t)j + 1u;
                                blended++;
                                blend = true;
                            }
                        }
                    }
                }
                const uint64_t bal = __builtin_amdgcn_ballot_w64(blend);
                const bool mine = (int)lane == jl;
                wb_lo = mine ? (uint32_t)bal : wb_lo;
                wb_hi = mine ? (uint32_t)(bal >> 32) : wb_hi;
                if (__builtin_amdgcn_ballot_w64(!done) == 0) {
                    wave_live = false;
                    break;
                }
            }
            if (c + (int)lane < n)
                ballots[(size_t)wave * R + (range.x + cbase + (uint32_t)c + lane)] = ((uint64_t)wb_hi << 32) | wb_lo;
        }
    }
    if (inside) {
        n_contrib[(size_t)view * HW + pix_id] = last_contributor;
        pair_counts[(size_t)view * HW + pix_id] = make_uint2(contributor, blended);
        float* oc = out_color + (size_t)view * 3 * HW;
        oc[0 * HW + pix_id] = C0 + T * bg_color[0];
        oc[1 * HW + pix_id] = C1 + T * bg_color[1];
        oc[2 * HW + pix_id] = C2 + T * bg_color[2];
        out_alpha[(size_t)view * HW + pix_id] = weight;
        out_depth[(size_t)view * HW + pix_id] = Dd;
    }
}

constexpr int kAcc = 10;  // colour rgb, depth, mean2D xy, conic x/y/w, opacity

// The backward blend kernels write ONE row of kAcc floats per list position (inst[position][kAcc], position =
// range.x ... range.y - 1 of the tile) and never accumulate across workgroups: every row of every tile is stored
// exactly once -- zeros for the entries no pixel reaches -- and preprocess_backward_kernel gathers the rows of a
// Gaussian.  zero_rows: rows [first, first + count) <- 0 by the whole workgroup.
__device__ __forceinline__ void zero_rows(float* __restrict__ inst, uint32_t first, uint32_t count, uint32_t tid,
                                          uint32_t threads)
{
    float2* p = reinterpret_cast<float2*>(inst + (size_t)first * kAcc);
    for (uint32_t i = tid; i < count * (kAcc / 2); i += threads) p[i] = make_float2(0.f, 0.f);
}

// Per-pixel state of the reverse walk (backward.cu:461-487).
struct PixState {
    float T, T_final, last_alpha, last_c0, last_c1, last_c2, last_depth;
    float accum_rec0, accum_rec1, accum_rec2, accum_depth_rec, accum_alpha_rec;
    float dLp0, dLp1, dLp2, dLpd, dLa, bg_dot;
    float pixf_y;
    uint32_t last_contributor;
};

// PPL = pixels per lane.  A 16x16 tile is handled by 256/PPL threads (4/PPL wave64s); lane l owns
// column l&15 and rows (l>>4) + 4*k' ... so that the per-entry cross-lane reduction (the dominant
// cost at PPL = 1: ablating it halves the kernel time) is paid once per PPL pixels: the lane first
// sums its own pixels' partials in registers.  PPL = 4 -> one wave per tile, no cross-wave
// combine; used when the launch has enough tiles to fill the chip (batched multi-view).
template <int PPL>
__global__ __launch_bounds__(kTilePix / PPL) void render_backward_kernel(
    int W, int H, uint32_t gx, uint32_t gy, uint32_t tiles_total, const uint2* __restrict__ ranges,
    const uint32_t* __restrict__ point_list, const float2* __restrict__ means2D,
    const float4* __restrict__ conic_opacity, const float4* __restrict__ rgbd, const float* __restrict__ bg_color,
    const float* __restrict__ alphas, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
    const float* __restrict__ dL_dpixel_depths, const float* __restrict__ dL_dalphas, float* __restrict__ acc)
{
    constexpr int THREADS = kTilePix / PPL;
    constexpr int WAVES = THREADS / 64;
    constexpr int ROWS_PER_PASS = THREADS / 16;   // tile rows covered by one pixel slot of all threads
    constexpr int ROUND = THREADS;                // list entries staged per round (keeps LDS per wave constant)
    __shared__ float2 s_xy[ROUND];
    __shared__ float4 s_co[ROUND];
    __shared__ float4 s_fd[ROUND];
    __shared__ uint32_t s_id[ROUND];
    __shared__ float s_thr[ROUND];                // ln(1 / (255 opacity)): alpha >= 1/255  <=>  power >= s_thr
    __shared__ float s_acc[ROUND * kAcc];
    __shared__ uint32_t s_mask[ROUND];            // strip_alive_mask per staged entry (0 beyond the list end)
    __shared__ uint32_t s_wmax[4];

    const uint32_t tile = block_to_tile(blockIdx.x, tiles_total);
    const uint32_t tpv = gx * gy;
    const uint32_t view = tile / tpv;
    const uint32_t lt = tile - view * tpv;
    const uint32_t ty = lt / gx, tx = lt - ty * gx;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t px = tx * kTile + (tid & 15u);
    const float pixf_x = (float)px;
    const size_t HW = (size_t)H * W;

    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
    const float bg0 = bg_color[0], bg1 = bg_color[1], bg2 = bg_color[2];

    const float tile_x0 = (float)(tx * kTile), tile_y0 = (float)(ty * kTile);
    uint32_t wave_strips = 0;                     // slot q of this wave covers strip wave + q * WAVES
#pragma unroll
    for (int q = 0; q < PPL; q++) wave_strips |= 1u << (wave + q * WAVES);

    PixState ps[PPL];
    uint32_t lane_max = 0;
#pragma unroll
    for (int q = 0; q < PPL; q++) {
        const uint32_t py = ty * kTile + (tid >> 4) + q * ROWS_PER_PASS;
        const bool inside = px < (uint32_t)W && py < (uint32_t)H;
        const size_t pix_id = (size_t)view * HW + (size_t)W * py + px;
        PixState& p = ps[q];
        p.pixf_y = (float)py;
        p.T_final = inside ? (1 - alphas[pix_id]) : 0;
        p.T = p.T_final;
        p.last_contributor = inside ? n_contrib[pix_id] : 0;
        p.dLp0 = p.dLp1 = p.dLp2 = p.dLpd = p.dLa = 0;
        if (inside) {
            const float* dp = dL_dpixels + (size_t)view * 3 * HW + ((size_t)W * py + px);
            p.dLp0 = dp[0]; p.dLp1 = dp[HW]; p.dLp2 = dp[2 * HW];
            p.dLpd = dL_dpixel_depths[pix_id];
            p.dLa = dL_dalphas[pix_id];
        }
        p.bg_dot = bg0 * p.dLp0 + bg1 * p.dLp1 + bg2 * p.dLp2;
        p.last_alpha = p.last_c0 = p.last_c1 = p.last_c2 = p.last_depth = 0;
        p.accum_rec0 = p.accum_rec1 = p.accum_rec2 = p.accum_depth_rec = p.accum_alpha_rec = 0;
        lane_max = max(lane_max, p.last_contributor);
    }

    // Entries whose ordinal is >= every pixel's last contributor are dead for the whole wave /
    // workgroup: skip them (the reference walks them and `continue`s per pixel, backward.cu:517-519).
    const uint32_t wmax = wave_max_u32(lane_max);
    if (lane == 0) s_wmax[wave] = wmax;
    for (int i = tid; i < ROUND * kAcc; i += THREADS) s_acc[i] = 0.f;
    __syncthreads();
    uint32_t bmax = s_wmax[0];
#pragma unroll
    for (int w = 1; w < WAVES; w++) bmax = max(bmax, s_wmax[w]);
    const int first_p_block = total - (int)bmax;  // first reverse position any pixel uses
    const int first_p_wave = total - (int)wmax;
    const int rounds = (total + ROUND - 1) / ROUND;
    {   // rows of the list entries behind every pixel's last contributor (whole skipped rounds): zeros
        const uint32_t skipped = bmax == 0 ? (uint32_t)total : (uint32_t)((first_p_block / ROUND) * ROUND);
        zero_rows(acc, range.y - skipped, skipped, tid, THREADS);
        if (bmax == 0) return;
    }

    for (int i = first_p_block / ROUND; i < rounds; i++) {
        const int round_base = i * ROUND;
        const int n = min(ROUND, total - round_base);
        __syncthreads();  // previous round's flush is complete
        {
            const int e = (int)tid;                   // ROUND == THREADS: one entry per thread
            uint32_t alive = 0;
            if (e < n) {
                const uint32_t id = point_list[range.y - (uint32_t)(round_base + e) - 1];
                s_id[e] = id;
                const float2 xy = means2D[id];
                s_xy[e] = xy;
                const float4 c4 = conic_opacity[id];
                s_co[e] = c4;
                const float thr = alpha_threshold_exact(c4.w);
                s_thr[e] = thr;
                s_fd[e] = rgbd[id];
                alive = strip_alive_mask(xy, c4, -thr + 1e-2f + 1e-3f * fabsf(thr), tile_x0, tile_y0);
            }
            s_mask[e] = alive;
        }
        __syncthreads();
        if (wmax != 0) {
            const int jstart = max(0, first_p_wave - round_base);
            for (int c = jstart & ~63; c < n; c += 64) {
              // 64 entries' strip bits -> scalar bitmap of the entries that can reach one of this wave's strips
              const uint32_t mv = s_mask[c + lane];
              uint64_t bits = __builtin_amdgcn_ballot_w64((mv & wave_strips) != 0);
              if (c < jstart) bits &= ~0ull << (jstart - c);
              while (bits) {
                const int jl = (int)__builtin_ctzll(bits);
                bits &= bits - 1;
                const int j = c + jl;
                const uint32_t mj = PPL > 1 ? (uint32_t)__builtin_amdgcn_readlane((int)mv, jl) : 0u;
                const uint32_t ordinal = (uint32_t)(total - 1 - (round_base + j));
                const float2 xy = s_xy[j];
                const float4 co = s_co[j];
                const float thr = s_thr[j];
                const float dx = xy.x - pixf_x;
                // power in the reference's expression order (backward.cu:528, as the forward pass): for large splats
                // a dx^2, c dy^2 and b dx dy reach 1e3..1e4 and cancel to O(1), so a re-associated (Horner) form
                // differs from the forward pass's G by up to 1e-3 relative.  a dx^2 and b dx are per-lane constants.
                const float pa = __fmul_rn(__fmul_rn(co.x, dx), dx), pb = __fmul_rn(co.y, dx);
                float v[kAcc];
#pragma unroll
                for (int k = 0; k < kAcc; k++) v[k] = 0.f;
                bool any_valid = false;
#pragma unroll
                for (int q = 0; q < PPL; q++) {
                    if (PPL > 1 && !((mj >> (wave + q * WAVES)) & 1u)) continue;   // strip of slot q cannot be reached
                    PixState& p = ps[q];
                    const float dy = xy.y - p.pixf_y;
                    const float power = power_exact(pa, pb, co.z, dy);
                    // alpha >= 1/255 decided on the exponent, with the exact per-entry threshold: the same pairs as
                    // the forward pass and the oracle, and the exp is only paid by contributing pairs
                    const bool valid = (ordinal < p.last_contributor) && !(power > 0.0f) && (power >= thr);
                    if (valid) {
                        any_valid = true;
                        // v_exp_f32 path (relative error ~3e-7); the forward pass keeps libm expf for n_contrib
                        const float G = __expf(power);
                        const float alpha = fminf(0.99f, co.w * G);
                        const float inv = __builtin_amdgcn_rcpf(1.f - alpha);  // shared by T/(1-a), T_final/(1-a)
                        p.T = p.T * inv;
                        const float dchannel_dcolor = alpha * p.T;
                        const float4 fd = s_fd[j];
                        float dL_dopa = 0.0f;
                        p.accum_rec0 = p.last_alpha * p.last_c0 + (1.f - p.last_alpha) * p.accum_rec0;
                        p.accum_rec1 = p.last_alpha * p.last_c1 + (1.f - p.last_alpha) * p.accum_rec1;
                        p.accum_rec2 = p.last_alpha * p.last_c2 + (1.f - p.last_alpha) * p.accum_rec2;
                        p.last_c0 = fd.x; p.last_c1 = fd.y; p.last_c2 = fd.z;
                        dL_dopa += (fd.x - p.accum_rec0) * p.dLp0;
                        dL_dopa += (fd.y - p.accum_rec1) * p.dLp1;
                        dL_dopa += (fd.z - p.accum_rec2) * p.dLp2;
                        v[0] += dchannel_dcolor * p.dLp0;
                        v[1] += dchannel_dcolor * p.dLp1;
                        v[2] += dchannel_dcolor * p.dLp2;
                        p.accum_depth_rec = p.last_alpha * p.last_depth + (1.f - p.last_alpha) * p.accum_depth_rec;
                        p.last_depth = fd.w;
                        dL_dopa += (fd.w - p.accum_depth_rec) * p.dLpd;
                        v[3] += dchannel_dcolor * p.dLpd;
                        p.accum_alpha_rec = p.last_alpha + (1.f - p.last_alpha) * p.accum_alpha_rec;
                        dL_dopa += (1 - p.accum_alpha_rec) * p.dLa;
                        dL_dopa *= p.T;
                        p.last_alpha = alpha;
                        dL_dopa += (-p.T_final * inv) * p.bg_dot;
                        const float dL_dG = co.w * dL_dopa;
                        const float gdx = G * dx, gdy = G * dy;
                        const float dG_ddelx = -gdx * co.x - gdy * co.y;
                        const float dG_ddely = -gdy * co.z - gdx * co.y;
                        v[4] += dL_dG * dG_ddelx * ddelx_dx;
                        v[5] += dL_dG * dG_ddely * ddely_dy;
                        v[6] += -0.5f * gdx * dx * dL_dG;
                        v[7] += -0.5f * gdx * dy * dL_dG;
                        v[8] += -0.5f * gdy * dy * dL_dG;
                        v[9] += G * dL_dopa;
                    }
                }
                if (!__any(any_valid)) continue;
#if GD_ABLATE == 1
#pragma unroll
                for (int k = 0; k < kAcc; k++) asm volatile("" ::"v"(v[k]));
#else
                const float z = wave_reduce10(v, lane);
                const uint32_t vid = 4u * (lane >> 4) + ((lane & 4u) ? 2u + ((lane >> 3) & 1u) : ((lane >> 3) & 1u));
                if ((lane & 3u) == 0 && vid < (uint32_t)kAcc) {
                    if (WAVES == 1) s_acc[j * kAcc + vid] = z;   // single writer per tile
                    else atomicAdd(&s_acc[j * kAcc + vid], z);
                }
#endif
              }
            }
        }
        __syncthreads();
        // flush this round: the row of list position range.y - 1 - (round_base + e) (no atomics: one writer per row)
        for (int e = tid; e < n; e += THREADS) {
            float2* dst = reinterpret_cast<float2*>(acc + (size_t)(range.y - 1u - (uint32_t)(round_base + e)) * kAcc);
#pragma unroll
            for (int k = 0; k < kAcc; k += 2) {
                float2* src = reinterpret_cast<float2*>(&s_acc[e * kAcc + k]);
                dst[k / 2] = *src;
                *src = make_float2(0.f, 0.f);
            }
        }
    }
}


// =================================================================================================================
// render_backward_lists_kernel -- the reverse-order gradient pass re-organised around PER-PIXEL LISTS.
//
// The wave-uniform walk above executes its ~60-instruction body for all 64 lanes of a strip although on the
// benchmark scene only a third of the 64 pixels blend a given (strip, entry) pair -- and for half of the pairs the
// strip culling lets through, none does -- and then pays a 38-instruction cross-lane reduction per pair: VALU-issue
// bound at 7 % of the fp32 roof.  Here nothing is evaluated for a pair that does not contribute:
//
//   *  the forward pass leaves, per (strip, list position), the 64-bit ballot of the pixels that blended the entry
//      (render_forward_kernel, `ballots`): exactly the pairs backward.cu:517-533 lets through, so the reverse pass
//      needs neither the contribution test nor the strip culling;
//   A  per group of 64 list entries a wave loads its 64 ballots (lane = entry), turns their popcounts into record
//      offsets with one DPP scan and transposes the non-zero ones into a per-PIXEL list (lane = pixel, bit = entry);
//   B  (lane = pixel, each lane walks ITS OWN list)  the sequential part of backward.cu:517-578 -- exp, alpha,
//      T /= (1 - alpha), the accumulated-colour recurrence, dL/dalpha.  Lanes advance independently, so a step keeps
//      about half of the lanes busy instead of a third, and the five recurrences of the reference (3 colours, depth,
//      alpha) collapse into ONE because only their dot product with the pixel's (dL/dC, dL/ddepth, dL/dalpha) is ever
//      used.  Output: a record {alpha T, G dL/dalpha, pixel} per contributing pair at slot base[entry] +
//      rank-of-the-pixel-in-the-entry's-ballot (v_mbcnt), i.e. grouped by entry;
//   C  (lane = quarter of an entry's records)  gathers the records and accumulates the ten sums of
//      backward.cu:555-598 in registers -- colour / depth gradients and the moments sum w, sum w d, sum w d d^T of
//      the pixel offsets d, from which dL/dmean2D, dL/dconic and dL/dopacity follow by one multiplication per entry.
//      No cross-lane reduction; LDS atomics combine the quarters and the tile's four strips; the flush to
//      acc[vp][10] is unchanged.
// =================================================================================================================
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v)
{
    // Hillis-Steele inside each row of 16 lanes (row_shr with zero fill), then the row totals (row_bcast 15 / 31)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);   // row_bcast:31 -> rows 2, 3
    return v;
}

template <int ROUND, int CAP>
__global__ __launch_bounds__(kTilePix) void render_backward_lists_kernel(
    int W, int H, uint32_t gx, uint32_t gy, uint32_t tiles_total, const uint2* __restrict__ ranges,
    const uint32_t* __restrict__ point_list, const float2* __restrict__ means2D,
    const float4* __restrict__ conic_opacity, const float4* __restrict__ rgbd, const float* __restrict__ bg_color,
    const float* __restrict__ alphas, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
    const float* __restrict__ dL_dpixel_depths, const float* __restrict__ dL_dalphas, float* __restrict__ acc,
    const uint64_t* __restrict__ ballots, uint32_t R, int ablate)
{
    static_assert(ROUND % 64 == 0 && CAP >= 64, "round = whole 64-entry groups; one entry has up to 64 records");
    constexpr int THREADS = kTilePix;
    __shared__ float2 s_xy[ROUND];
    __shared__ float4 s_co[ROUND];
    __shared__ float4 s_fd[ROUND];
    __shared__ uint32_t s_id[ROUND];
    __shared__ float s_acc[4][ROUND * kAcc];   // PER WAVE (strip): every (strip, entry) row has exactly one writer -> plain stores
    __shared__ uint4 s_tab[4][64];          // per wave, per entry of the current group: {ballot lo, hi, record base, count}
    __shared__ uint8_t s_nz[4][64];         // per wave: the entries of the group that have records, compacted
    __shared__ float2 s_rec[4][CAP];        // per wave: {alpha T, G dL/dalpha} per contributing pair, grouped by entry
    __shared__ uint8_t s_rid[4][CAP];       //           ... and the pixel (lane) it belongs to
    __shared__ float4 s_pix[kTilePix];      // per pixel: dL/dC rgb, dL/ddepth
    __shared__ uint32_t s_wmax[4];

    const uint32_t tile = block_to_tile(blockIdx.x, tiles_total);
    const uint32_t tpv = gx * gy;
    const uint32_t view = tile / tpv;
    const uint32_t lt = tile - view * tpv;
    const uint32_t ty = lt / gx, tx = lt - ty * gx;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t px = tx * kTile + (tid & 15u), py = ty * kTile + (tid >> 4);
    const float pixf_x = (float)px, pixf_y = (float)py;
    const size_t HW = (size_t)H * W;
    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    const float tile_x0 = (float)(tx * kTile), tile_y0 = (float)(ty * kTile);
    const uint64_t* my_ballots = ballots + (size_t)wave * R;

    // ---- per-pixel constants and the state of the reverse walk (backward.cu:461-487) ----
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const size_t pix_id = (size_t)view * HW + (size_t)W * py + px;
    const float T_final = inside ? (1 - alphas[pix_id]) : 0;
    float T = T_final;
    const uint32_t last_contributor = inside ? n_contrib[pix_id] : 0;
    float dLp0 = 0, dLp1 = 0, dLp2 = 0, dLpd = 0, dLa = 0;
    if (inside) {
        const float* dp = dL_dpixels + (size_t)view * 3 * HW + ((size_t)W * py + px);
        dLp0 = dp[0]; dLp1 = dp[HW]; dLp2 = dp[2 * HW];
        dLpd = dL_dpixel_depths[pix_id];
        dLa = dL_dalphas[pix_id];
    }
    const float bgT = T_final * (bg_color[0] * dLp0 + bg_color[1] * dLp1 + bg_color[2] * dLp2);
    s_pix[tid] = make_float4(dLp0, dLp1, dLp2, dLpd);
    float A = 0.f, last_alpha = 0.f, last_s = 0.f;   // A = sum_k accum_rec_k dL_k of the reference's five recurrences

    const uint32_t wmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(last_contributor));
    if (lane == 0) s_wmax[wave] = wmax;
    for (int i = tid; i < 4 * ROUND * kAcc; i += THREADS) (&s_acc[0][0])[i] = 0.f;
    __syncthreads();
    const uint32_t bmax = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
    const int first_p_block = total - (int)bmax;  // first reverse position any pixel uses
    const int first_p_wave = total - (int)wmax;
    const int rounds = (total + ROUND - 1) / ROUND;
    {   // rows of the list entries behind every pixel's last contributor (whole skipped rounds): zeros
        const uint32_t skipped = bmax == 0 ? (uint32_t)total : (uint32_t)((first_p_block / ROUND) * ROUND);
        zero_rows(acc, range.y - skipped, skipped, tid, THREADS);
        if (bmax == 0) return;
    }
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

    for (int i = first_p_block / ROUND; i < rounds; i++) {
        const int round_base = i * ROUND;
        const int n = min(ROUND, total - round_base);
        __syncthreads();  // previous round's flush is complete
        for (int e = (int)tid; e < n; e += THREADS) {
            const uint32_t id = point_list[range.y - (uint32_t)(round_base + e) - 1];
            s_id[e] = id;
            s_xy[e] = means2D[id];
            s_co[e] = conic_opacity[id];
            s_fd[e] = rgbd[id];
        }
        __syncthreads();
        const int jstart = max(0, first_p_wave - round_base);
        for (int c = jstart & ~63; wmax != 0 && c < n; c += 64) {
            // ---------------- pass A: ballots -> record offsets and per-pixel lists ----------------
            const int e_mine = c + (int)lane;
            uint64_t bal = 0;
            if (e_mine < n && e_mine >= jstart) bal = my_ballots[range.y - 1u - (uint32_t)(round_base + e_mine)];
            const uint32_t bal_lo = (uint32_t)bal, bal_hi = (uint32_t)(bal >> 32);
            const uint32_t cnt = (uint32_t)__builtin_popcountll(bal);
            const uint64_t nz = __builtin_amdgcn_ballot_w64(cnt != 0);
            if (nz == 0) continue;
            const uint32_t incl = wave_inclusive_scan_u32(cnt);
            const uint32_t base = incl - cnt;
            uint32_t list_lo = 0, list_hi = 0;   // bit b: this pixel blended entry c + b
            {
                const uint32_t sh = lane & 31u;
                const bool upper = lane >= 32u;
                for (uint32_t t = (uint32_t)nz; t; t &= t - 1) {
                    const int b = __builtin_ctz(t);
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)bal_lo, b);
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)bal_hi, b);
                    list_lo |= (((upper ? hi : lo) >> sh) & 1u) << b;
                }
                for (uint32_t t = (uint32_t)(nz >> 32); t; t &= t - 1) {
                    const int b = __builtin_ctz(t);
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)bal_lo, b + 32);
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)bal_hi, b + 32);
                    list_hi |= (((upper ? hi : lo) >> sh) & 1u) << b;
                }
            }
            // sub-ranges [b0, b1) of entries whose records fit the record buffer (normally the whole group)
            uint32_t b0 = 0;
            while (b0 < 64u) {
                const uint32_t start = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)b0);
                uint64_t fits = __builtin_amdgcn_ballot_w64(incl - start <= (uint32_t)CAP);   // monotone from lane b0 on
                fits |= (1ull << b0) - 1ull;
                const uint32_t b1 = fits == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~fits);
                const uint64_t rmask = (b1 < 64u ? (1ull << b1) - 1ull : ~0ull) & ~((1ull << b0) - 1ull);
                const uint64_t nzr = nz & rmask;
                b0 = b1;
                if (nzr == 0) continue;
                const bool in_range = ((rmask >> lane) & 1ull) != 0;
                s_tab[wave][lane] = make_uint4(bal_lo, bal_hi, base - start, cnt);
                if (in_range && cnt != 0) {
                    const uint32_t k = __builtin_amdgcn_mbcnt_hi((uint32_t)(nzr >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nzr, 0u));
                    s_nz[wave][k] = (uint8_t)lane;
                }
                const uint32_t nnz = (uint32_t)__builtin_popcountll(nzr);
                __builtin_amdgcn_wave_barrier();
                // ---------------- pass B: every pixel walks its own list ----------------
                if (!(ablate & 1)) {
                    uint64_t m = (((uint64_t)list_hi << 32) | list_lo) & rmask;
                    uint4 row; float2 xy; float4 co, fd;
                    bool have = m != 0;
                    if (have) {
                        const uint32_t b = (uint32_t)__builtin_ctzll(m);
                        m &= m - 1;
                        row = s_tab[wave][b]; xy = s_xy[c + b]; co = s_co[c + b]; fd = s_fd[c + b];
                    }
                    while (have) {
                        // next entry's data is requested before this entry's arithmetic (LDS latency under the VALU chain)
                        const bool have_next = m != 0;
                        uint4 row_n = row; float2 xy_n = xy; float4 co_n = co, fd_n = fd;
                        if (have_next) {
                            const uint32_t b = (uint32_t)__builtin_ctzll(m);
                            m &= m - 1;
                            row_n = s_tab[wave][b]; xy_n = s_xy[c + b]; co_n = s_co[c + b]; fd_n = s_fd[c + b];
                        }
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(row.y, __builtin_amdgcn_mbcnt_lo(row.x, 0u));
                        const float dx = xy.x - pixf_x, dy = xy.y - pixf_y;
                        const float power = power_exact(__fmul_rn(__fmul_rn(co.x, dx), dx), __fmul_rn(co.y, dx), co.z, dy);
                        const float G = __expf(power);
                        const float alpha = fminf(0.99f, co.w * G);
                        const float inv = __builtin_amdgcn_rcpf(1.f - alpha);   // shared by T/(1-a), T_final/(1-a)
                        T = T * inv;
                        const float sdot = fd.x * dLp0 + fd.y * dLp1 + fd.z * dLp2 + fd.w * dLpd + dLa;
                        A = last_alpha * last_s + (1.f - last_alpha) * A;
                        last_s = sdot;
                        last_alpha = alpha;
                        const float dL_dopa = (sdot - A) * T - inv * bgT;
                        s_rec[wave][row.z + rank] = make_float2(alpha * T, G * dL_dopa);
                        s_rid[wave][row.z + rank] = (uint8_t)lane;
                        row = row_n; xy = xy_n; co = co_n; fd = fd_n;
                        have = have_next;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                // ---------------- pass C: lane = quarter of an entry's records ----------------
                if (!(ablate & 2)) {
                    for (uint32_t it0 = 0; it0 < 4u * nnz; it0 += 64u) {     // whole quads: 4 nnz is a multiple of 4
                        const uint32_t it = it0 + lane;
                        const bool active = it < 4u * nnz;
                        const uint32_t b = active ? s_nz[wave][it >> 2] : 0u;
                        const uint4 row = s_tab[wave][b];
                        const uint32_t qlen = (row.w + 3u) >> 2;
                        const uint32_t r0 = (it & 3u) * qlen;
                        const uint32_t r1 = active ? min(row.w, r0 + qlen) : r0;
                        const float2 xy = s_xy[c + b];
                        const float4 co = s_co[c + b];
                        const float ex = xy.x - tile_x0, ey = xy.y - (tile_y0 + 4.0f * (float)wave);   // relative to the strip
                        float v[kAcc];
#pragma unroll
                        for (int k = 0; k < kAcc; k++) v[k] = 0.f;
                        float sx = 0, sy = 0;
                        for (uint32_t r = row.z + r0; r < row.z + r1; r++) {
                            const float2 rc = s_rec[wave][r];
                            const uint32_t l = s_rid[wave][r];
                            const float4 g4 = s_pix[wave * 64u + l];
                            const float dx = ex - (float)(l & 15u), dy = ey - (float)(l >> 4);
                            v[0] += rc.x * g4.x; v[1] += rc.x * g4.y; v[2] += rc.x * g4.z; v[3] += rc.x * g4.w;
                            v[9] += rc.y;
                            const float gdx = rc.y * dx, gdy = rc.y * dy;
                            sx += gdx; sy += gdy;
                            v[6] += gdx * dx; v[7] += gdx * dy; v[8] += gdy * dy;
                        }
                        // dL_dG G = opacity * (G dL/dalpha): the common factor of the geometric terms (backward.cu:580-598)
                        const float o = co.w;
                        v[4] = -ddelx_dx * o * (co.x * sx + co.y * sy);
                        v[5] = -ddely_dy * o * (co.z * sy + co.y * sx);
                        v[6] *= -0.5f * o; v[7] *= -0.5f * o; v[8] *= -0.5f * o;
                        // the four quarters of an entry sit in one quad: two DPP adds per value; the quad's first lane is
                        // the only writer of this (strip, entry) row (ds_add_f32 costs ~60 LDS cycles per wave-instruction
                        // plus ~2 per lane on this part -- ten of them per group were half of the kernel)
#pragma unroll
                        for (int k = 0; k < kAcc; k++) {
                            v[k] += dpp_term<0xB1, 0xf, 0xf>(v[k]);   // quad_perm [1,0,3,2]
                            v[k] += dpp_term<0x4E, 0xf, 0xf>(v[k]);   // quad_perm [2,3,0,1]
                        }
                        if (active && (lane & 3u) == 0) {
                            float* dst = &s_acc[wave][(c + b) * kAcc];     // 8-byte aligned (40-byte rows)
                            *reinterpret_cast<float2*>(dst + 0) = make_float2(v[0], v[1]);
                            *reinterpret_cast<float2*>(dst + 2) = make_float2(v[2], v[3]);
                            *reinterpret_cast<float2*>(dst + 4) = make_float2(v[4], v[5]);
                            *reinterpret_cast<float2*>(dst + 6) = make_float2(v[6], v[7]);
                            *reinterpret_cast<float2*>(dst + 8) = make_float2(v[8], v[9]);
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        // flush this round: the four strips' rows are added in a fixed order and stored as the row of list position
        // range.y - 1 - (round_base + e): no atomics, one writer per row, bitwise reproducible
        for (int e = tid; e < n; e += THREADS) {
            float2* dst = reinterpret_cast<float2*>(acc + (size_t)(range.y - 1u - (uint32_t)(round_base + e)) * kAcc);
#pragma unroll
            for (int k = 0; k < kAcc; k += 2) {
                float2 t = make_float2(0.f, 0.f);
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    float2* src = reinterpret_cast<float2*>(&s_acc[w][e * kAcc + k]);
                    const float2 u = *src;
                    t.x += u.x; t.y += u.y;
                    if (u.x != 0.f || u.y != 0.f) *src = make_float2(0.f, 0.f);
                }
                dst[k / 2] = t;
            }
        }
    }
}

}  // namespace

void launch_render_forward(hipStream_t s, int V, int W, int H, int tiles_x, int tiles_y, const uint2* ranges,
                           const uint32_t* point_list, const GeomState& g, const float* bg, float* out_color,
                           float* out_depth, float* out_alpha, uint32_t* n_contrib, uint2* pair_counts,
                           uint64_t* ballots, uint32_t R)
{
    const uint32_t tiles_total = (uint32_t)V * tiles_x * tiles_y;
    hipLaunchKernelGGL(render_forward_kernel, dim3(tiles_total), dim3(kTilePix), 0, s, W, H, (uint32_t)tiles_x,
                       (uint32_t)tiles_y, tiles_total, ranges, point_list, g.means2D, g.conic_opacity, g.rgbd, bg,
                       out_color, out_depth, out_alpha, n_contrib, pair_counts, ballots, R);
}

void launch_render_backward(hipStream_t s, int V, int W, int H, int tiles_x, int tiles_y, const uint2* ranges,
                            const uint32_t* point_list, const GeomState& g, const float* bg, const float* alphas,
                            const uint32_t* n_contrib, const float* dL_dpix, const float* dL_dpix_depth,
                            const float* dL_dalphas, float* acc, const uint64_t* ballots, uint32_t R)
{
    const uint32_t tiles_total = (uint32_t)V * tiles_x * tiles_y;
    // GD_RASTER_BWD_IMPL=tree selects the wave-uniform walk with the halving-tree reduction (round 1; kept for A/B
    // measurements and as a second implementation the parity tests run); default: per-pixel lists.  Read once.
    static const int impl = [] {
        const char* e = getenv("GD_RASTER_BWD_IMPL");
        return (e && e[0] == 't') ? 1 : 0;
    }();
    static const int ablate = [] { const char* e = getenv("GD_RASTER_BWD_ABLATE"); return e ? atoi(e) : 0; }();
    if (impl == 1) {
        hipLaunchKernelGGL(render_backward_kernel<1>, dim3(tiles_total), dim3(kTilePix), 0, s, W, H,
                           (uint32_t)tiles_x, (uint32_t)tiles_y, tiles_total, ranges, point_list, g.means2D,
                           g.conic_opacity, g.rgbd, bg, alphas, n_contrib, dL_dpix, dL_dpix_depth, dL_dalphas, acc);
    } else {
        hipLaunchKernelGGL((render_backward_lists_kernel<GD_BWD_ROUND, GD_BWD_CAP>), dim3(tiles_total), dim3(kTilePix), 0, s, W, H,
                           (uint32_t)tiles_x, (uint32_t)tiles_y, tiles_total, ranges, point_list, g.means2D,
                           g.conic_opacity, g.rgbd, bg, alphas, n_contrib, dL_dpix, dL_dpix_depth, dL_dalphas, acc,
                           ballots, R, ablate);
    }
}

}  // namespace gd
