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

#ifndef GD_BWD_CAP
#define GD_BWD_CAP 512   // records (contributing pairs) a strip buffers per group of 64 list entries
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

// Single, separately rounded fp32 operations.  (HIP's __fmul_rn / __fadd_rn are ordinary inline functions compiled
// with the default fast contraction: after inlining the compiler still fuses them into FMAs -- measured: 7 % of the
// pixels of a forward pass differed from the oracle by one ulp.  Operators written under `fp contract(off)` carry no
// contraction licence.)
__device__ __forceinline__ float mul_rn(float a, float b)
{
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b)
{
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b)
{
#pragma clang fp contract(off)
    return a - b;
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

// exp of the blend, defined operation by operation (oracle/gd_oracle.c gd_expf: Cody-Waite reduction + Cephes degree-5
// polynomial, every step one correctly rounded fp32 operation): the forward pass then reproduces the oracle BIT FOR BIT
// -- the blended pairs, n_contrib, the pixels and the alpha image whose complement is the backward pass's T_final.
__device__ __forceinline__ float gd_expf(float x)
{
#pragma clang fp contract(off)
    if (x < -87.0f) return 0.0f;
    const float n = rintf(mul_rn(x, 1.44269504088896341f));
    float r = __fmaf_rn(n, -0.693359375f, x);
    r = __fmaf_rn(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __fmaf_rn(p, r, 1.3981999507e-3f);
    p = __fmaf_rn(p, r, 8.3334519073e-3f);
    p = __fmaf_rn(p, r, 4.1665795894e-2f);
    p = __fmaf_rn(p, r, 1.6666665459e-1f);
    p = __fmaf_rn(p, r, 5.0000001201e-1f);
    p = __fmaf_rn(p, mul_rn(r, r), r);
    p = add_rn(p, 1.0f);
    return ldexpf(p, (int)n);
}

// Smallest fp32 exponent p with  !(min(0.99, o * expf(p)) < 1/255): `power >= thr` is then the forward pass's
// contribution test itself (forward.cu:346-348), decided without evaluating the exponential per pair.
// A few accurate expf per STAGED entry (once per tile and entry).
__device__ __forceinline__ float alpha_threshold_exact(const float o)
{
#pragma clang fp contract(off)
    const float k = 1.0f / 255.0f;
    float t = -logf(255.0f * o);
    if (!(o > 0.0f) || !isfinite(t)) return INFINITY;     // opacity 0 (or NaN): nothing ever contributes
#pragma unroll 1
    for (int it = 0; it < 8; it++) {      // walk down while the next lower exponent still passes
        const float d = nextafterf(t, -INFINITY);
        if (fminf(0.99f, mul_rn(o, gd_expf(d))) < k) break;
        t = d;
    }
#pragma unroll 1
    for (int it = 0; it < 16; it++) {     // walk up while this exponent fails
        if (!(fminf(0.99f, mul_rn(o, gd_expf(t))) < k)) break;
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
    // every blend operation is rounded on its own, like the oracle's (-ffp-contract=off): images match it bit for bit
#pragma clang fp contract(off)
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
    uint32_t ballots_written = 0;   // list positions [0, ballots_written) of this wave's strip have their ballot stored

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
            // walks exactly these (pixel, entry) pairs (ballots[list position][strip]; zero = nobody, also for the
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
                    const float power = power_exact(mul_rn(mul_rn(co.x, dx), dx), mul_rn(co.y, dx), co.z, dy);
                    // alpha >= 1/255  <=>  power >= s_thr (exact, alpha_threshold_exact): only contributing pairs pay
                    // for the exponential
                    if (!(power > 0.0f) && !(power < s_thr[j])) {
                        const float alpha = fminf(0.99f, mul_rn(co.w, gd_expf(power)));
                        {
                            // the oracle's operations, each rounded on its own (forward.cu:349-363 without contraction)
                            const float test_T = mul_rn(T, sub_rn(1.0f, alpha));
                            if (test_T < 0.0001f) {
                                done = true;
                                contributor = cbase + (uint32_t)j + 1u;
                            } else {
                                const float4 fd = s_fd[j];
                                C0 = add_rn(C0, mul_rn(mul_rn(fd.x, alpha), T));
                                C1 = add_rn(C1, mul_rn(mul_rn(fd.y, alpha), T));
                                C2 = add_rn(C2, mul_rn(mul_rn(fd.z, alpha), T));
                                weight = add_rn(weight, mul_rn(alpha, T));
                                Dd = add_rn(Dd, mul_rn(mul_rn(fd.w, alpha), T));
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
                ballots[(size_t)(range.x + cbase + (uint32_t)c + lane) * 4u + wave] = ((uint64_t)wb_hi << 32) | wb_lo;
            ballots_written = min((uint32_t)total, cbase + (uint32_t)c + 64u);
        }
    }
    // the entries this wave never looked at (its pixels saturated earlier) blend nothing: every (position, strip) word
    // of the tile is defined, the backward pass and its per-Gaussian gather rely on it
    for (uint32_t p = ballots_written + lane; p < (uint32_t)total; p += 64u) ballots[(size_t)(range.x + p) * 4u + wave] = 0;
    if (inside) {
        n_contrib[(size_t)view * HW + pix_id] = last_contributor;
        pair_counts[(size_t)view * HW + pix_id] = make_uint2(contributor, blended);
        float* oc = out_color + (size_t)view * 3 * HW;
        oc[0 * HW + pix_id] = add_rn(C0, mul_rn(T, bg_color[0]));
        oc[1 * HW + pix_id] = add_rn(C1, mul_rn(T, bg_color[1]));
        oc[2 * HW + pix_id] = add_rn(C2, mul_rn(T, bg_color[2]));
        out_alpha[(size_t)view * HW + pix_id] = weight;
        out_depth[(size_t)view * HW + pix_id] = Dd;
    }
}

constexpr int kAcc = 10;  // colour rgb, depth, mean2D xy, conic x/y/w, opacity

// =================================================================================================================
// The reverse-order gradient pass, organised around PER-PIXEL LISTS (render_backward_strip_kernel below).
//
// A wave-uniform walk of the tile's list (the reference's organisation, and round 1's) executes its ~60-instruction
// body for all 64 lanes of a strip although on the benchmark scene only a third of the 64 pixels blend a given (strip,
// entry) pair -- and for half of the pairs a strip-level culling lets through, none does -- and then pays a
// 38-instruction cross-lane reduction per pair: VALU-issue bound at 7 % of the fp32 roof.  Here nothing is evaluated
// for a pair that does not contribute:
//
//   *  the forward pass leaves, per (strip, list position), the 64-bit ballot of the pixels that blended the entry
//      (render_forward_kernel, `ballots`): exactly the pairs backward.cu:517-533 lets through, so the reverse pass
//      needs neither the contribution test nor the strip culling;
//   A  per dense group (up to 64 entries that HAVE records, compacted from the strip's part of the list) a wave turns
//      the ballots' popcounts into record offsets with one DPP scan and transposes the ballots into a per-PIXEL list
//      (lane = pixel, bit = entry);
//   B  (lane = pixel, each lane walks ITS OWN list)  the sequential part of backward.cu:517-578 -- exp, alpha,
//      T /= (1 - alpha), the accumulated-colour recurrence, dL/dalpha.  Lanes advance independently, and the five
//      recurrences of the reference (3 colours, depth, alpha) collapse into ONE because only their dot product with
//      the pixel's (dL/dC, dL/ddepth, dL/dalpha) is ever used.  Output: a record {alpha T, G dL/dalpha, pixel} per
//      contributing pair at slot base[entry] + rank-of-the-pixel-in-the-entry's-ballot (v_mbcnt): grouped by entry;
//   C  (lane = quarter of an entry's records)  gathers the records and accumulates the ten sums of
//      backward.cu:555-598 in registers -- colour / depth gradients and the moments sum w, sum w d, sum w d d^T of
//      the pixel offsets d, from which dL/dmean2D, dL/dconic and dL/dopacity follow by one multiplication per entry.
//      Two DPP adds per value combine the quad; its first lane stores the row rows4[slot][strip][kAcc] and the flag.
//      Rows are written once and never accumulated: instance_sum_kernel (raster_preprocess.hip) adds the flagged rows
//      of each (view, Gaussian).
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

// =================================================================================================================
// render_backward_strip_kernel -- one INDEPENDENT wave per (tile, 16x4 strip): the list-based reverse pass described
// above, without any workgroup barrier.
//
// With the forward pass's ballots a strip knows which list entries it needs (about a fifth of the tile's list), so
// nothing has to be staged by the tile as a whole.  The wave streams the ballots / Gaussian ids / instance slots of its
// part of the list in chunks of 64 positions (three chunks in flight) and COMPACTS the entries with a non-zero ballot
// into DENSE GROUPS of up to 64 (v_mbcnt ranks; a chunk that does not fit is split across two groups) -- most strips
// have fewer than 64 such entries, i.e. exactly one group -- gathers the centre / conic / colour of just those, runs
// the passes A-C described above on the group and stores ONE row of ten sums per (instance, strip) that has a
// contributing pixel -- rows4[slot][strip][10] + a flag byte, written once, never accumulated; `slot` is where
// duplicate_kernel put the instance, so the rows of a Gaussian are contiguous and instance_sum_kernel just adds the
// flagged ones.  (Groups of 64 list POSITIONS, as first written, carried ~13 useful entries each: 4-8 % slower on
// every workload tried -- per-group overheads and shorter per-pixel lists.)
// No atomics, no __syncthreads, no zero filling of rows; gradients are bitwise reproducible.
// =================================================================================================================
template <int CAP>
__global__ __launch_bounds__(64) void render_backward_strip_kernel(
    int W, int H, uint32_t gx, uint32_t gy, uint32_t tiles_total, const uint2* __restrict__ ranges,
    const uint32_t* __restrict__ point_list, const float2* __restrict__ means2D,
    const float4* __restrict__ conic_opacity, const float4* __restrict__ rgbd, const float* __restrict__ bg_color,
    const float* __restrict__ alphas, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
    const float* __restrict__ dL_dpixel_depths, const float* __restrict__ dL_dalphas, float* __restrict__ rows4,
    uint8_t* __restrict__ flags, const uint64_t* __restrict__ ballots, const uint32_t* __restrict__ slot_of, int ablate)
{
    static_assert(CAP >= 64, "one entry has up to 64 records");
    __shared__ float2 s_xy[64];            // the current group's entries (only those with a non-zero ballot are filled)
    __shared__ float4 s_co[64];
    __shared__ float4 s_fd[64];
    __shared__ uint4 s_tab[64];            // {ballot lo, hi, record base | count << 16, instance slot}
    __shared__ float2 s_rec[CAP];          // {alpha T, G dL/dalpha} per contributing pair, grouped by entry
    __shared__ uint8_t s_rid[CAP];         // ... and its pixel (lane)
    __shared__ float4 s_pix[64];           // per pixel: dL/dC rgb, dL/ddepth

    // workgroup b runs on XCD b % 8: the four strips of a tile and neighbouring tiles share an XCD (they share most
    // of their Gaussians in that XCD's L2)
    const uint32_t nblk = tiles_total * 4u;
    uint32_t unit = blockIdx.x;
    if ((nblk & 7u) == 0) unit = (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3);
    const uint32_t tile = unit >> 2;
    const uint32_t strip = unit & 3u;
    const uint32_t tpv = gx * gy;
    const uint32_t view = tile / tpv;
    const uint32_t lt = tile - view * tpv;
    const uint32_t ty = lt / gx, tx = lt - ty * gx;
    const uint32_t lane = threadIdx.x;
    const uint32_t px = tx * kTile + (lane & 15u), py = ty * kTile + 4u * strip + (lane >> 4);
    const float pixf_x = (float)px, pixf_y = (float)py;
    const size_t HW = (size_t)H * W;
    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    const float strip_x0 = (float)(tx * kTile), strip_y0 = (float)(ty * kTile + 4u * strip);

    // ---- per-pixel constants and the state of the reverse walk (backward.cu:461-487) ----
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const size_t pix_id = (size_t)view * HW + (size_t)W * py + px;
    const float T_final = inside ? (1 - alphas[pix_id]) : 0;
    float T = T_final;
    const uint32_t last_contributor = inside ? n_contrib[pix_id] : 0;
    const uint32_t wmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(last_contributor));
    if (wmax == 0) return;        // no pixel of the strip blended anything: all its ballots are zero, no rows
    float dLp0 = 0, dLp1 = 0, dLp2 = 0, dLpd = 0, dLa = 0;
    if (inside) {
        const float* dp = dL_dpixels + (size_t)view * 3 * HW + ((size_t)W * py + px);
        dLp0 = dp[0]; dLp1 = dp[HW]; dLp2 = dp[2 * HW];
        dLpd = dL_dpixel_depths[pix_id];
        dLa = dL_dalphas[pix_id];
    }
    const float bgT = T_final * (bg_color[0] * dLp0 + bg_color[1] * dLp1 + bg_color[2] * dLp2);
    s_pix[lane] = make_float4(dLp0, dLp1, dLp2, dLpd);
    float A = 0.f, last_alpha = 0.f, last_s = 0.f;   // A = sum_k accum_rec_k dL_k of the reference's five recurrences
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

    // reverse index e = 0 is the LAST list entry; the strip's first useful one is e = total - wmax.
    // Chunk stream: 64 list positions per chunk (lane = position), three chunks in flight.
    const int jstart = total - (int)wmax;
    auto load_chunk = [&](int c, uint64_t& bal, uint32_t& id, uint32_t& slot) {
        const int e = c + (int)lane;
        bal = 0; id = 0; slot = 0;
        if (e < total && e >= jstart) {
            const uint32_t pos = range.y - 1u - (uint32_t)e;
            bal = ballots[(size_t)pos * 4u + strip];
            id = point_list[pos];
            slot = slot_of[pos];
        }
    };
    int c = jstart & ~63;
    uint64_t bal_cur, bal_nxt, bal_nn;
    uint32_t id_cur, id_nxt, id_nn, slot_cur, slot_nxt, slot_nn;
    load_chunk(c, bal_cur, id_cur, slot_cur);
    load_chunk(c + 64, bal_nxt, id_nxt, slot_nxt);
    load_chunk(c + 128, bal_nn, id_nn, slot_nn);
    uint64_t avail = ~0ull;          // lanes of the current chunk not yet handed to a group

    while (c < total) {
        // ---------------- compaction: the next (up to) 64 entries that have records, in list order ----------------
        uint32_t fill = 0;
        while (c < total) {
            const uint64_t nzmask = __builtin_amdgcn_ballot_w64(bal_cur != 0) & avail;
            const uint32_t cnt_nz = (uint32_t)__builtin_popcountll(nzmask);
            const uint32_t room = 64u - fill;
            const uint32_t myrank = __builtin_amdgcn_mbcnt_hi((uint32_t)(nzmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nzmask, 0u));
            const bool takeme = ((nzmask >> lane) & 1ull) != 0 && myrank < room;
            if (takeme) s_tab[fill + myrank] = make_uint4((uint32_t)bal_cur, (uint32_t)(bal_cur >> 32), id_cur, slot_cur);
            if (cnt_nz <= room) {
                fill += cnt_nz;
                c += 64;
                bal_cur = bal_nxt; id_cur = id_nxt; slot_cur = slot_nxt;
                bal_nxt = bal_nn; id_nxt = id_nn; slot_nxt = slot_nn;
                load_chunk(c + 128, bal_nn, id_nn, slot_nn);
                avail = ~0ull;
                if (fill == 64u) break;
            } else {
                avail = nzmask & ~__builtin_amdgcn_ballot_w64(takeme);
                fill = 64u;
                break;
            }
        }
        if (fill == 0) break;
        __builtin_amdgcn_wave_barrier();
        {
            // ---------------- lane j = staged entry j: its data, record offsets, the per-pixel lists ----------------
            const bool staged = lane < fill;
            const uint4 ent = staged ? s_tab[lane] : make_uint4(0, 0, 0, 0);
            const uint32_t bal_lo = ent.x, bal_hi = ent.y, slot_cur_e = ent.w;
            if (staged) { s_xy[lane] = means2D[ent.z]; s_co[lane] = conic_opacity[ent.z]; s_fd[lane] = rgbd[ent.z]; }
            const uint32_t cnt = (uint32_t)__builtin_popcount(bal_lo) + (uint32_t)__builtin_popcount(bal_hi);
            const uint32_t incl = wave_inclusive_scan_u32(cnt);
            const uint32_t base = incl - cnt;
            uint32_t list_lo = 0, list_hi = 0;   // bit j: this pixel blended staged entry j
            {
                const uint32_t sh = lane & 31u;
                const bool upper = lane >= 32u;
                const int f0 = (int)(fill < 32u ? fill : 32u);
                for (int b = 0; b < f0; b++) {
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)bal_lo, b);
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)bal_hi, b);
                    list_lo |= (((upper ? hi : lo) >> sh) & 1u) << b;
                }
                for (int b = 32; b < (int)fill; b++) {
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)bal_lo, b);
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)bal_hi, b);
                    list_hi |= (((upper ? hi : lo) >> sh) & 1u) << (b - 32);
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
                const uint32_t bs = b0;                       // staged entries [bs, min(b1, fill)) are this sub-range
                b0 = b1;
                if (bs >= fill) break;
                const uint32_t nnz = (b1 < fill ? b1 : fill) - bs;
                s_tab[lane] = make_uint4(bal_lo, bal_hi, ((base - start) & 0xffffu) | (cnt << 16), slot_cur_e);
                __builtin_amdgcn_wave_barrier();
                // ---------------- pass B: every pixel walks its own list ----------------
                if (!(ablate & 1)) {
                    struct Ent { uint4 row; float2 xy; float4 co, fd; };
                    // the list is walked as two 32-bit halves (entries 0-31, then 32-63): one v_ffbl per pop
                    for (uint32_t half = 0; half < 2u; half++) {
                        uint32_t m = (half ? list_hi : list_lo) & (uint32_t)(rmask >> (32u * half));
                        const uint32_t hoff = 32u * half;
                        auto fetch = [&](Ent& q) {           // pops the list's next entry and requests its data
                            const uint32_t b = (uint32_t)__builtin_ctz(m) + hoff;
                            m &= m - 1u;
                            q.row = s_tab[b]; q.xy = s_xy[b]; q.co = s_co[b]; q.fd = s_fd[b];
                        };
                        auto step = [&](const Ent& q) {      // backward.cu:534-578 for one contributing (pixel, entry) pair
                            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(q.row.y, __builtin_amdgcn_mbcnt_lo(q.row.x, 0u));
                            const float dx = q.xy.x - pixf_x, dy = q.xy.y - pixf_y;
                            // the forward pass's operation order: needle-shaped splats cancel to a few ulps here and a
                            // re-associated power (5 fused ops: -3.5 % kernel time) fails the needle parity test
                            const float power = power_exact(mul_rn(mul_rn(q.co.x, dx), dx), mul_rn(q.co.y, dx), q.co.z, dy);
                            const float G = __expf(power);
                            const float alpha = fminf(0.99f, q.co.w * G);
                            const float inv = __builtin_amdgcn_rcpf(1.f - alpha);   // shared by T/(1-a), T_final/(1-a)
                            T = T * inv;
                            const float sdot = q.fd.x * dLp0 + q.fd.y * dLp1 + q.fd.z * dLp2 + q.fd.w * dLpd + dLa;
                            A = last_alpha * last_s + (1.f - last_alpha) * A;
                            last_s = sdot;
                            last_alpha = alpha;
                            const float dL_dopa = (sdot - A) * T - inv * bgT;
                            const uint32_t at = (q.row.z & 0xffffu) + rank;
                            s_rec[at] = make_float2(alpha * T, G * dL_dopa);
                            s_rid[at] = (uint8_t)lane;
                        };
                        // two-deep software pipeline, unrolled so that the two entry buffers never need copying
                        if (m != 0) {
                            Ent e0, e1;
                            fetch(e0);
                            while (true) {
                                const bool more1 = m != 0;
                                if (more1) fetch(e1);
                                step(e0);
                                if (!more1) break;
                                const bool more0 = m != 0;
                                if (more0) fetch(e0);
                                step(e1);
                                if (!more0) break;
                            }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                // ---------------- pass C: lane = quarter of an entry's records ----------------
                if (!(ablate & 2)) {
                    for (uint32_t it0 = 0; it0 < 4u * nnz; it0 += 64u) {     // whole quads: 4 nnz is a multiple of 4
                        const uint32_t it = it0 + lane;
                        const bool active = it < 4u * nnz;
                        const uint32_t b = active ? bs + (it >> 2) : 0u;
                        const uint4 row = s_tab[b];
                        const uint32_t rcnt = row.z >> 16, rbase = row.z & 0xffffu;
                        const uint32_t qlen = (rcnt + 3u) >> 2;
                        const uint32_t r0 = (it & 3u) * qlen;
                        const uint32_t r1 = active ? min(rcnt, r0 + qlen) : r0;
                        const float2 xy = s_xy[b];
                        const float4 co = s_co[b];
                        const float ex = xy.x - strip_x0, ey = xy.y - strip_y0;
                        float v[kAcc];
#pragma unroll
                        for (int k = 0; k < kAcc; k++) v[k] = 0.f;
                        float sx = 0, sy = 0;
                        struct Rec { float2 rc; uint32_t l; float4 g4; };
                        auto accum = [&](const Rec& q) {
                            const float dx = ex - (float)(q.l & 15u), dy = ey - (float)(q.l >> 4);
                            v[0] += q.rc.x * q.g4.x; v[1] += q.rc.x * q.g4.y; v[2] += q.rc.x * q.g4.z; v[3] += q.rc.x * q.g4.w;
                            v[9] += q.rc.y;
                            const float gdx = q.rc.y * dx, gdy = q.rc.y * dy;
                            sx += gdx; sy += gdy;
                            v[6] += gdx * dx; v[7] += gdx * dy; v[8] += gdy * dy;
                        };
                        uint32_t r = rbase + r0;
                        const uint32_t rend = rbase + r1;
                        auto fetchr = [&](Rec& q) {
                            q.rc = s_rec[r];
                            q.l = s_rid[r];
                            q.g4 = s_pix[q.l];
                            r++;
                        };
                        if (r < rend) {
                            Rec q0, q1;
                            fetchr(q0);
                            while (true) {
                                const bool more1 = r < rend;
                                if (more1) fetchr(q1);
                                accum(q0);
                                if (!more1) break;
                                const bool more0 = r < rend;
                                if (more0) fetchr(q0);
                                accum(q1);
                                if (!more0) break;
                            }
                        }
                        // dL_dG G = opacity * (G dL/dalpha): the common factor of the geometric terms (backward.cu:580-598)
                        const float o = co.w;
                        v[4] = -ddelx_dx * o * (co.x * sx + co.y * sy);
                        v[5] = -ddely_dy * o * (co.z * sy + co.y * sx);
                        v[6] *= -0.5f * o; v[7] *= -0.5f * o; v[8] *= -0.5f * o;
                        // the four quarters of an entry sit in one quad: two DPP adds per value; the quad's first lane
                        // stores the row of (list position, strip)
#pragma unroll
                        for (int k = 0; k < kAcc; k++) {
                            v[k] += dpp_term<0xB1, 0xf, 0xf>(v[k]);   // quad_perm [1,0,3,2]
                            v[k] += dpp_term<0x4E, 0xf, 0xf>(v[k]);   // quad_perm [2,3,0,1]
                        }
                        if (active && (lane & 3u) == 0) {
                            const size_t at = (size_t)row.w * 4u + strip;     // (instance slot, strip)
                            flags[at] = 1;
                            float2* dst = reinterpret_cast<float2*>(rows4 + at * kAcc);
                            dst[0] = make_float2(v[0], v[1]);
                            dst[1] = make_float2(v[2], v[3]);
                            dst[2] = make_float2(v[4], v[5]);
                            dst[3] = make_float2(v[6], v[7]);
                            dst[4] = make_float2(v[8], v[9]);
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
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
                            const float* dL_dalphas, float* rows4, uint8_t* flags, const uint64_t* ballots,
                            const uint32_t* slot_of)
{
    const uint32_t tiles_total = (uint32_t)V * tiles_x * tiles_y;
    // GD_RASTER_BWD_ABLATE (timing experiments only, wrong results): 1 = skip pass B, 2 = skip pass C.  Read once.
    static const int ablate = [] { const char* e = getenv("GD_RASTER_BWD_ABLATE"); return e ? atoi(e) : 0; }();
    hipLaunchKernelGGL((render_backward_strip_kernel<GD_BWD_CAP>), dim3(tiles_total * 4u), dim3(64), 0, s, W, H,
                       (uint32_t)tiles_x, (uint32_t)tiles_y, tiles_total, ranges, point_list, g.means2D,
                       g.conic_opacity, g.rgbd, bg, alphas, n_contrib, dL_dpix, dL_dpix_depth, dL_dalphas, rows4,
                       flags, ballots, slot_of, ablate);

}

}  // namespace gd
