// nn_conv3x3.hip -- 3x3 / stride 1 / pad 1 convolution on NHWC bf16 as an implicit GEMM on
// gfx950 matrix cores (v_mfma_f32_32x32x16_bf16), with fused per-image bias and residual add.
//
// This is the dominant dense contraction of the SDS guidance step: ~85 % of the VAE-encoder and
// ~55 % of the UNet FLOPs are 3x3 convolutions (diffusers' ResnetBlock2D / Downsample / Upsample,
// dispatched by PyTorch to MIOpen in the reference: call sites
// Garment_3DGS/threestudio/models/guidance/stable_diffusion_guidance.py:153-157,165-166).  MIOpen's
// kernels for these shapes run at 300-490 TFLOP/s on MI355X (tools/conv_shapes_bench.py).
//
// GEMM view: out[m][co] = sum_{tap, ci} in[pixel m shifted by tap][ci] * w[co][tap][ci],
// M = N*H*W pixels, N = Cout, K = 9*Cin.  Both operands are K-contiguous in HBM (NHWC
// activations; weights stored [Cout][3][3][Cin] = PyTorch's channels_last weight layout), so a
// K-step of BK = 64 channels of ONE tap is a plain 128-byte row per pixel / per output channel.
//
// Workgroup = 4 wave64 (2x2), tile 128 output channels x 128 pixels, each wave 64x64 = 2x2 MFMA
// tiles of 32x32; the weight tile is the MFMA A operand and the pixel tile the B operand, so a
// lane ends up holding 4 consecutive output channels of one pixel per accumulator quad -> 8-byte
// NHWC stores without an LDS transpose.
//
// HBM -> LDS goes through global_load_lds_dwordx4 (no VGPR staging): the LDS image is lane-linear,
// so the XOR swizzle that keeps ds_read_b128 at <= 2-way bank conflicts is applied on the SOURCE
// address (which 16-byte chunk a lane fetches) and again on the fragment read.  Zero padding of
// the halo: out-of-image lanes fetch from a 16-byte zero buffer instead of branching.
// Two LDS stages (2 x 32 KiB): the loads of K-step s+1 are in flight while step s is multiplied.
//
// The same kernel computes the input gradient of the convolution (weights frozen, so dgrad is
// the only backward): conv3x3(dy, w') with w'[ci][tap][co] = w[co][8 - tap][ci].
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <stdio.h>

#include <mutex>
#include <vector>

#include "../../include/gd_nn.h"

#ifndef GD_CONV_ABLATE
#define GD_CONV_ABLATE 0
#endif

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// Timing-only switches of tools/nn_variants.sh (never defined in a product build):
//   GD_PATCH_EPI=1  the patch-staged kernels skip their epilogue stores (wrong results; the condition is never true at
//                   run time, so the compiler keeps the code that computes the values)
//   GD_PATCH_EPI=2  8-byte stores straight from the MFMA result layout instead of the LDS-transposed epilogue
//                   (correct results; the A/B of that epilogue)
#ifndef GD_PATCH_EPI
#define GD_PATCH_EPI 0
#endif
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int BK = 64;    // channels of one tap per K-step (128-byte rows)

__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f)
{
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two fp32 -> packed bf16 (round to nearest even) in ONE instruction: v_cvt_pk_bf16_f32 (gfx950)
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi)
{
    f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// byte offset of logical (row, 16-B chunk j) inside a swizzled [rows][64] bf16 tile image
__device__ __forceinline__ int swz(int row, int j)
{
    return (row >> 1) * 256 + (((((row & 1) << 3) | j) ^ ((row >> 1) & 15)) << 4);
}

__device__ __forceinline__ void bload_lds16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff,
                                            char* lds_wave_base)
{
    // buffer_load_dwordx4 ... offen lds: 16 B per lane straight into LDS (wave-uniform base +
    // lane*16); a lane whose voffset is beyond num_records gets ZEROS -- that is the halo padding
    // and the ragged-tile masking, with no branch and no zero page.
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff,
                                             soff, 0, 0);
}

constexpr uint32_t kOOB = 0x80000000u;   // voffset that fails the buffer range check (tensors are < 2 GiB)

// Geometry of one implicit-GEMM launch.  GEMM row m = (image n, grid point (a, b)) on an Hg x Wg grid; its
// K dimension walks `ntaps` taps x Cin channels, tap t reading input pixel (a*sy + ty[t], b*sx + tx[t]) (zero
// outside the image) against weight tap widx[t] of the [Cout][9][Cin] weight tensor; the result goes to output
// pixel (a*osy + ooy, b*osx + oox) of an Hout x Wout image.  This one description covers
//   3x3 / stride 1 / pad 1        : grid = image, taps (ky-1, kx-1)
//   3x3 / stride 2 / pad (lo, 1)  : grid = output image, sy = sx = 2, taps (ky - lo, kx - lo)
//   input gradient of the latter  : four launches, one per input-pixel parity class (py, px), each with only the
//                                   taps that reach that class (4 + 2 + 2 + 1 = 9 taps in total, no zero-insertion
//                                   waste), osy = osx = 2, (ooy, oox) = (py, px).
// ty / tx / widx are packed 4 bits per tap (ty, tx biased by +8) so the tap walk stays in scalar registers.
struct ConvGeom {
    int Hin, Win, Hg, Wg, sy, sx, Hout, Wout, osy, osx, ooy, oox, ntaps, back;
    int wtaps;            // taps per output channel in the weight tensor ([Cout][wtaps][Cin]; 0 = 9)
    uint64_t ty4, tx4, w4;
};

// Tile = BN output channels x BM pixels, WN x WM waves, each wave (BN/WN) x (BM/WM) built from
// 32x32x16 MFMAs.  bias_img_stride: 0 -> bias[co]; Cout -> bias[n][co].
template <int BN, int BM, int WN, int WM>
__device__ __forceinline__ void conv3x3_nhwc_bf16_body(
    const uint16_t* __restrict__ in, const uint16_t* __restrict__ wt, const uint16_t* __restrict__ bias,
    int bias_img_stride, const uint16_t* __restrict__ residual, uint16_t* __restrict__ out, int Nimg, const ConvGeom& g,
    int Cin, int Cout, int tiles_n, int nwg, float* __restrict__ partial, int steps_per_split, int bid_raw, int split_idx)
{
    constexpr int THREADS = 64 * WN * WM;
    constexpr int NA = BM * 8 / THREADS;      // 16-B chunks of the pixel tile per thread per K-step
    constexpr int NB = BN * 8 / THREADS;      // ... of the weight tile
    constexpr int FA = BN / WN / 32;          // MFMA tiles per wave along channels
    constexpr int FB = BM / WM / 32;          // ... along pixels
    constexpr int kStage = (BM + BN) * BK * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware tile order: consecutive logical tiles (same pixel tile, different Cout tile, then
    // the next pixel tile) stay on one XCD's L2.
    int bid = bid_raw;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int tn = bid % tiles_n, tm = bid / tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int HW = g.Hg * g.Wg;
    const int64_t M = (int64_t)Nimg * HW;
    const int H = g.Hin, W = g.Win;

    // Buffer descriptors.  The activation descriptor's base is moved back by `back` pixels (one image row +
    // one pixel for the pad-1 convolution) so that per-lane voffsets (base pixel of row m, chunk) and the
    // wave-uniform soffset (tap, channel step) are both non-negative; valid lanes never address bytes
    // before `in`.
    const uint32_t row_bytes = (uint32_t)Cin * 2u;
    const uint32_t back = (uint32_t)g.back * row_bytes;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)in - back), 0, (int)((uint32_t)Nimg * (uint32_t)(H * W) * row_bytes + back), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)wt, 0, (int)((uint32_t)Cout * (uint32_t)g.wtaps * row_bytes), 0x00020000);

    // LDS image: 256-byte lines = two consecutive 128-byte tile rows = 16 slots of 16 B; logical
    // slot c = (row&1)*8 + chunk is stored at slot c ^ (line & 15): a 64-lane fragment read then
    // touches 16 distinct slots per 16-lane service group (conflict free).  The image is written
    // lane-linearly by the LDS-DMA loads, so the permutation is applied to WHICH (row, chunk) a
    // lane fetches.
    uint32_t a_off[NA];              // byte offset of (pixel row, chunk) relative to rs_in, or kOOB
    int a_y[NA], a_x[NA];
    uint32_t b_off[NB];
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int q = tid + THREADS * i;
        const int line = q >> 4, c = (q & 15) ^ (line & 15);
        const int r = 2 * line + (c >> 3);
        const int64_t m = (int64_t)m0 + r;
        if (m < M) {
            const int nimg = (int)(m / HW);
            const int rem = (int)(m - (int64_t)nimg * HW);
            const int ga = rem / g.Wg;
            a_y[i] = ga * g.sy;
            a_x[i] = (rem - ga * g.Wg) * g.sx;
            a_off[i] = (uint32_t)((nimg * H + a_y[i]) * W + a_x[i]) * row_bytes + (uint32_t)(c & 7) * 16u;
        } else {
            a_y[i] = -100000; a_x[i] = 0; a_off[i] = kOOB;
        }
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
        const int q = tid + THREADS * i;
        const int line = q >> 4, c = (q & 15) ^ (line & 15);
        const int r = 2 * line + (c >> 3);
        const int co = n0 + r;
        b_off[i] = co < Cout ? (uint32_t)co * (uint32_t)g.wtaps * row_bytes + (uint32_t)(c & 7) * 16u : kOOB;
    }
    const int kc = Cin / BK;         // K-steps per tap
    // split-K (small-M layers): blockIdx.y walks contiguous ranges of the (tap, channel step) sequence, fp32 partial
    // sums go to partial[split]
    const int step0 = partial ? split_idx * steps_per_split : 0;
    const int step1 = partial ? min(g.ntaps * kc, step0 + steps_per_split) : g.ntaps * kc;
    const int nsteps = step1 - step0;
    const int tap0 = step0 / kc;

    // loader state: (tap, channel step) of the NEXT stage to fetch; halo validity is re-evaluated
    // once per tap, the per-K-step cost is one scalar add
    int ld_tap = tap0, ld_c = step0 - tap0 * kc;
    uint32_t a_voff[NA];
    auto set_tap = [&](int tap) {
        const int dy = (int)((g.ty4 >> (4 * tap)) & 15u) - 8, dx = (int)((g.tx4 >> (4 * tap)) & 15u) - 8;
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int yy = a_y[i] + dy, xx = a_x[i] + dx;
            const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            a_voff[i] = ok ? a_off[i] : kOOB;
        }
    };
    set_tap(tap0);
    auto issue = [&](int buf) {
        char* sA = smem + buf * kStage;                      // pixel tile  [BM][64] bf16
        char* sB = sA + BM * BK * 2;                         // weight tile [BN][64] bf16
        const int dy = (int)((g.ty4 >> (4 * ld_tap)) & 15u) - 8, dx = (int)((g.tx4 >> (4 * ld_tap)) & 15u) - 8;
        const uint32_t soff_a = (uint32_t)(dy * W + dx + g.back) * row_bytes + (uint32_t)ld_c * (BK * 2);
        const uint32_t soff_b = (uint32_t)((g.w4 >> (4 * ld_tap)) & 15u) * row_bytes + (uint32_t)ld_c * (BK * 2);
#if GD_CONV_ABLATE != 1 && GD_CONV_ABLATE != 2 && GD_CONV_ABLATE < 6
#pragma unroll
        for (int i = 0; i < NA; i++) bload_lds16(rs_in, a_voff[i], soff_a, sA + (wave * 64 + THREADS * i) * 16);
#endif
#if GD_CONV_ABLATE != 2 && GD_CONV_ABLATE < 6
#pragma unroll
        for (int i = 0; i < NB; i++) bload_lds16(rs_w, b_off[i], soff_b, sB + (wave * 64 + THREADS * i) * 16);
#endif
        if (++ld_c == kc) {
            ld_c = 0;
            if (++ld_tap < g.ntaps) set_tap(ld_tap);
        }
    };

    f32x16 acc[FA][FB];
#pragma unroll
    for (int a = 0; a < FA; a++)
#pragma unroll
        for (int b = 0; b < FB; b++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc[a][b][k] = 0.f;

    const int wc = wave % WN, wp = wave / WN;   // wave's channel / pixel block
    const int frow = lane & 31, fk = lane >> 5;
    // fragment read offsets for kk = 0; for kk > 0 the slot index is XORed with 2*kk (<< 4 bytes)
    uint32_t w_rd[FA], p_rd[FB];
#pragma unroll
    for (int a = 0; a < FA; a++) w_rd[a] = (uint32_t)(BM * BK * 2 + swz(wc * (BN / WN) + a * 32 + frow, fk));
#pragma unroll
    for (int b = 0; b < FB; b++) p_rd[b] = (uint32_t)swz(wp * (BM / WM) + b * 32 + frow, fk);

#if GD_CONV_ABLATE == 4 || GD_CONV_ABLATE >= 6
    bf16x8_t wf0[FA], pf0[FB];
#pragma unroll
    for (int a = 0; a < FA; a++) wf0[a] = *(const bf16x8_t*)(smem + w_rd[a]);
#pragma unroll
    for (int b = 0; b < FB; b++) pf0[b] = *(const bf16x8_t*)(smem + p_rd[b]);
#endif
    issue(0);
#if GD_CONV_ABLATE == 8
    for (int s = 0; s < (nsteps > 1 ? 1 : nsteps); s++) {
#else
    for (int s = 0; s < nsteps; s++) {
#endif
        const int buf = s & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if GD_CONV_ABLATE != 5 && GD_CONV_ABLATE != 6
        __syncthreads();                       // stage `buf` landed for everyone; stage buf^1 free
#endif
        if (s + 1 < nsteps) issue(buf ^ 1);
        const char* st = smem + buf * kStage;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            bf16x8_t wf[FA], pf[FB];
#if GD_CONV_ABLATE == 4 || GD_CONV_ABLATE >= 6
            if (s > 0) {
#pragma unroll
                for (int a = 0; a < FA; a++)
#pragma unroll
                    for (int b = 0; b < FB; b++)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf0[a], pf0[b], acc[a][b], 0, 0, 0);
                continue;
            }
#endif
#pragma unroll
            for (int a = 0; a < FA; a++) wf[a] = *(const bf16x8_t*)(st + (w_rd[a] ^ (uint32_t)(kk << 5)));
#pragma unroll
            for (int b = 0; b < FB; b++) pf[b] = *(const bf16x8_t*)(st + (p_rd[b] ^ (uint32_t)(kk << 5)));
#pragma unroll
            for (int a = 0; a < FA; a++)
#pragma unroll
                for (int b = 0; b < FB; b++)
#if GD_CONV_ABLATE == 3
                    if (a == 0 && b == 0) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[a], pf[b], acc[a][b], 0, 0, 0);
                    else { acc[a][b][0] += __builtin_bit_cast(float, (int)wf[a][0] ^ (int)pf[b][1]); }
#else
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[a], pf[b], acc[a][b], 0, 0, 0);
#endif
        }
    }

    // ---- epilogue: D[i = channel][j = pixel]; lane: pixel column lane&31, rows (reg&3)+8*(reg>>2)+4*(lane>>5)
#pragma unroll
    for (int b = 0; b < FB; b++) {
        const int64_t m = (int64_t)m0 + wp * (BM / WM) + b * 32 + (lane & 31);
        if (m >= M) continue;
        const int nimg = (int)(m / HW);
        const int rem = (int)(m - (int64_t)nimg * HW);
        const int ga = rem / g.Wg, gb = rem - ga * g.Wg;
        const size_t opix = ((size_t)nimg * g.Hout + (size_t)(ga * g.osy + g.ooy)) * g.Wout + (size_t)(gb * g.osx + g.oox);
        const uint16_t* bias_n = bias ? bias + (size_t)nimg * bias_img_stride : nullptr;
#pragma unroll
        for (int a = 0; a < FA; a++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int co = n0 + wc * (BN / WN) + a * 32 + 8 * q + 4 * fk;
                if (co >= Cout) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = acc[a][b][4 * q + e];
                if (partial) {
                    // fp32 partial of this K range in MFMA fragment order: every store instruction of a wave writes one
                    // contiguous 1 KB block of the tile's 64 KB image (conv_splitk_reduce_kernel decodes the same
                    // order).  Indexed by GEMM row / channel, each lane's 16 bytes fell in a line of their own:
                    // the scattered partial stores were HALF the time of the one-view-per-GPU convolutions
                    // (tools/conv_small_ablate.py: 640 -> 640 @ 32^2, 2 latents, 32 -> 17 us without them)
                    // (split launches always use the 128 x 128 / 4-wave tile: launch_conv)
                    *(float4*)(partial + (((size_t)split_idx * nwg + bid) * (BN * BM / 4) +
                                          (((wave * FB + b) * FA + a) * 4 + q) * 64 + lane) * 4) = make_float4(v[0], v[1], v[2], v[3]);
                    continue;
                }
                if (bias_n) {
                    const uint2 bb = *(const uint2*)(bias_n + co);
                    v[0] += bf2f((uint16_t)(bb.x & 0xffff)); v[1] += bf2f((uint16_t)(bb.x >> 16));
                    v[2] += bf2f((uint16_t)(bb.y & 0xffff)); v[3] += bf2f((uint16_t)(bb.y >> 16));
                }
                if (residual) {
                    const uint2 rr = *(const uint2*)(residual + opix * Cout + co);
                    v[0] += bf2f((uint16_t)(rr.x & 0xffff)); v[1] += bf2f((uint16_t)(rr.x >> 16));
                    v[2] += bf2f((uint16_t)(rr.y & 0xffff)); v[3] += bf2f((uint16_t)(rr.y >> 16));
                }
                uint2 o;
                o.x = pack_bf16(v[0], v[1]);
                o.y = pack_bf16(v[2], v[3]);
                *(uint2*)(out + opix * Cout + co) = o;
            }
        }
    }
}

template <int BN, int BM, int WN, int WM>
__global__ __launch_bounds__(64 * WN * WM) void conv3x3_nhwc_bf16_kernel(
    const uint16_t* __restrict__ in, const uint16_t* __restrict__ wt, const uint16_t* __restrict__ bias,
    int bias_img_stride, const uint16_t* __restrict__ residual, uint16_t* __restrict__ out, int Nimg, const ConvGeom g,
    int Cin, int Cout, int tiles_n, int nwg, float* __restrict__ partial, int steps_per_split)
{
    conv3x3_nhwc_bf16_body<BN, BM, WN, WM>(in, wt, bias, bias_img_stride, residual, out, Nimg, g, Cin, Cout, tiles_n, nwg,
                                           partial, steps_per_split, (int)blockIdx.x, (int)blockIdx.y);
}

// Up to four launches of the kernel above that differ only in their geometry (and weight tensor) as ONE launch:
// blockIdx.z picks the class.  The parity classes of a stride-2 input gradient (4 + 2 + 2 + 1 taps) and of the
// upsample-fused convolution (4 x 4 taps) were four back-to-back launches with K loops of 2-16 steps each -- four ramps,
// four tails, and the one-tap class alone cannot cover its own epilogue; together the classes fill each other's gaps.
struct ConvGeomSet {
    ConvGeom g[4];
    const uint16_t* wt[4];
    int nwg[4];
    int n;
};

template <int BN, int BM, int WN, int WM>
__global__ __launch_bounds__(64 * WN * WM) void conv3x3_nhwc_bf16_multi_kernel(
    const uint16_t* __restrict__ in, const uint16_t* __restrict__ bias, int bias_img_stride,
    const uint16_t* __restrict__ residual, uint16_t* __restrict__ out, int Nimg, const ConvGeomSet gs, int Cin, int Cout,
    int tiles_n)
{
    const int cls = (int)blockIdx.z;
    const int nwg = gs.nwg[cls];
    if ((int)blockIdx.x >= nwg) return;          // (workgroup-uniform) classes of ragged images differ by a tile row
    conv3x3_nhwc_bf16_body<BN, BM, WN, WM>(in, gs.wt[cls], bias, bias_img_stride, residual, out, Nimg, gs.g[cls], Cin, Cout,
                                           tiles_n, nwg, nullptr, 0, (int)blockIdx.x, 0);
}

// ------------------------------------------------------------------------------------------------
// Patch-staged 3x3 / stride 1 / pad 1 convolution with GroupNorm(+SiLU) fused into the activation
// loader:   y = conv3x3( act( GN(x) ) ) + bias (+ residual)
//
// The implicit-GEMM kernel above re-fetches every activation element nine times (once per tap) from L2
// straight into LDS, which leaves no place to transform it.  Here the M tile is a 16x16 SPATIAL patch of
// one image: per 64-channel K chunk its 18x18 halo'd input patch (41 KB) goes global -> registers ->
// (x * a[n,c] + b[n,c] -> SiLU -> bf16) -> LDS exactly once, and the nine taps read it at shifted
// addresses.  a = gamma * rstd, b = beta - mean * a come from the statistics pass (gd_nn_groupnorm_stats),
// so the normalised / activated tensor -- a full read + write of the activation in the unfused form --
// never exists in HBM, and the L2 -> LDS activation traffic of the convolution drops ~7x.  Zero padding
// is applied after the transform (a halo pixel outside the image is 0, not SiLU(b)).  Weights stream per
// (tap, chunk) through the same swizzled LDS-DMA path as above.
//
// LDS: activation patch [324 px][64 ch] bf16, pixel p's 16-B chunk j at p*128 + ((j ^ ((p>>1)&7)) << 4);
// MFMA pixel columns are assigned so that each ds_read_b128 service group (16 lanes) reads 16 CONSECUTIVE
// patch pixels of one row: their (p mod 16) are distinct whatever the tap shift, so every read is
// conflict free (MI355X_MICROARCH.md, LDS table: groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...).
constexpr int kPatch = 18, kPatchPix = kPatch * kPatch;   // 16x16 tile + 1-pixel halo

__device__ __forceinline__ float silu_fast(float z) { return z * __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }

// Sum over the 16 lanes of a DPP row, result in every lane (quad swaps, half-row mirror, row mirror).
__device__ __forceinline__ float row16_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));
    return v;
}

// GroupNorm statistics of the OUTPUT from the epilogue (stat_part != NULL): the next layer of a ResnetBlock2D chain
// is a GroupNorm over this very tensor, and its statistics pass is a full re-read of it from HBM (the 512^2 level of
// the VAE encoder: 537 MB, 100 us per call at 8 views).  Each lane sums the bf16-ROUNDED values it stores (what the
// consumer will read) and their squares over its 2 pixels x 4 consecutive channels (one group: needs
// Cout / G % 4 == 0), a DPP row reduction folds 16 pixels, and one lane per row writes the pair: per (image,
// channel quad) tiles_per_image * 8 fp32 partials, summed in fp64 by gd_nn_groupnorm_finish_partials.  No atomics,
// so the statistics are deterministic; every row is written by every launch (edge tiles write what they have).
__device__ __forceinline__ void stat_accumulate(float2& st, uint2 o)
{
    const float r0 = __uint_as_float(o.x << 16), r1 = __uint_as_float(o.x & 0xffff0000u);
    const float r2 = __uint_as_float(o.y << 16), r3 = __uint_as_float(o.y & 0xffff0000u);
    st.x += (r0 + r1) + (r2 + r3);
    st.y += fmaf(r0, r0, fmaf(r1, r1, fmaf(r2, r2, r3 * r3)));
}

// MFMA pixel column fn -> (row rr in {0,1}, x) of the wave's 2 x 16 pixel block, such that each 16-lane LDS service
// group reads one patch row
__device__ __forceinline__ void patch_col(int fn, int& rr, int& fx)
{
    if (fn < 4) { rr = 0; fx = fn; }
    else if (fn < 12) { rr = 1; fx = fn - 4; }
    else if (fn < 16) { rr = 0; fx = fn - 8; }
    else if (fn < 20) { rr = 1; fx = fn - 8; }
    else if (fn < 28) { rr = 0; fx = fn - 12; }
    else { rr = 1; fx = fn - 16; }
}

constexpr int kTrRow = 80;                 // bytes per pixel row of the epilogue's transposition buffer (64 + pad)
constexpr int kTrWave = 64 * kTrRow;       // per wave: 64 pixels x 32 channels

// Epilogue of the patch-staged kernels: out = bf16(acc + bias (+ residual)), optional GroupNorm partial sums.
// In the MFMA result layout a lane owns 4 consecutive channels of a pixel and neighbouring lanes are neighbouring
// PIXELS, so a direct store instruction is 64 separate 8-byte writes (one per cache line): with those the stores cost
// 6-23 % of the VAE encoder's convolutions (GD_PATCH_EPI=1 timing, tools/nn_variants.sh).  With `tr` (a wave-private
// kTrWave bytes of LDS) and Cout % 8 == 0 the packed values of one 32-channel block go through LDS and leave as
// 16 bytes per lane, four lanes per pixel: 64 contiguous bytes per pixel and instruction.  Loop order (channel quad,
// pixel): the partial sums of one quad live in two registers.
template <int BN, int WN, int FA, int FB>
__device__ __forceinline__ void patch_epilogue(f32x16 (&acc)[FA][FB], char* tr, int nimg, int tyi, int txi, int n0, int wc,
                                               int wp, int lane, int H, int W, int Cout, const uint16_t* __restrict__ bias,
                                               int bias_img_stride, const uint16_t* __restrict__ residual,
                                               uint16_t* __restrict__ out, float* __restrict__ stat_part, int tpi,
                                               int tiles_x)
{
    const int fk = lane >> 5, fn = lane & 31;
    int rr, fx;
    patch_col(fn, rr, fx);
    bool ok[FB];
    size_t opix[FB];
#pragma unroll
    for (int b = 0; b < FB; b++) {
        const int oy = tyi * 16 + 4 * wp + 2 * b + rr, ox = txi * 16 + fx;
        ok[b] = oy < H && ox < W;
        opix[b] = ((size_t)nimg * H + oy) * W + ox;
    }
    const uint16_t* bias_n = bias ? bias + (size_t)nimg * bias_img_stride : nullptr;
    const int stat_rows = tpi * 8;
    const int stat_row = (tyi * tiles_x + txi) * 8 + wp * 2 + ((lane >> 4) & 1);
    const bool wide = GD_PATCH_EPI != 2 && tr != nullptr && (Cout & 7) == 0;
#pragma unroll
    for (int a = 0; a < FA; a++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int co = n0 + wc * (BN / WN) + a * 32 + 8 * q + 4 * fk;
            const bool cok = co < Cout;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias_n && cok) {
                const uint2 bb = *(const uint2*)(bias_n + co);
                bv[0] = bf2f((uint16_t)(bb.x & 0xffff)); bv[1] = bf2f((uint16_t)(bb.x >> 16));
                bv[2] = bf2f((uint16_t)(bb.y & 0xffff)); bv[3] = bf2f((uint16_t)(bb.y >> 16));
            }
            float2 st = make_float2(0.f, 0.f);   // sum, sum of squares of the stored values (stat_part)
#pragma unroll
            for (int b = 0; b < FB; b++) {
                if (!ok[b] || !cok) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = acc[a][b][4 * q + e] + bv[e];
                if (residual) {
                    const uint2 rv = *(const uint2*)(residual + opix[b] * Cout + co);
                    v[0] += bf2f((uint16_t)(rv.x & 0xffff)); v[1] += bf2f((uint16_t)(rv.x >> 16));
                    v[2] += bf2f((uint16_t)(rv.y & 0xffff)); v[3] += bf2f((uint16_t)(rv.y >> 16));
                }
                uint2 o;
                o.x = pack_bf16(v[0], v[1]);
                o.y = pack_bf16(v[2], v[3]);
                if (wide) *(uint2*)(tr + (b * 32 + fn) * kTrRow + 16 * q + 8 * fk) = o;
                else if (GD_PATCH_EPI != 1 || H < 0) *(uint2*)(out + opix[b] * Cout + co) = o;
                if (stat_part) stat_accumulate(st, o);
            }
            if (stat_part) {
                const float sx = row16_sum(st.x), sy = row16_sum(st.y);
                if ((lane & 15) == 0 && cok)
                    *(float2*)(stat_part + (((size_t)nimg * (Cout >> 2) + (co >> 2)) * stat_rows + stat_row) * 2) =
                        make_float2(sx, sy);
            }
        }
        if (wide) {
            // LDS operations of one wave execute in order: the reads below see the writes above
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < FB * 2; i++) {
                const int pr = (lane >> 2) + 16 * i, ch = lane & 3;     // pixel of the wave's block, 8-channel chunk
                int rr2, fx2;
                patch_col(pr & 31, rr2, fx2);
                const int oy = tyi * 16 + 4 * wp + 2 * (pr >> 5) + rr2, ox = txi * 16 + fx2;
                const int co8 = n0 + wc * (BN / WN) + a * 32 + 8 * ch;
                if (oy < H && ox < W && co8 < Cout) {
                    const uint4 v = *(const uint4*)(tr + pr * kTrRow + ch * 16);
                    if (GD_PATCH_EPI != 1 || H < 0) *(uint4*)(out + (((size_t)nimg * H + oy) * W + ox) * Cout + co8) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// GN = true: GroupNorm(+SiLU) applied in the loader (register staged).  GN = false: plain convolution, the patch
// is fetched by LDS-DMA like the weights -- 41 LDS-DMA wave-instructions per 64 input channels (5.2 patch + 36
// weight pieces) instead of the implicit-GEMM kernel's 72.
template <int BN, int WN, int WM, bool GN>
__global__ __launch_bounds__(64 * WN * WM) void conv3x3_gn_patch_kernel(
    const uint16_t* __restrict__ in, const uint16_t* __restrict__ wt, const uint16_t* __restrict__ bias,
    int bias_img_stride, const uint16_t* __restrict__ residual, uint16_t* __restrict__ out, int Nimg, int H, int W,
    int Cin, int Cout, const float* __restrict__ mean_rstd, const uint16_t* __restrict__ gamma,
    const uint16_t* __restrict__ beta, int G, int apply_silu, int tiles_n, int tiles_x, int tiles_y, int nwg,
    float* __restrict__ stat_part)
{
    constexpr int BM = 256;
    constexpr int THREADS = 64 * WN * WM;
    constexpr int NB = BN * 8 / THREADS;                       // 16-B chunks of the weight tile per thread per step
    constexpr int NA = (kPatchPix * 8 + THREADS - 1) / THREADS;  // ... of the activation patch per K chunk
    constexpr int FA = BN / WN / 32, FB = BM / WM / 32;
    static_assert(FB == 2 && THREADS % 8 == 0, "wave pixel block = 4 patch rows x 16");
    constexpr int kAStage = (kPatchPix + 4) * BK * 2;          // 41984 B: 324 pixels + 4 of padding so that the last
                                                               // (half-filled) LDS-DMA piece of wave 0 stays inside
    constexpr int kBStage = BN * BK * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                    // 2 stages
    char* sB = smem + 2 * kAStage;      // 2 stages

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int tn = bid % tiles_n, tm = bid / tiles_n;
    const int tpi = tiles_x * tiles_y;
    const int nimg = tm / tpi, trem = tm - nimg * tpi;
    const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;
    const int y0 = tyi * 16 - 1, x0 = txi * 16 - 1;           // image coords of patch pixel (0, 0)
    const int n0 = tn * BN;

    const uint32_t row_bytes = (uint32_t)Cin * 2u;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        (void*)in, 0, (int)((uint32_t)Nimg * (uint32_t)(H * W) * row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)wt, 0, (int)((uint32_t)Cout * 9u * row_bytes), 0x00020000);

    // ---- activation loader: thread <-> fixed channel octet (tid & 7), NA patch pixels
    const int a_chunk = tid & 7;
    uint32_t a_goff[NA];      // byte offset of (pixel, octet) in `in`, or kOOB (outside image / beyond the patch)
    uint32_t a_lds[NA];       // byte offset inside a patch stage
    uint32_t a_keep = 0;      // bit i: slot i is a real patch pixel inside the image (else it must stay zero)
    uint32_t a_slot = 0;      // bit i: slot i exists (pix < 324)
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int pix = (tid + THREADS * i) >> 3;
        const int py = pix / kPatch, px = pix - py * kPatch;
        const int gy = y0 + py, gx = x0 + px;
        const bool slot = pix < kPatchPix;
        const bool inimg = slot && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        // GN: this thread always handles channel octet tid & 7 and swizzles the LDS address; DMA: the LDS slot is
        // lane-linear, so the swizzle picks WHICH octet the lane fetches
        const int oct = GN ? a_chunk : (a_chunk ^ ((pix >> 1) & 7));
        a_goff[i] = inimg ? (uint32_t)((nimg * H + gy) * W + gx) * row_bytes + (uint32_t)oct * 16u : kOOB;
        a_lds[i] = (uint32_t)pix * 128u + (uint32_t)((a_chunk ^ ((pix >> 1) & 7)) << 4);
        if (inimg) a_keep |= 1u << i;
        if (slot) a_slot |= 1u << i;
    }
    uint4 a_reg[NA];
    auto loadA = [&](int c) {
#pragma unroll
        for (int i = 0; i < NA; i++) {
            auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)a_goff[i], (int)(c * (BK * 2)), 0);
            a_reg[i] = __builtin_bit_cast(uint4, v);
        }
    };
    const int cg = mean_rstd ? Cin / G : 1;
    auto storeA = [&](int buf, int c) {
        char* dst = sA + buf * kAStage;
        float sc[8], sh[8];
        if (mean_rstd) {
            const int ch0 = c * BK + a_chunk * 8;
            const uint4 gq = *(const uint4*)(gamma + ch0), bq = *(const uint4*)(beta + ch0);
            const uint32_t gw[4] = {gq.x, gq.y, gq.z, gq.w}, bw[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int g = (ch0 + k) / cg;
                const float2 mr = *(const float2*)(mean_rstd + ((size_t)nimg * G + g) * 2);
                const float gm = bf2f((uint16_t)(gw[k >> 1] >> ((k & 1) * 16)));
                const float bt = bf2f((uint16_t)(bw[k >> 1] >> ((k & 1) * 16)));
                sc[k] = gm * mr.y;
                sh[k] = bt - mr.x * sc[k];
            }
        }
#pragma unroll
        for (int i = 0; i < NA; i++) {
            if (!((a_slot >> i) & 1u)) continue;
            uint4 v = a_reg[i];
            if (mean_rstd) {
                uint32_t w4[4] = {v.x, v.y, v.z, v.w};
                const bool keep = (a_keep >> i) & 1u;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float lo = __uint_as_float(w4[k] << 16), hi = __uint_as_float(w4[k] & 0xffff0000u);
                    lo = lo * sc[2 * k] + sh[2 * k];
                    hi = hi * sc[2 * k + 1] + sh[2 * k + 1];
                    if (apply_silu) { lo = silu_fast(lo); hi = silu_fast(hi); }
                    w4[k] = keep ? pack_bf16(lo, hi) : 0u;
                }
                v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            }
            *(uint4*)(dst + a_lds[i]) = v;
        }
    };

    // ---- weight loader (LDS-DMA, swizzled on the source side; as in the implicit-GEMM kernel)
    uint32_t b_off[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) {
        const int q = tid + THREADS * i;
        const int line = q >> 4, cc = (q & 15) ^ (line & 15);
        const int r = 2 * line + (cc >> 3);
        const int co = n0 + r;
        b_off[i] = co < Cout ? (uint32_t)co * 9u * row_bytes + (uint32_t)(cc & 7) * 16u : kOOB;
    }
    auto issueB = [&](int buf, int tap, int c) {
        char* dst = sB + buf * kBStage;
        const uint32_t soff = (uint32_t)tap * row_bytes + (uint32_t)c * (BK * 2);
#pragma unroll
        for (int i = 0; i < NB; i++) bload_lds16(rs_w, b_off[i], soff, dst + (wave * 64 + THREADS * i) * 16);
    };

    f32x16 acc[FA][FB];
#pragma unroll
    for (int a = 0; a < FA; a++)
#pragma unroll
        for (int b = 0; b < FB; b++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc[a][b][k] = 0.f;

    const int wc = wave % WN, wp = wave / WN;   // wave's channel block / pixel block (4 patch rows)
    const int fk = lane >> 5, fn = lane & 31;
    // MFMA pixel column fn -> (row rr in {0,1}, x) such that each 16-lane LDS service group is one row
    int rr, fx;
    if (fn < 4) { rr = 0; fx = fn; }
    else if (fn < 12) { rr = 1; fx = fn - 4; }
    else if (fn < 16) { rr = 0; fx = fn - 8; }
    else if (fn < 20) { rr = 1; fx = fn - 8; }
    else if (fn < 28) { rr = 0; fx = fn - 12; }
    else { rr = 1; fx = fn - 16; }
    uint32_t w_rd[FA];
    int p_base[FB];           // patch pixel index of the lane's pixel for tap (0, 0)
#pragma unroll
    for (int a = 0; a < FA; a++) w_rd[a] = (uint32_t)swz(wc * (BN / WN) + a * 32 + fn, fk);
#pragma unroll
    for (int b = 0; b < FB; b++) p_base[b] = (4 * wp + 2 * b + rr) * kPatch + fx;

    const int kc = Cin / BK;
    const int nsteps = 9 * kc;
    // prologue: patch of chunk 0 and weights of step 0
    auto issueA = [&](int buf, int c) {     // GN = false: LDS-DMA of the patch; piece NA-1 only has pixels in wave 0
        char* dst = sA + buf * kAStage;
#pragma unroll
        for (int i = 0; i < NA; i++) {
            if (i == NA - 1 && wave * 64 + THREADS * i >= kPatchPix * 8) continue;
            bload_lds16(rs_in, a_goff[i], (uint32_t)c * (BK * 2), dst + (wave * 64 + THREADS * i) * 16);
        }
    };
    if (GN) {
        loadA(0);
        issueB(0, 0, 0);
        storeA(0, 0);
    } else {
        issueA(0, 0);
        issueB(0, 0, 0);
    }
    int s = 0;
    for (int c = 0; c < kc; c++) {
        const char* pa = sA + (c & 1) * kAStage;
        for (int tap = 0; tap < 9; tap++, s++) {
            const int bufB = s & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (s + 1 < nsteps) issueB(bufB ^ 1, tap == 8 ? 0 : tap + 1, tap == 8 ? c + 1 : c);
            if (c + 1 < kc) {
                if (GN) {
                    if (tap == 0) loadA(c + 1);
                    else if (tap == 1) storeA((c + 1) & 1, c + 1);   // (staggering the store per wave pair: slower, tried)
                } else if (tap == 0) {
                    issueA((c + 1) & 1, c + 1);   // the other patch stage was last read in chunk c - 1
                }
            }
            const char* pb = sB + bufB * kBStage;
            const int tapoff = (tap / 3) * kPatch + (tap % 3);
            uint32_t p_rd[FB], p_sw[FB];
#pragma unroll
            for (int b = 0; b < FB; b++) {
                const int p = p_base[b] + tapoff;
                p_rd[b] = (uint32_t)p * 128u;
                p_sw[b] = (uint32_t)((p >> 1) & 7);
            }
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                bf16x8_t wf[FA], pf[FB];
#pragma unroll
                for (int a = 0; a < FA; a++) wf[a] = *(const bf16x8_t*)(pb + (w_rd[a] ^ (uint32_t)(kk << 5)));
#pragma unroll
                for (int b = 0; b < FB; b++)
                    pf[b] = *(const bf16x8_t*)(pa + p_rd[b] + ((((uint32_t)(2 * kk + fk)) ^ p_sw[b]) << 4));
#pragma unroll
                for (int a = 0; a < FA; a++)
#pragma unroll
                    for (int b = 0; b < FB; b++)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[a], pf[b], acc[a][b], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: every wave is done with the patch stage of the last chunk once all have passed this barrier; its
    // memory becomes the transposition buffer
    if (GD_PATCH_EPI != 2) __syncthreads();
    patch_epilogue<BN, WN, FA, FB>(acc, sA + ((kc - 1) & 1) * kAStage + wave * kTrWave, nimg, tyi, txi, n0, wc, wp, lane, H, W,
                                   Cout, bias, bias_img_stride, residual, out, stat_part, tpi, tiles_x);
}

// The same patch-staged convolution as a PERSISTENT kernel for the plain (GN = false) case; the GroupNorm variants
// above keep one tile per workgroup (their register-staged transform leaves no registers for the cross-tile
// prefetch: restructured the same way they ran 2-3 % slower).
template <int BN, int WN, int WM, bool GN>
__global__ __launch_bounds__(64 * WN * WM) void conv3x3_patch_stream_kernel(
    const uint16_t* __restrict__ in, const uint16_t* __restrict__ wt, const uint16_t* __restrict__ bias,
    int bias_img_stride, const uint16_t* __restrict__ residual, uint16_t* __restrict__ out, int Nimg, int H, int W,
    int Cin, int Cout, const float* __restrict__ mean_rstd, const uint16_t* __restrict__ gamma,
    const uint16_t* __restrict__ beta, int G, int apply_silu, int tiles_n, int tiles_x, int tiles_y, int nwg,
    float* __restrict__ stat_part)
{
    constexpr int BM = 256;
    constexpr int THREADS = 64 * WN * WM;
    constexpr int NB = BN * 8 / THREADS;                       // 16-B chunks of the weight tile per thread per step
    constexpr int NA = (kPatchPix * 8 + THREADS - 1) / THREADS;  // ... of the activation patch per K chunk
    constexpr int FA = BN / WN / 32, FB = BM / WM / 32;
    static_assert(FB == 2 && THREADS % 8 == 0, "wave pixel block = 4 patch rows x 16");
    constexpr int kAStage = (kPatchPix + 4) * BK * 2;          // 41984 B: 324 pixels + 4 of padding so that the last
                                                               // (half-filled) LDS-DMA piece of wave 0 stays inside
    constexpr int kBStage = BN * BK * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                    // 2 stages
    char* sB = smem + 2 * kAStage;      // 2 stages

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // PERSISTENT workgroups (plain variant, GN = false): workgroup b runs tiles b, b + gridDim.x, ... as one
    // continuous stream of (chunk, tap) steps -- the patch of the next tile's first chunk and its first weight slice
    // are fetched during the last chunk of the current tile, so the load prologue (half of all patch traffic when
    // Cin = 128) and the epilogue stores overlap with MFMA work instead of standing at the head and tail of every
    // workgroup: +3...5 % on the plain convolutions (tools/patch_conv_bench.py).
    const int tpi = tiles_x * tiles_y;
    const bool remap = (nwg & 7) == 0 && (GN || (gridDim.x & 7) == 0);   // XCD-contiguous tile order
    struct Tile { int nimg, tyi, txi, n0; };
    auto decode = [&](int vb) {
        int bid = vb;
        if (remap) bid = (vb & 7) * (nwg >> 3) + (vb >> 3);
        const int tn = bid % tiles_n, tm = bid / tiles_n;
        Tile t;
        t.nimg = tm / tpi;
        const int trem = tm - t.nimg * tpi;
        t.tyi = trem / tiles_x;
        t.txi = trem - t.tyi * tiles_x;
        t.n0 = tn * BN;
        return t;
    };

    const uint32_t row_bytes = (uint32_t)Cin * 2u;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        (void*)in, 0, (int)((uint32_t)Nimg * (uint32_t)(H * W) * row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)wt, 0, (int)((uint32_t)Cout * 9u * row_bytes), 0x00020000);

    // ---- activation loader: thread <-> fixed channel octet (tid & 7), NA patch pixels
    const int a_chunk = tid & 7;
    uint32_t a_goff[NA];      // byte offset of (pixel, octet) in `in`, or kOOB (outside image / beyond the patch)
    uint32_t a_lds[NA];       // byte offset inside a patch stage
    uint32_t a_keep = 0;      // bit i: slot i is a real patch pixel inside the image (else it must stay zero)
    uint32_t a_slot = 0;      // bit i: slot i exists (pix < 324)
    int ld_nimg = 0;          // image of the tile whose patch is being loaded (its GroupNorm statistics row)
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int pix = (tid + THREADS * i) >> 3;
        a_lds[i] = (uint32_t)pix * 128u + (uint32_t)((a_chunk ^ ((pix >> 1) & 7)) << 4);
        if (pix < kPatchPix) a_slot |= 1u << i;
    }
    // loader state of the tile whose patch chunks are fetched next
    auto setupA = [&](const Tile& t) {
        const int y0 = t.tyi * 16 - 1, x0 = t.txi * 16 - 1;   // image coords of patch pixel (0, 0)
        a_keep = 0;
        ld_nimg = t.nimg;
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int pix = (tid + THREADS * i) >> 3;
            const int py = pix / kPatch, px = pix - py * kPatch;
            const int gy = y0 + py, gx = x0 + px;
            const bool inimg = pix < kPatchPix && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            // GN: this thread always handles channel octet tid & 7 and swizzles the LDS address; DMA: the LDS slot
            // is lane-linear, so the swizzle picks WHICH octet the lane fetches
            const int oct = GN ? a_chunk : (a_chunk ^ ((pix >> 1) & 7));
            a_goff[i] = inimg ? (uint32_t)((t.nimg * H + gy) * W + gx) * row_bytes + (uint32_t)oct * 16u : kOOB;
            if (inimg) a_keep |= 1u << i;
        }
    };
    uint4 a_reg[NA];
    auto loadA = [&](int c) {
#pragma unroll
        for (int i = 0; i < NA; i++) {
            auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)a_goff[i], (int)(c * (BK * 2)), 0);
            a_reg[i] = __builtin_bit_cast(uint4, v);
        }
    };
    const int cg = mean_rstd ? Cin / G : 1;
    auto storeA = [&](int buf, int c) {
        char* dst = sA + buf * kAStage;
        float sc[8], sh[8];
        if (mean_rstd) {
            const int ch0 = c * BK + a_chunk * 8;
            const uint4 gq = *(const uint4*)(gamma + ch0), bq = *(const uint4*)(beta + ch0);
            const uint32_t gw[4] = {gq.x, gq.y, gq.z, gq.w}, bw[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int g = (ch0 + k) / cg;
                const float2 mr = *(const float2*)(mean_rstd + ((size_t)ld_nimg * G + g) * 2);
                const float gm = bf2f((uint16_t)(gw[k >> 1] >> ((k & 1) * 16)));
                const float bt = bf2f((uint16_t)(bw[k >> 1] >> ((k & 1) * 16)));
                sc[k] = gm * mr.y;
                sh[k] = bt - mr.x * sc[k];
            }
        }
#pragma unroll
        for (int i = 0; i < NA; i++) {
            if (!((a_slot >> i) & 1u)) continue;
            uint4 v = a_reg[i];
            if (mean_rstd) {
                uint32_t w4[4] = {v.x, v.y, v.z, v.w};
                const bool keep = (a_keep >> i) & 1u;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float lo = __uint_as_float(w4[k] << 16), hi = __uint_as_float(w4[k] & 0xffff0000u);
                    lo = lo * sc[2 * k] + sh[2 * k];
                    hi = hi * sc[2 * k + 1] + sh[2 * k + 1];
                    if (apply_silu) { lo = silu_fast(lo); hi = silu_fast(hi); }
                    w4[k] = keep ? pack_bf16(lo, hi) : 0u;
                }
                v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            }
            *(uint4*)(dst + a_lds[i]) = v;
        }
    };

    // ---- weight loader (LDS-DMA, swizzled on the source side; as in the implicit-GEMM kernel)
    uint32_t b_off[NB];
    auto setupB = [&](const Tile& t) {
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int q = tid + THREADS * i;
            const int line = q >> 4, cc = (q & 15) ^ (line & 15);
            const int r = 2 * line + (cc >> 3);
            const int co = t.n0 + r;
            b_off[i] = co < Cout ? (uint32_t)co * 9u * row_bytes + (uint32_t)(cc & 7) * 16u : kOOB;
        }
    };
    auto issueB = [&](int buf, int tap, int c) {
        char* dst = sB + buf * kBStage;
        const uint32_t soff = (uint32_t)tap * row_bytes + (uint32_t)c * (BK * 2);
#pragma unroll
        for (int i = 0; i < NB; i++) bload_lds16(rs_w, b_off[i], soff, dst + (wave * 64 + THREADS * i) * 16);
    };

    f32x16 acc[FA][FB];

    const int wc = wave % WN, wp = wave / WN;   // wave's channel block / pixel block (4 patch rows)
    const int fk = lane >> 5, fn = lane & 31;
    // MFMA pixel column fn -> (row rr in {0,1}, x) such that each 16-lane LDS service group is one row
    int rr, fx;
    if (fn < 4) { rr = 0; fx = fn; }
    else if (fn < 12) { rr = 1; fx = fn - 4; }
    else if (fn < 16) { rr = 0; fx = fn - 8; }
    else if (fn < 20) { rr = 1; fx = fn - 8; }
    else if (fn < 28) { rr = 0; fx = fn - 12; }
    else { rr = 1; fx = fn - 16; }
    uint32_t w_rd[FA];
    int p_base[FB];           // patch pixel index of the lane's pixel for tap (0, 0)
#pragma unroll
    for (int a = 0; a < FA; a++) w_rd[a] = (uint32_t)swz(wc * (BN / WN) + a * 32 + fn, fk);
#pragma unroll
    for (int b = 0; b < FB; b++) p_base[b] = (4 * wp + 2 * b + rr) * kPatch + fx;

    const int kc = Cin / BK;
    // prologue: patch of chunk 0 and weights of step 0
    auto issueA = [&](int buf, int c) {     // GN = false: LDS-DMA of the patch; piece NA-1 only has pixels in wave 0
        char* dst = sA + buf * kAStage;
#pragma unroll
        for (int i = 0; i < NA; i++) {
            if (i == NA - 1 && wave * 64 + THREADS * i >= kPatchPix * 8) continue;
            bload_lds16(rs_in, a_goff[i], (uint32_t)c * (BK * 2), dst + (wave * 64 + THREADS * i) * 16);
        }
    };
    int vb = blockIdx.x;
    Tile cur = decode(vb);
    setupB(cur);
    setupA(cur);
#pragma unroll
    for (int a = 0; a < FA; a++)
#pragma unroll
        for (int b = 0; b < FB; b++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc[a][b][k] = 0.f;
    if (GN) {
        loadA(0);
        issueB(0, 0, 0);
        storeA(0, 0);
    } else {
        issueA(0, 0);
        issueB(0, 0, 0);
    }
    int s = 0;    // running (chunk, tap) step count: weight buffer parity
    int ga = 0;   // running chunk count: patch stage parity
    for (;;) {
        const int vnext = vb + (int)gridDim.x;
        // the GroupNorm variants keep one tile per workgroup: with the register-staged transform the cross-tile
        // prefetch costs more registers (spills in the 256-channel variant) than the hidden prologue returns
        const bool has_next = !GN && vnext < nwg;
        const Tile nxt = decode(has_next ? vnext : vb);
        for (int c = 0; c < kc; c++, ga++) {
            const char* pa = sA + (ga & 1) * kAStage;
            const bool last_chunk = c + 1 == kc;
            // every patch chunk of the current tile is on its way or in LDS: the loader moves on to the next tile
            if (last_chunk && has_next) setupA(nxt);
            for (int tap = 0; tap < 9; tap++, s++) {
                const int bufB = s & 1;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (!(last_chunk && tap == 8)) {
                    issueB(bufB ^ 1, tap == 8 ? 0 : tap + 1, tap == 8 ? c + 1 : c);
                } else if (has_next) {
                    setupB(nxt);
                    issueB(bufB ^ 1, 0, 0);
                }
                if (!last_chunk || has_next) {
                    const int nc = last_chunk ? 0 : c + 1;
                    const int st = (ga + 1) & 1;   // that patch stage was last read one chunk ago
                    if (GN) {
                        if (tap == 0) loadA(nc);
                        else if (tap == 1) storeA(st, nc);   // (staggering the store per wave pair: slower, tried)
                    } else if (tap == 0) {
                        issueA(st, nc);
                    }
                }
                const char* pb = sB + bufB * kBStage;
                const int tapoff = (tap / 3) * kPatch + (tap % 3);
                uint32_t p_rd[FB], p_sw[FB];
#pragma unroll
                for (int b = 0; b < FB; b++) {
                    const int p = p_base[b] + tapoff;
                    p_rd[b] = (uint32_t)p * 128u;
                    p_sw[b] = (uint32_t)((p >> 1) & 7);
                }
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    bf16x8_t wf[FA], pf[FB];
#pragma unroll
                    for (int a = 0; a < FA; a++) wf[a] = *(const bf16x8_t*)(pb + (w_rd[a] ^ (uint32_t)(kk << 5)));
#pragma unroll
                    for (int b = 0; b < FB; b++)
                        pf[b] = *(const bf16x8_t*)(pa + p_rd[b] + ((((uint32_t)(2 * kk + fk)) ^ p_sw[b]) << 4));
#pragma unroll
                    for (int a = 0; a < FA; a++)
#pragma unroll
                        for (int b = 0; b < FB; b++)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[a], pf[b], acc[a][b], 0, 0, 0);
                }
            }
        }

        // ---- epilogue of the current tile (the next tile's first patch chunk and weights are already in flight, into
        // the OTHER patch stage: the stage of the last chunk is free once every wave has passed this barrier, and stays
        // free until the first step of the next tile issues its second chunk -- behind that step's barrier)
        if (GD_PATCH_EPI != 2) __syncthreads();
        patch_epilogue<BN, WN, FA, FB>(acc, sA + ((ga - 1) & 1) * kAStage + wave * kTrWave, cur.nimg, cur.tyi, cur.txi, cur.n0,
                                       wc, wp, lane, H, W, Cout, bias, bias_img_stride, residual, out, stat_part, tpi,
                                       tiles_x);
        if (!has_next) break;
        cur = nxt;
        vb = vnext;
#pragma unroll
        for (int a = 0; a < FA; a++)
#pragma unroll
            for (int b = 0; b < FB; b++)
#pragma unroll
                for (int k = 0; k < 16; k++) acc[a][b][k] = 0.f;
    }
}

#include "nn_conv_wino.h"
#include "nn_conv_wide.h"
#include "nn_conv_first_dgrad.h"
#ifdef GD_NN_EXPERIMENTAL_STREAM   // -Itools/experimental: the weight-streaming small-map experiment, built by tools/ only
#include "nn_conv_stream.h"
#endif
#ifdef GD_NN_EXPERIMENTAL_REGW   // -Itools/experimental: the filter-bank-in-registers experiment, built by tools/ only
#include "nn_conv_regw.h"
#endif

// out = bf16( sum_s partial[s] + bias + residual ): second half of the split-K path, 4 channels per thread, splits
// added in index order (deterministic).  Partials are tile images in the convolution's MFMA fragment order (coalesced
// 16-byte accesses on both sides); GEMM row -> output pixel is the convolution's map (identity for stride-1 layers,
// every second pixel for the parity classes of a stride-2 input gradient).
template <int BM_T>
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const float* __restrict__ partial, int S, int nwg,
                                                                 int tiles_n, int64_t M, int Cout, int HWg, int Wg, int Hout,
                                                                 int Wout, int osy, int osx, int ooy, int oox,
                                                                 const uint16_t* __restrict__ bias, int bias_img_stride,
                                                                 const uint16_t* __restrict__ residual,
                                                                 uint16_t* __restrict__ out)
{
    // one thread = one float4 of a 128-channel x BM_T-pixel tile image in the convolution's fragment order:
    //   r = (((wave * 2 + b) * 2 + a) * 4 + q) * 64 + lane   (wave = pixel block of 64 * 2 + channel half of the tile;
    //   4 waves for the 128 x 128 tile, 8 for 128 channels x 256 pixels)
    constexpr int kQuads = 128 * BM_T / 4, kBlocks = kQuads / 256;
    const int tile = blockIdx.x / kBlocks;
    const int r = ((blockIdx.x % kBlocks) << 8) | threadIdx.x;
    const int lane = r & 63, q = (r >> 6) & 3, a = (r >> 8) & 1, b = (r >> 9) & 1, wave = r >> 10;
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    const int64_t m = (int64_t)tm * BM_T + (wave >> 1) * 64 + b * 32 + (lane & 31);
    const int co = tn * 128 + (wave & 1) * 64 + a * 32 + 8 * q + 4 * (lane >> 5);
    if (m >= M || co >= Cout) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* p = (const float4*)partial + (size_t)tile * kQuads + r;
    for (int sidx = 0; sidx < S; sidx++) {
        const float4 t = p[(size_t)sidx * nwg * kQuads];
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const size_t nimg = (size_t)(m / HWg);
    const int rem = (int)(m - (int64_t)nimg * HWg);
    const int ga = rem / Wg, gb = rem - ga * Wg;
    const size_t o = ((nimg * Hout + (size_t)(ga * osy + ooy)) * Wout + (size_t)(gb * osx + oox)) * Cout + co;
    if (bias) {
        const uint2 bb = *(const uint2*)(bias + nimg * (size_t)bias_img_stride + co);
        v.x += bf2f((uint16_t)(bb.x & 0xffff)); v.y += bf2f((uint16_t)(bb.x >> 16));
        v.z += bf2f((uint16_t)(bb.y & 0xffff)); v.w += bf2f((uint16_t)(bb.y >> 16));
    }
    if (residual) {
        const uint2 rr = *(const uint2*)(residual + o);
        v.x += bf2f((uint16_t)(rr.x & 0xffff)); v.y += bf2f((uint16_t)(rr.x >> 16));
        v.z += bf2f((uint16_t)(rr.y & 0xffff)); v.w += bf2f((uint16_t)(rr.y >> 16));
    }
    uint2 ov;
    ov.x = pack_bf16(v.x, v.y);
    ov.y = pack_bf16(v.z, v.w);
    *(uint2*)(out + o) = ov;
}

// First convolution of the VAE encoder / UNet: Cin <= 4 (image or latent -> features), 3x3 / s1 / p1, + bias.
// K = 9 * Cin <= 36 is far too short for the MFMA pipeline to matter; the layer is an output-write stream
// (N*H*W*Cout*2 bytes).  One thread = 4 consecutive pixels of a row x 8 output channels: per input row the 6 x Cin
// window sits in registers, every weight octet (LDS, fp32) is read once per 4 pixels; outputs leave as 16-byte
// NHWC stores.  Workgroup = Cout/8 channel octets x (256 / (Cout/8)) pixel groups.
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_first_kernel(const uint16_t* __restrict__ in,
                                                            const uint16_t* __restrict__ wt,
                                                            const uint16_t* __restrict__ bias,
                                                            uint16_t* __restrict__ out, int Nimg, int H, int W, int Cout)
{
    constexpr int PX = 4;                                          // pixels per thread
    extern __shared__ __attribute__((aligned(16))) float s_w[];   // [9*CIN][Cout]
    const int K = 9 * CIN;
    for (int i = threadIdx.x; i < K * Cout; i += 256) {
        const int k = i / Cout, co = i - k * Cout;                 // wt: [Cout][3][3][CIN]
        s_w[i] = bf2f(wt[(size_t)co * K + k]);
    }
    __syncthreads();
    const int octs = Cout >> 3;
    const int oct = threadIdx.x % octs, grp = threadIdx.x / octs;
    const int gpb = 256 / octs;                                    // pixel groups per workgroup
    const int gx = (W + PX - 1) / PX;                              // pixel groups per row
    const int64_t total = (int64_t)Nimg * H * gx;
    if (grp >= gpb) return;
    // persistent workgroups: the weights are staged once per workgroup, then it walks its share of the pixel groups
    for (int64_t g = (int64_t)blockIdx.x * gpb + grp; g < total; g += (int64_t)gridDim.x * gpb) {
    const int n = (int)((uint32_t)g / (uint32_t)(H * gx));          // total < 2^31 is checked by the host
    const int rem = (int)((uint32_t)g - (uint32_t)n * (uint32_t)(H * gx));
    const int y = rem / gx, x0 = (rem - y * gx) * PX;

    float acc[PX][8];
    {
        const uint4 bq = bias ? *(const uint4*)(bias + oct * 8) : make_uint4(0, 0, 0, 0);
        const uint32_t bw[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
        for (int p = 0; p < PX; p++)
#pragma unroll
            for (int k = 0; k < 8; k++) acc[p][k] = bf2f((uint16_t)(bw[k >> 1] >> ((k & 1) * 16)));
    }
#pragma unroll 1
    for (int r = 0; r < 3; r++) {
        const int yy = y + r - 1;
        if ((unsigned)yy >= (unsigned)H) continue;                 // zero row: contributes nothing
        const uint16_t* row = in + ((size_t)n * H + yy) * W * CIN;
        float win[PX + 2][CIN];
#pragma unroll
        for (int c = 0; c < PX + 2; c++) {
            const int xx = x0 + c - 1;
            const bool ok = (unsigned)xx < (unsigned)W;
#pragma unroll
            for (int ci = 0; ci < CIN; ci++) win[c][ci] = ok ? bf2f(row[(size_t)(ok ? xx : 0) * CIN + ci]) : 0.f;
        }
        const float* wr = s_w + (size_t)r * 3 * CIN * Cout + oct * 8;
#pragma unroll
        for (int kx = 0; kx < 3; kx++)
#pragma unroll
            for (int ci = 0; ci < CIN; ci++) {
                const float4 w0 = *(const float4*)(wr + (kx * CIN + ci) * Cout);
                const float4 w1 = *(const float4*)(wr + (kx * CIN + ci) * Cout + 4);
                const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                for (int p = 0; p < PX; p++) {
                    const float xv = win[p + kx][ci];
#pragma unroll
                    for (int k = 0; k < 8; k++) acc[p][k] += xv * wv[k];
                }
            }
    }
#pragma unroll
    for (int p = 0; p < PX; p++) {
        if (x0 + p >= W) break;
        uint4 o;
        o.x = pack_bf16(acc[p][0], acc[p][1]); o.y = pack_bf16(acc[p][2], acc[p][3]);
        o.z = pack_bf16(acc[p][4], acc[p][5]); o.w = pack_bf16(acc[p][6], acc[p][7]);
        *(uint4*)(out + (((size_t)n * H + y) * W + x0 + p) * Cout + oct * 8) = o;
    }
    }
}

// The VAE encoder's first convolution (3 -> 128 channels on the full-resolution image) on the matrix cores.  The VALU
// kernel above spends 27 fp32 FMAs per output -- 0.42 wave instructions per output at 4.5 cycles each, 365 us for
// the 8 x 512^2 x 128 tensor of the benchmark, three times the 537 MB output write.  Here a wave owns 32 consecutive
// pixels of a row x all 128 channels: the im2col operand B[k][pixel] (k = (ky, kx, ci), K = 9 * CIN padded to KS * 16)
// is gathered straight from the image (6 MB, L2 resident; 8 * KS two-byte loads per lane), the weights A[channel][k]
// stay in registers for the life of the (persistent) wave, and 4 * KS MFMAs produce the tile; the epilogue is the
// patch-staged kernels' (bf16 pack, 8-byte NHWC stores; the bias is a K column).  stat_part != NULL: GroupNorm partial sums of the
// stored values as in those kernels, but accumulated in registers over ALL tiles of an image that the wave
// processes and written once per (image, wave): rows = 8 * gridDim.x per image, every row written by every launch.
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_first_mfma_kernel(const uint16_t* __restrict__ in,
                                                                 const uint16_t* __restrict__ wt,
                                                                 const uint16_t* __restrict__ bias,
                                                                 uint16_t* __restrict__ out, int Nimg, int H, int W,
                                                                 float* __restrict__ stat_part)
{
    constexpr int K = 9 * CIN, KS = (K + 16) / 16, FA = 4, Cout = 128;   // K + 1 columns: the last one is the bias
    const int lane = threadIdx.x & 63, fn = lane & 31, fk = lane >> 5;
    const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int segs_x = (W + 31) >> 5;
    const int64_t segs = (int64_t)H * segs_x;

    // this lane's K slice: k = 16 * s + 8 * fk + j.  tap[s][j] = element offset of (ky - 1, kx - 1, ci) from the pixel,
    // times 64, plus the border conditions the tap needs (1: y > 0, 2: y < H - 1, 4: x > 0, 8: x < W - 1, 16: never;
    // 32: the constant 1.0 that multiplies the bias column)
    int tap[KS][8];
    bf16x8_t wa[FA][KS];
#pragma unroll
    for (int s = 0; s < KS; s++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int k = 16 * s + 8 * fk + j;
            const int ky = k / (3 * CIN), kx = (k / CIN) % 3, ci = k % CIN;
            const int need = k == K ? 32 : k > K ? 16 : ((ky == 0 ? 1 : 0) | (ky == 2 ? 2 : 0) | (kx == 0 ? 4 : 0) | (kx == 2 ? 8 : 0));
            const int off = k >= K ? 0 : ((ky - 1) * W + (kx - 1)) * CIN + ci;
            tap[s][j] = off * 64 + need;
        }
#pragma unroll
        for (int a = 0; a < FA; a++) {
            short w8[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int k = 16 * s + 8 * fk + j;
                // wt: [Cout][3][3][CIN]; the bias rides in the first padding column (its B entry is 1.0)
                w8[j] = k < K ? (short)wt[(size_t)(32 * a + fn) * K + k] : (k == K && bias) ? (short)bias[32 * a + fn] : (short)0;
            }
            wa[a][s] = bf16x8_t{w8[0], w8[1], w8[2], w8[3], w8[4], w8[5], w8[6], w8[7]};
        }
    }

    int64_t t = wave_g;
    for (int n = 0; n < Nimg; n++) {
        float2 st[FA][4];
#pragma unroll
        for (int a = 0; a < FA; a++)
#pragma unroll
            for (int q = 0; q < 4; q++) st[a][q] = make_float2(0.f, 0.f);
        for (; t < (int64_t)(n + 1) * segs; t += nwaves) {
            const int seg = (int)(t - (int64_t)n * segs);
            const int y = seg / segs_x, x = (seg - y * segs_x) * 32 + fn;
            const bool pix_ok = x < W;
            const int have = pix_ok ? ((y > 0 ? 1 : 0) | (y < H - 1 ? 2 : 0) | (x > 0 ? 4 : 0) | (x < W - 1 ? 8 : 0)) : 0;
            const uint16_t* base = in + (((size_t)n * H + y) * W + (pix_ok ? x : 0)) * CIN;
            f32x16 acc[FA];
#pragma unroll
            for (int a = 0; a < FA; a++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; s++) {
                short p8[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int need = tap[s][j] & 63;
                    const bool ok = pix_ok && (need & ~have) == 0;
                    p8[j] = need == 32 ? (short)0x3F80 : ok ? (short)base[tap[s][j] >> 6] : (short)0;
                }
                const bf16x8_t pb = bf16x8_t{p8[0], p8[1], p8[2], p8[3], p8[4], p8[5], p8[6], p8[7]};
#pragma unroll
                for (int a = 0; a < FA; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[a][s], pb, acc[a], 0, 0, 0);
            }
            // lanes l and l + 32 hold the two halves of each 8-channel octet of the same pixel: v_permlane32_swap
            // regroups two octets so that every lane stores 16 contiguous bytes (half as many write requests as
            // 8-byte stores -- this kernel is bound by them, not by bytes)
            uint16_t* orow = out + (((size_t)n * H + y) * W + x) * Cout + 8 * fk;
#pragma unroll
            for (int a = 0; a < FA; a++) {
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    uint2 o0, o1;
                    o0.x = pack_bf16(acc[a][4 * q], acc[a][4 * q + 1]);
                    o0.y = pack_bf16(acc[a][4 * q + 2], acc[a][4 * q + 3]);
                    o1.x = pack_bf16(acc[a][4 * q + 4], acc[a][4 * q + 5]);
                    o1.y = pack_bf16(acc[a][4 * q + 6], acc[a][4 * q + 7]);
                    if (stat_part && pix_ok) {
                        stat_accumulate(st[a][q], o0);
                        stat_accumulate(st[a][q + 1], o1);
                    }
                    const auto sx = __builtin_amdgcn_permlane32_swap(o0.x, o1.x, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(o0.y, o1.y, false, false);
                    if (pix_ok) *(uint4*)(orow + 32 * a + 8 * q) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                }
            }
        }
        if (stat_part) {
            const int rows = nwaves * 2, row = wave_g * 2 + ((lane >> 4) & 1);
#pragma unroll
            for (int a = 0; a < FA; a++) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float sx = row16_sum(st[a][q].x), sy = row16_sum(st[a][q].y);
                    const int quad = 8 * a + 2 * q + fk;
                    if ((lane & 15) == 0)
                        *(float2*)(stat_part + (((size_t)n * (Cout >> 2) + quad) * rows + row) * 2) = make_float2(sx, sy);
                }
            }
        }
    }
}

// w'[ci][tap][co] = w[co][8 - tap][ci]  (dgrad weights; run once per layer, weights are frozen)
__global__ void conv3x3_flip_weights_kernel(const uint16_t* __restrict__ w, uint16_t* __restrict__ wf, int Cout,
                                            int Cin)
{
    const size_t total = (size_t)Cout * 9 * Cin;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        const int tap = (int)((i / Cout) % 9);
        const int ci = (int)(i / ((size_t)Cout * 9));
        wf[i] = w[((size_t)co * 9 + (8 - tap)) * Cin + ci];
    }
}

int g_first_grid = 2048;   // persistent workgroups of the first-conv kernel (GD_NN_FIRST_GRID overrides, tuning)
int g_num_cus = 256;       // MI355X; refreshed from the device properties at the first patch launch
int g_patch_persistent = -1;   // GD_NN_PATCH_PERSISTENT=0: one workgroup per tile (A/B)
int g_route_scale = 1;     // kernel selection sees a batch of N * g_route_scale images (gd_nn_conv_set_route_scale)
int g_force_split = -1;    // tuning hook: -1 heuristic, 1 = never split, S > 1 = force S ranges of the K-step sequence
int g_force_variant = -1;  // tuning hook: 0 = 128x128, 1 = 128x256, 2 = 256x256, -1 = heuristic

// optional event timing of the conv kernel (bench.py's roofline line)
struct ConvProf {
    std::mutex mu;
    bool on = false;
    struct Rec { hipEvent_t a, b; };
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;
    double total_ms = 0, total_flops = 0, total_bytes = 0;
    int64_t launches = 0;
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        return hipEventCreate(&e) == hipSuccess ? e : nullptr;
    }
} g_cprof;

thread_local char g_err[256] = "";
int fail(int code, const char* msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

}  // namespace

extern "C" {

const char* gd_nn_conv_last_error(void) { return g_err; }

// host side of one launch: tile variant choice, event profiling, kernel launch
// Split-K factor for layers whose 128x128 tile grid cannot fill the chip (2 workgroups x 256 CUs): the nine taps
// are dealt to 3 or 9 workgroups per tile, fp32 partials are combined by conv_splitk_reduce_kernel.
// tools/splitk_sweep.py on MI355X (UNet maps at batch 1 and 2): the best split gives ~420 workgroups in flight while
// every workgroup keeps >= 7-8 K-steps; beyond 24 ranges the fp32 partial traffic costs more than the idle CUs.
static int pick_split(int64_t M, int64_t tiles, int steps, int target = 420)
{
    (void)M;
    int s = (int)((target + tiles / 2) / tiles);
    const int cap = steps * 2 / 15 < 24 ? steps * 2 / 15 : 24;
    if (s > cap) s = cap;
    return s < 2 ? 1 : s;
}

// fp32 scratch of a split-K launch: `split` images of the padded 128 x 128 tile grid (every workgroup stores its
// accumulators in MFMA fragment order, one contiguous 64 KB block per tile and K range)
static size_t split_ws_bytes(int split, int64_t M, int Cout)
{
    // (tiles of 128 channels x 128 pixels, or x 256 pixels -- split_tile_px; the pixel dimension padded to 256 covers both)
    return (size_t)split * (size_t)((M + 255) / 256 * 2) * (size_t)((Cout + 127) / 128) * 16384u * sizeof(float);
}

// Pixel extent of the split-K tile: 256 (the 128-channel x 256-pixel / 8-wave tile) on the larger of the small maps -- a
// layer whose whole filter bank is re-read once per pixel tile is bound by that traffic (1280 -> 1280 @ 8^2 x 16 latents:
// 236 MB of weight reads in 51 us), and the longer tile halves it -- else 128.  GD_NN_SPLIT_PX=128/256 forces (A/B).
int g_split_px = 0;
static int split_tile_px(int64_t M, int Cout)
{
    static int env_read = 0;
    if (!env_read) { if (const char* e = getenv("GD_NN_SPLIT_PX")) g_split_px = atoi(e); env_read = 1; }
    if (g_split_px == 128 || g_split_px == 256) return g_split_px;
    // profiles/r05_split_px_sweep.txt (UNet layers at 1 / 2 / 4 / 16 latents): the long tile wins from 1024 GEMM rows on when the
    // layer has >= 8 channel tiles (1280-wide: 1.03-1.15x), from 2048 rows on otherwise (640-wide: 1.0-1.06x); below, the
    // 128-pixel tile's finer split fills the chip better (up to 1.35x)
    M *= g_route_scale;
    return (M >= 1024 && Cout >= 1024) || M >= 2048 ? 256 : 128;
}

// Shallow, narrow layers on a half-empty tile grid (the UNet's 320 -> 320 convolutions on the 64^2 maps of one or two latents):
// 64 x 64 tiles, one launch, instead of two or three K ranges + the reduce launch -- 1.23-1.28x on MI355X
// (profiles/r05_small_tile_sweep.txt; every deeper or wider small-M layer is faster split, same file)
static bool small_tile_nosplit(int64_t M, int Cout, int ntaps, int Cin)
{
    if (g_force_split >= 0 || g_force_variant >= 0) return false;
    M *= g_route_scale;
    const int64_t t128 = ((M + 127) / 128) * ((Cout + 127) / 128), t64 = ((M + 63) / 64) * ((Cout + 63) / 64);
    return ntaps == 9 && Cin <= 320 && Cout <= 320 && Cout % 8 == 0 && t128 < 256 && t64 >= 256;
}

static int choose_split(int64_t M, int Cout, int ntaps, int Cin)
{
    const int steps = ntaps * (Cin / BK);
    if (g_force_split >= 0) return g_force_split > 1 && g_force_split <= steps ? g_force_split : 1;
    if (small_tile_nosplit(M, Cout, ntaps, Cin)) return 1;
    const int px = split_tile_px(M, Cout);
    M *= g_route_scale;
    if (((M + 127) / 128) * ((Cout + 127) / 128) >= 256) return 1;       // the unsplit 128 x 128 grid fills the chip
    const int64_t tiles = ((M + px - 1) / px) * ((Cout + 127) / 128);
    if (tiles >= 256 || Cout % 8 || steps < 8) return 1;
    // (the 8-wave 128 x 256 tile holds 96 KB of LDS: one workgroup per CU, so one wave of 256 workgroups is the target)
    return pick_split(M, tiles, steps, px == 256 ? 256 : 420);
}

static int launch_conv(hipStream_t s, const void* x, const void* weight, const void* bias, int bias_img_stride,
                       const void* residual, void* y, int N, ConvGeom g, int Cin, int Cout, void* ws = nullptr,
                       size_t ws_bytes = 0)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return fail(GD_NN_ERR_HIP, "hipGetDevice failed");
    // taps -> how far the activation descriptor must reach back so every soffset is >= 0
    int minlin = 0;
    for (int t = 0; t < g.ntaps; t++) {
        const int dy = (int)((g.ty4 >> (4 * t)) & 15u) - 8, dx = (int)((g.tx4 >> (4 * t)) & 15u) - 8;
        minlin = dy * g.Win + dx < minlin ? dy * g.Win + dx : minlin;
    }
    g.back = -minlin;
    if (g.wtaps == 0) g.wtaps = 9;
    if (((double)N * g.Hin * g.Win + g.back) * Cin * 2.0 >= 2147483648.0 || (double)Cout * 9.0 * Cin * 2.0 >= 2147483648.0)
        return fail(GD_NN_ERR_INVALID_ARG, "conv3x3: activation / weight tensor must be < 2 GiB (32-bit buffer offsets)");
    const int64_t M = (int64_t)N * g.Hg * g.Wg;
    if (M <= 0) return GD_NN_OK;
    // tile choice: the 256x256 / 8-wave tile has twice the MFMA work per byte staged through LDS;
    // use it when Cout fills it and there are enough tiles for 256 CUs, else 128 channels x 256
    // pixels, else the 128x128 / 4-wave tile.
    const int64_t Mo = (int64_t)N * g.Hout * g.Wout;
    int split = ws ? choose_split(M, Cout, g.ntaps, Cin) : 1;
    if (split > 1 && ws_bytes < split_ws_bytes(split, M, Cout)) split = 1;
    float* partial = split > 1 ? (float*)ws : nullptr;
    const int total_steps = g.ntaps * (Cin / BK);
    const int tps = split > 1 ? (total_steps + split - 1) / split : total_steps;
    if (split > 1) split = (total_steps + tps - 1) / tps;   // no empty ranges
    const int spx = split > 1 ? split_tile_px(M, Cout) : 128;
    int variant = split > 1 ? (spx == 256 ? 1 : 0) : g_force_variant;
    if (variant < 0 && small_tile_nosplit(M, Cout, g.ntaps, Cin)) variant = 6;
    if (variant < 0) {
        // rules distilled from tools/conv_kernel_bench.py on MI355X: the 256x256 tile wins whenever Cout fills it
        // and there is most of a wave of tiles; 128 ch x 256 px wins for long pixel dimensions with deep K or huge M;
        // otherwise the 128x128 tile (2 workgroups per CU) hides latency best.
        const int64_t Ms = M * g_route_scale;
        const int64_t t256 = ((Ms + 255) / 256) * ((Cout + 255) / 256);
        const int64_t t128x256 = ((Ms + 255) / 256) * ((Cout + 127) / 128);
        if (Cout % 256 == 0 && t256 >= 192) variant = 2;
        else if (Cout <= 32 && Ms >= (1 << 18)) variant = 4;   // few output channels (first conv's dgrad, Cout = 4)
        else if (t128x256 >= 512 && (Cin >= 512 || Ms >= (1 << 20))) variant = 1;
        else variant = 0;
    }
    hipEvent_t ea = nullptr, eb = nullptr;
    if (g_cprof.on) {
        std::lock_guard<std::mutex> lk(g_cprof.mu);
        ea = g_cprof.get(); eb = g_cprof.get();
        if (ea && eb) (void)hipEventRecord(ea, s);
    }
#define GD_LAUNCH(BN_, BM_, WN_, WM_)                                                                              \
    do {                                                                                                           \
        auto kern = conv3x3_nhwc_bf16_kernel<BN_, BM_, WN_, WM_>;                                                  \
        constexpr int lds = 2 * (BN_ + BM_) * BK * 2;                                                              \
        static bool attr_set[16] = {false};                                                                        \
        if (!attr_set[dev]) {                                                                                      \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);         \
            attr_set[dev] = true;                                                                                  \
        }                                                                                                          \
        const int tiles_m = (int)((M + BM_ - 1) / BM_), tiles_n = (Cout + BN_ - 1) / BN_;                          \
        const int nwg = tiles_m * tiles_n;                                                                         \
        hipLaunchKernelGGL(kern, dim3(nwg, split), dim3(64 * WN_ * WM_), lds, s, (const uint16_t*)x,               \
                           (const uint16_t*)weight, (const uint16_t*)bias, bias_img_stride,                        \
                           (const uint16_t*)residual, (uint16_t*)y, N, g, Cin, Cout, tiles_n, nwg, partial, tps);  \
    } while (0)
    if (variant == 2) GD_LAUNCH(256, 256, 2, 4);
    else if (variant == 4) GD_LAUNCH(32, 256, 1, 4);   // 32 channels x 256 pixels, 4 waves: 1/4 of the padded MFMA work
    else if (variant == 1) GD_LAUNCH(128, 256, 2, 4);
    else if (variant == 5) GD_LAUNCH(64, 128, 2, 2);   // small-M layers: finer tiles instead of split-K + a reduce launch
    else if (variant == 6) GD_LAUNCH(64, 64, 2, 2);
    else if (variant == 7) GD_LAUNCH(128, 64, 2, 2);
    else GD_LAUNCH(128, 128, 2, 2);
#undef GD_LAUNCH
    if (split > 1) {
        const int tiles_n = (Cout + 127) / 128, nwg = (int)((M + spx - 1) / spx) * tiles_n;
        if (spx == 256)
            hipLaunchKernelGGL(conv_splitk_reduce_kernel<256>, dim3((unsigned)nwg * 32u), dim3(256), 0, s, partial, split, nwg,
                               tiles_n, M, Cout, g.Hg * g.Wg, g.Wg, g.Hout, g.Wout, g.osy, g.osx, g.ooy, g.oox,
                               (const uint16_t*)bias, bias_img_stride, (const uint16_t*)residual, (uint16_t*)y);
        else
            hipLaunchKernelGGL(conv_splitk_reduce_kernel<128>, dim3((unsigned)nwg * 16u), dim3(256), 0, s, partial, split, nwg,
                               tiles_n, M, Cout, g.Hg * g.Wg, g.Wg, g.Hout, g.Wout, g.osy, g.osx, g.ooy, g.oox,
                               (const uint16_t*)bias, bias_img_stride, (const uint16_t*)residual, (uint16_t*)y);
    }
    if (ea && eb) {
        (void)hipEventRecord(eb, s);
        std::lock_guard<std::mutex> lk(g_cprof.mu);
        g_cprof.pending.push_back({ea, eb});
        g_cprof.total_flops += 2.0 * (double)M * Cout * (double)g.ntaps * Cin;
        // algorithmic HBM bytes: every input element once, the taps' weights once, every output element once
        g_cprof.total_bytes += 2.0 * ((double)N * g.Hin * g.Win * Cin + (double)g.ntaps * Cin * Cout + (double)M * Cout +
                                      (residual ? (double)M * Cout : 0.0));
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

static void add_tap(ConvGeom& g, int dy, int dx, int widx)
{
    g.ty4 |= (uint64_t)(dy + 8) << (4 * g.ntaps);
    g.tx4 |= (uint64_t)(dx + 8) << (4 * g.ntaps);
    g.w4 |= (uint64_t)widx << (4 * g.ntaps);
    g.ntaps++;
}

// One launch for up to four geometries of the same layer (conv3x3_nhwc_bf16_multi_kernel).  Returns 1 if the classes were
// launched, 0 if the shape wants split-K or differing tiles (the caller then launches the classes one by one), < 0 on error.
int g_multi_class = 1;      // GD_NN_CONV_MULTI=0: the classes back to back as before round 5 (same-box A/B)
static int launch_conv_multi(hipStream_t s, const void* x, const void* const* weights, const void* bias, int bias_img_stride,
                             const void* residual, void* y, int N, ConvGeom* geoms, int ncls, int Cin, int Cout, bool may_split)
{
    static int env_read = 0;
    if (!env_read) { if (const char* e = getenv("GD_NN_CONV_MULTI")) g_multi_class = atoi(e); env_read = 1; }
    if (!g_multi_class || ncls < 2 || ncls > 4 || g_force_variant >= 0 || g_force_split >= 0) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return fail(GD_NN_ERR_HIP, "hipGetDevice failed");
    int64_t Mmax = 0, Msum = 0;
    double flops = 0, bytes = 0;
    for (int c = 0; c < ncls; c++) {
        ConvGeom& g = geoms[c];
        int minlin = 0;
        for (int t = 0; t < g.ntaps; t++) {
            const int dy = (int)((g.ty4 >> (4 * t)) & 15u) - 8, dx = (int)((g.tx4 >> (4 * t)) & 15u) - 8;
            minlin = dy * g.Win + dx < minlin ? dy * g.Win + dx : minlin;
        }
        g.back = -minlin;
        if (g.wtaps == 0) g.wtaps = 9;
        if (((double)N * g.Hin * g.Win + g.back) * Cin * 2.0 >= 2147483648.0 || (double)Cout * 9.0 * Cin * 2.0 >= 2147483648.0)
            return fail(GD_NN_ERR_INVALID_ARG, "conv3x3: activation / weight tensor must be < 2 GiB (32-bit buffer offsets)");
        const int64_t M = (int64_t)N * g.Hg * g.Wg;
        if (M <= 0) return 0;
        if (may_split && choose_split(M, Cout, g.ntaps, Cin) > 1) return 0;     // small maps: the split-K path, class by class
        Mmax = M > Mmax ? M : Mmax;
        Msum += M;
        flops += 2.0 * (double)M * Cout * (double)g.ntaps * Cin;
        bytes += 2.0 * ((double)g.ntaps * Cin * Cout + (double)M * Cout + (residual ? (double)M * Cout : 0.0));
    }
    bytes += 2.0 * (double)N * geoms[0].Hin * geoms[0].Win * Cin;
    // tile choice on the SUM of the classes' rows (they run concurrently), same rules as launch_conv
    int variant = 0;
    {
        const int64_t Ms = Msum * g_route_scale;
        const int64_t t256 = ((Ms + 255) / 256) * ((Cout + 255) / 256);
        const int64_t t128x256 = ((Ms + 255) / 256) * ((Cout + 127) / 128);
        if (Cout % 256 == 0 && t256 >= 192) variant = 2;
        else if (t128x256 >= 512 && (Cin >= 512 || Ms >= (1 << 20))) variant = 1;
    }
    hipEvent_t ea = nullptr, eb = nullptr;
    if (g_cprof.on) {
        std::lock_guard<std::mutex> lk(g_cprof.mu);
        ea = g_cprof.get(); eb = g_cprof.get();
        if (ea && eb) (void)hipEventRecord(ea, s);
    }
#define GD_LAUNCH_M(BN_, BM_, WN_, WM_)                                                                            \
    do {                                                                                                           \
        auto kern = conv3x3_nhwc_bf16_multi_kernel<BN_, BM_, WN_, WM_>;                                            \
        constexpr int lds = 2 * (BN_ + BM_) * BK * 2;                                                              \
        static bool attr_set[16] = {false};                                                                        \
        if (!attr_set[dev]) {                                                                                      \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);         \
            attr_set[dev] = true;                                                                                  \
        }                                                                                                          \
        ConvGeomSet gs = {};                                                                                       \
        gs.n = ncls;                                                                                               \
        const int tiles_n = (Cout + BN_ - 1) / BN_;                                                                \
        int nmax = 0;                                                                                              \
        for (int c = 0; c < ncls; c++) {                                                                           \
            gs.g[c] = geoms[c];                                                                                    \
            gs.wt[c] = (const uint16_t*)weights[c];                                                                \
            const int64_t M = (int64_t)N * geoms[c].Hg * geoms[c].Wg;                                              \
            gs.nwg[c] = (int)((M + BM_ - 1) / BM_) * tiles_n;                                                      \
            nmax = gs.nwg[c] > nmax ? gs.nwg[c] : nmax;                                                            \
        }                                                                                                          \
        hipLaunchKernelGGL(kern, dim3(nmax, 1, ncls), dim3(64 * WN_ * WM_), lds, s, (const uint16_t*)x,            \
                           (const uint16_t*)bias, bias_img_stride, (const uint16_t*)residual, (uint16_t*)y, N, gs, \
                           Cin, Cout, tiles_n);                                                                    \
    } while (0)
    // (An LDS-transposed epilogue for the short-K classes -- 16-byte stores, eight lanes per pixel row -- was built and measured
    // in round 5: 43.68 / 43.76 ms per 8-view step against 43.62 / 43.63 with the direct 8-byte stores, same box; removed.)
    if (variant == 2) GD_LAUNCH_M(256, 256, 2, 4);
    else if (variant == 1) GD_LAUNCH_M(128, 256, 2, 4);
    else GD_LAUNCH_M(128, 128, 2, 2);
#undef GD_LAUNCH_M
    if (ea && eb) {
        (void)hipEventRecord(eb, s);
        std::lock_guard<std::mutex> lk(g_cprof.mu);
        g_cprof.pending.push_back({ea, eb});
        g_cprof.total_flops += flops;
        g_cprof.total_bytes += bytes;
    }
    (void)Mmax;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return 1;
}


static int launch_patch(void* stream, const void* x, const float* mean_rstd, const void* gamma, const void* beta,
                        int groups, int apply_silu, const void* weight, const void* bias, int bias_img_stride,
                        const void* residual, void* y, int N, int H, int W, int Cin, int Cout,
                        float* stat_part = nullptr);

// Plain stride-1 convolutions run on the persistent patch-staged kernel (LDS-DMA patch) once its (image, 16x16
// patch, 128- or 256-channel slab) grid has >= 256 workgroups: 1.08-1.24x faster than the implicit-GEMM tiles on
// every such shape of the workload, Cout % 256 == 0 included (tools/patch_conv_bench.py: 256->256@256^2 622 -> 537 us,
// 512->512@128^2 534 -> 487 us, 1280->1280@32^2 549 -> 510 us); below that (16^2 maps, one view per GPU) the
// implicit-GEMM kernel's finer tiles and split-K win by 1.3-3x.
static bool prefer_patch(int N, int H, int W, int Cout)
{
    if (g_force_variant >= 0 || g_force_split >= 0) return g_force_variant == 3;
    if (Cout < 64 || H < 16 || W < 16) return false;
    const int bn = Cout % 256 == 0 ? 256 : 128;
    const int64_t wgs = (int64_t)N * g_route_scale * ((H + 15) / 16) * ((W + 15) / 16) * ((Cout + bn - 1) / bn);
    return wgs >= 256;
}

size_t gd_nn_conv3x3_ws_bytes(int N, int H, int W, int Cin, int Cout)
{
    if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
    if (prefer_patch(N, H, W, Cout)) return 0;
    const int64_t M = (int64_t)N * H * W;
    const int split = choose_split(M, Cout, 9, Cin);
    return split > 1 ? split_ws_bytes(split, M, Cout) : 0;
}

int gd_nn_conv_force_split(int s)
{
    g_force_split = s;
    return GD_NN_OK;
}

int gd_nn_conv_set_route_scale(int k)
{
    if (k < 1) return fail(GD_NN_ERR_INVALID_ARG, "conv route scale must be >= 1");
    g_route_scale = k;
    return GD_NN_OK;
}

int gd_nn_conv3x3_forward_ws(void* stream, const void* x, const void* weight, const void* bias, int bias_img_stride,
                             const void* residual, void* y, int N, int H, int W, int Cin, int Cout, void* ws,
                             size_t ws_bytes)
{
    if (!x || !weight || !y) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || Cin % BK || Cout % 4 || Cin <= 0 || Cout <= 0)
        return fail(GD_NN_ERR_INVALID_ARG, "conv3x3: need Cin % 64 == 0 and Cout % 4 == 0");
    if (prefer_patch(N, H, W, Cout))
        return launch_patch(stream, x, nullptr, nullptr, nullptr, 0, 0, weight, bias, bias_img_stride, residual, y, N, H, W,
                            Cin, Cout);
    ConvGeom g = {};
    g.Hin = g.Hg = g.Hout = H;
    g.Win = g.Wg = g.Wout = W;
    g.sy = g.sx = g.osy = g.osx = 1;
    for (int ky = 0; ky < 3; ky++)
        for (int kx = 0; kx < 3; kx++) add_tap(g, ky - 1, kx - 1, ky * 3 + kx);
    return launch_conv((hipStream_t)stream, x, weight, bias, bias_img_stride, residual, y, N, g, Cin, Cout, ws, ws_bytes);
}

int gd_nn_conv3x3_forward(void* stream, const void* x, const void* weight, const void* bias, int bias_img_stride,
                          const void* residual, void* y, int N, int H, int W, int Cin, int Cout)
{
    if (!x || !weight || !y) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || Cin % BK || Cout % 4 || Cin <= 0 || Cout <= 0)
        return fail(GD_NN_ERR_INVALID_ARG, "conv3x3: need Cin % 64 == 0 and Cout % 4 == 0");
    ConvGeom g = {};
    g.Hin = g.Hg = g.Hout = H;
    g.Win = g.Wg = g.Wout = W;
    g.sy = g.sx = g.osy = g.osx = 1;
    for (int ky = 0; ky < 3; ky++)
        for (int kx = 0; kx < 3; kx++) add_tap(g, ky - 1, kx - 1, ky * 3 + kx);
    return launch_conv((hipStream_t)stream, x, weight, bias, bias_img_stride, residual, y, N, g, Cin, Cout);
}

static int launch_patch(void* stream, const void* x, const float* mean_rstd, const void* gamma, const void* beta,
                        int groups, int apply_silu, const void* weight, const void* bias, int bias_img_stride,
                        const void* residual, void* y, int N, int H, int W, int Cin, int Cout,
                        float* stat_part)
{
    if (!x || !weight || !y) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || Cin % BK || Cout % 4 || Cin <= 0 || Cout <= 0)
        return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_gn: need Cin % 64 == 0 and Cout % 4 == 0");
    if (mean_rstd && (!gamma || !beta || groups <= 0 || Cin % groups))
        return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_gn: GroupNorm needs gamma, beta and Cin % groups == 0");
    if ((double)N * H * W * Cin * 2.0 >= 2147483648.0 || (double)Cout * 9.0 * Cin * 2.0 >= 2147483648.0)
        return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_gn: activation / weight tensor must be < 2 GiB (32-bit buffer offsets)");
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return fail(GD_NN_ERR_HIP, "hipGetDevice failed");
    if (g_patch_persistent < 0) {
        const char* e = getenv("GD_NN_PATCH_PERSISTENT");
        g_patch_persistent = (e && atoi(e) == 0) ? 0 : 1;
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) g_num_cus = cus;
    }
    hipStream_t s = (hipStream_t)stream;
    const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
    const int64_t M = (int64_t)N * H * W;
    hipEvent_t ea = nullptr, eb = nullptr;
    if (g_cprof.on) {
        std::lock_guard<std::mutex> lk(g_cprof.mu);
        ea = g_cprof.get(); eb = g_cprof.get();
        if (ea && eb) (void)hipEventRecord(ea, s);
    }
#define GD_LAUNCH_P(BN_, GN_)                                                                                         \
    do {                                                                                                           \
        auto kern = GN_ ? conv3x3_gn_patch_kernel<BN_, 2, 4, true> : conv3x3_patch_stream_kernel<BN_, 2, 4, false>; \
        constexpr int lds = 2 * (kPatchPix + 4) * BK * 2 + 2 * BN_ * BK * 2;                                       \
        static bool attr_set[16] = {false};                                                                        \
        if (!attr_set[dev]) {                                                                                      \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);         \
            attr_set[dev] = true;                                                                                  \
        }                                                                                                          \
        const int tiles_n = (Cout + BN_ - 1) / BN_;                                                                \
        const int nwg = N * tiles_x * tiles_y * tiles_n;                                                           \
        /* persistent workgroups: one per CU (the LDS footprint allows no more), each walking tiles b, b + grid, */ \
        /* ...; a multiple of 8 keeps the XCD-contiguous tile order */                                             \
        int grid = nwg;                                                                                            \
        if (g_patch_persistent && !(GN_) && nwg > g_num_cus) grid = (nwg & 7) == 0 ? (g_num_cus & ~7) : g_num_cus; \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, (const uint16_t*)x, (const uint16_t*)weight,       \
                           (const uint16_t*)bias, bias_img_stride, (const uint16_t*)residual, (uint16_t*)y, N, H,  \
                           W, Cin, Cout, mean_rstd, (const uint16_t*)gamma, (const uint16_t*)beta, groups,         \
                           apply_silu, tiles_n, tiles_x, tiles_y, nwg, stat_part);                                 \
    } while (0)
    int bn = (Cout % 256 == 0) ? 256 : 128;
    if (g_force_variant == 1 || g_force_variant == 0) bn = 128;
    if (g_force_variant == 2) bn = 256;
    if (mean_rstd) {
        if (bn == 256) GD_LAUNCH_P(256, true);
        else GD_LAUNCH_P(128, true);
    } else {
        if (bn == 256) GD_LAUNCH_P(256, false);
        else GD_LAUNCH_P(128, false);
    }
#undef GD_LAUNCH_P
    if (ea && eb) {
        (void)hipEventRecord(eb, s);
        std::lock_guard<std::mutex> lk(g_cprof.mu);
        g_cprof.pending.push_back({ea, eb});
        g_cprof.total_flops += 2.0 * (double)M * Cout * 9.0 * Cin;
        g_cprof.total_bytes += 2.0 * ((double)M * Cin + 9.0 * Cin * Cout + (double)M * Cout +
                                      (residual ? (double)M * Cout : 0.0));
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

int gd_nn_conv3x3_gn_forward(void* stream, const void* x, const float* mean_rstd, const void* gamma, const void* beta,
                             int groups, int apply_silu, const void* weight, const void* bias, int bias_img_stride,
                             const void* residual, void* y, int N, int H, int W, int Cin, int Cout)
{
    return launch_patch(stream, x, mean_rstd, gamma, beta, groups, apply_silu, weight, bias, bias_img_stride, residual, y,
                        N, H, W, Cin, Cout);
}

size_t gd_nn_conv3x3_stat_rows(int N, int H, int W, int Cout, int gn_entry)
{
    if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout % 4) return 0;
    if (!gn_entry && !prefer_patch(N, H, W, Cout)) return 0;
    return (size_t)((H + 15) / 16) * ((W + 15) / 16) * 8;
}

int gd_nn_conv3x3_gn_forward_stats(void* stream, const void* x, const float* mean_rstd, const void* gamma,
                                   const void* beta, int groups, int apply_silu, const void* weight, const void* bias,
                                   int bias_img_stride, const void* residual, void* y, int N, int H, int W, int Cin,
                                   int Cout, float* stat_part)
{
    return launch_patch(stream, x, mean_rstd, gamma, beta, groups, apply_silu, weight, bias, bias_img_stride, residual, y,
                        N, H, W, Cin, Cout, stat_part);
}

int gd_nn_conv3x3_forward_stats(void* stream, const void* x, const void* weight, const void* bias, int bias_img_stride,
                                const void* residual, void* y, int N, int H, int W, int Cin, int Cout, float* stat_part)
{
    if (!stat_part) return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_forward_stats: stat_part is NULL");
    if (!prefer_patch(N, H, W, Cout))
        return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_forward_stats: this shape does not run on the patch-staged kernel "
                                           "(gd_nn_conv3x3_stat_rows() == 0)");
    return launch_patch(stream, x, nullptr, nullptr, nullptr, 0, 0, weight, bias, bias_img_stride, residual, y, N, H, W,
                        Cin, Cout, stat_part);
}

// ---- Winograd F(2,3)-along-x form of the stride-1 convolution (nn_conv_wino.h)

size_t gd_nn_conv3x3_wino_weights_bytes(int Cout, int Cin)
{
    if (Cout <= 0 || Cin <= 0 || Cin % kWinoCK) return 0;
    return (size_t)((Cout + 127) / 128) * 3 * (size_t)(Cin / kWinoCK) * (size_t)kWinoWStage;
}

int gd_nn_conv3x3_wino_weights(void* stream, const void* weight, void* u, int Cout, int Cin)
{
    if (!weight || !u || Cout <= 0 || Cin <= 0) return fail(GD_NN_ERR_INVALID_ARG, "wino_weights: bad argument");
    if (Cin % kWinoCK) return fail(GD_NN_ERR_INVALID_ARG, "wino_weights: Cin % 32 != 0");
    const size_t total = gd_nn_conv3x3_wino_weights_bytes(Cout, Cin) / 16;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(conv3x3_wino_weights_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)weight,
                       (uint16_t*)u, Cout, Cin);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

int gd_nn_conv3x3_wino_supported(int N, int H, int W, int Cin, int Cout)
{
    if (N <= 0 || H < 16 || W < 16 || Cin <= 0 || Cout < 64) return 0;
    if (Cin % kWinoCK || Cout % 8) return 0;
    if ((double)N * H * W * Cin * 2.0 >= 2147483648.0 || (double)gd_nn_conv3x3_wino_weights_bytes(Cout, Cin) >= 2147483648.0) return 0;
    return 1;
}

static int launch_wino(void* stream, const void* x, const float* mean_rstd, const void* gamma, const void* beta, int groups,
                       int apply_silu, const void* u, const void* bias, int bias_img_stride, const void* residual, void* y,
                       int N, int H, int W, int Cin, int Cout, float* stat_part)
{
    if (!x || !u || !y) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (!gd_nn_conv3x3_wino_supported(N, H, W, Cin, Cout))
        return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_wino: need Cin % 32 == 0, Cout % 8 == 0, Cout >= 64, H, W >= 16, tensors < 2 GiB");
    if (mean_rstd && (!gamma || !beta || groups <= 0 || Cin % groups))
        return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_wino: GroupNorm needs gamma, beta and Cin % groups == 0");
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return fail(GD_NN_ERR_HIP, "hipGetDevice failed");
    hipStream_t s = (hipStream_t)stream;
    const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
    const int64_t M = (int64_t)N * H * W;
    hipEvent_t ea = nullptr, eb = nullptr;
    if (g_cprof.on) {
        std::lock_guard<std::mutex> lk(g_cprof.mu);
        ea = g_cprof.get(); eb = g_cprof.get();
        if (ea && eb) (void)hipEventRecord(ea, s);
    }
    const int tiles_n = (Cout + 127) / 128;
    const int nwg = N * tiles_x * tiles_y * tiles_n;
    const int grid = nwg;      // one tile per workgroup (the persistent form measured slower: nn_conv_wino.h)
#define GD_LAUNCH_W(GN_)                                                                                           \
    do {                                                                                                           \
        auto kern = conv3x3_wino_kernel<GN_>;                                                                      \
        static bool attr_set[16] = {false};                                                                        \
        if (!attr_set[dev]) {                                                                                      \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kWinoLds);    \
            attr_set[dev] = true;                                                                                  \
        }                                                                                                          \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), kWinoLds, s, (const uint16_t*)x, (const uint16_t*)u,       \
                           (const uint16_t*)bias, bias_img_stride, (const uint16_t*)residual, (uint16_t*)y, N, H,  \
                           W, Cin, Cout, mean_rstd, (const uint16_t*)gamma, (const uint16_t*)beta, groups,         \
                           apply_silu, tiles_n, tiles_x, tiles_y, nwg, stat_part);                                 \
    } while (0)
    if (mean_rstd) GD_LAUNCH_W(true);
    else GD_LAUNCH_W(false);
#undef GD_LAUNCH_W
    if (ea && eb) {
        (void)hipEventRecord(eb, s);
        std::lock_guard<std::mutex> lk(g_cprof.mu);
        g_cprof.pending.push_back({ea, eb});
        // ALGORITHMIC flops of the convolution (direct form), whatever the kernel multiplies
        g_cprof.total_flops += 2.0 * (double)M * Cout * 9.0 * Cin;
        g_cprof.total_bytes += 2.0 * ((double)M * Cin + 9.0 * Cin * Cout + (double)M * Cout +
                                      (residual ? (double)M * Cout : 0.0));
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

int gd_nn_conv3x3_wino_forward(void* stream, const void* x, const void* u, const void* bias, int bias_img_stride,
                               const void* residual, void* y, int N, int H, int W, int Cin, int Cout, float* stat_part)
{
    return launch_wino(stream, x, nullptr, nullptr, nullptr, 0, 0, u, bias, bias_img_stride, residual, y, N, H, W, Cin,
                       Cout, stat_part);
}

int gd_nn_conv3x3_wino_gn_forward(void* stream, const void* x, const float* mean_rstd, const void* gamma, const void* beta,
                                  int groups, int apply_silu, const void* u, const void* bias, int bias_img_stride,
                                  const void* residual, void* y, int N, int H, int W, int Cin, int Cout, float* stat_part)
{
    if (!mean_rstd) return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_wino_gn: mean_rstd is NULL");
    return launch_wino(stream, x, mean_rstd, gamma, beta, groups, apply_silu, u, bias, bias_img_stride, residual, y, N, H, W,
                       Cin, Cout, stat_part);
}

// ---- wide-tile (128 channels x 16 x 32 pixels) direct form for layers with few output channels (nn_conv_wide.h)
size_t gd_nn_conv3x3_wide_weights_bytes(int Cout, int Cin)
{
    if (Cout <= 0 || Cin <= 0 || Cin % kWideCK) return 0;
    return (size_t)((Cout + 127) / 128) * 3 * (size_t)(Cin / kWideCK) * (size_t)kWideWStage;
}

int gd_nn_conv3x3_wide_weights(void* stream, const void* weight, void* u, int Cout, int Cin)
{
    if (!weight || !u || Cout <= 0 || Cin <= 0 || Cin % kWideCK) return fail(GD_NN_ERR_INVALID_ARG, "wide_weights: bad argument");
    const size_t total = gd_nn_conv3x3_wide_weights_bytes(Cout, Cin) / 16;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(conv3x3_wide_weights_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)weight,
                       (uint16_t*)u, Cout, Cin);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

#ifdef GD_NN_EXPERIMENTAL_STREAM   // tools/stream_variants.sh only: a measured negative, not in libgd_nn.so
#include "nn_conv_stream_api.h"
#endif

int gd_nn_conv3x3_wide_supported(int N, int H, int W, int Cin, int Cout)
{
    if (N <= 0 || H < 16 || W < 16 || Cin <= 0 || Cout < 64) return 0;
    if (Cin % kWideCK || Cout % 8) return 0;
    if ((double)N * H * W * Cin * 2.0 >= 2147483648.0 || (double)gd_nn_conv3x3_wide_weights_bytes(Cout, Cin) >= 2147483648.0) return 0;
    return 1;
}

static int launch_wide(void* stream, const void* x, const float* mean_rstd, const void* gamma, const void* beta, int groups,
                       int apply_silu, const void* u, const void* bias, int bias_img_stride, const void* residual, void* y,
                       int N, int H, int W, int Cin, int Cout, float* stat_part)
{
    if (!x || !u || !y) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (!gd_nn_conv3x3_wide_supported(N, H, W, Cin, Cout))
        return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_wide: need Cin % 32 == 0, Cout % 8 == 0, Cout >= 64, H, W >= 16, tensors < 2 GiB");
    if (mean_rstd && (!gamma || !beta || groups <= 0 || Cin % groups))
        return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_wide: GroupNorm needs gamma, beta and Cin % groups == 0");
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return fail(GD_NN_ERR_HIP, "hipGetDevice failed");
    hipStream_t s = (hipStream_t)stream;
    const int tiles_x = (W + 31) / 32, tiles_y = (H + 15) / 16;
    const int64_t M = (int64_t)N * H * W;
    hipEvent_t ea = nullptr, eb = nullptr;
    if (g_cprof.on) {
        std::lock_guard<std::mutex> lk(g_cprof.mu);
        ea = g_cprof.get(); eb = g_cprof.get();
        if (ea && eb) (void)hipEventRecord(ea, s);
    }
    const int tiles_n = (Cout + 127) / 128;
    const int nwg = N * tiles_x * tiles_y * tiles_n;
#define GD_LAUNCH_WD(GN_)                                                                                          \
    do {                                                                                                           \
        auto kern = conv3x3_wide_kernel<GN_>;                                                                      \
        static bool attr_set[16] = {false};                                                                        \
        if (!attr_set[dev]) {                                                                                      \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kWideLds);    \
            attr_set[dev] = true;                                                                                  \
        }                                                                                                          \
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), kWideLds, s, (const uint16_t*)x, (const uint16_t*)u,        \
                           (const uint16_t*)bias, bias_img_stride, (const uint16_t*)residual, (uint16_t*)y, N, H,  \
                           W, Cin, Cout, mean_rstd, (const uint16_t*)gamma, (const uint16_t*)beta, groups,         \
                           apply_silu, tiles_n, tiles_x, tiles_y, nwg, stat_part);                                 \
    } while (0)
    if (mean_rstd) GD_LAUNCH_WD(true);
    else GD_LAUNCH_WD(false);
#undef GD_LAUNCH_WD
    if (ea && eb) {
        (void)hipEventRecord(eb, s);
        std::lock_guard<std::mutex> lk(g_cprof.mu);
        g_cprof.pending.push_back({ea, eb});
        g_cprof.total_flops += 2.0 * (double)M * Cout * 9.0 * Cin;
        g_cprof.total_bytes += 2.0 * ((double)M * Cin + 9.0 * Cin * Cout + (double)M * Cout +
                                      (residual ? (double)M * Cout : 0.0));
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

int gd_nn_conv3x3_wide_forward(void* stream, const void* x, const void* u, const void* bias, int bias_img_stride,
                               const void* residual, void* y, int N, int H, int W, int Cin, int Cout, float* stat_part)
{
    return launch_wide(stream, x, nullptr, nullptr, nullptr, 0, 0, u, bias, bias_img_stride, residual, y, N, H, W, Cin,
                       Cout, stat_part);
}

int gd_nn_conv3x3_wide_gn_forward(void* stream, const void* x, const float* mean_rstd, const void* gamma, const void* beta,
                                  int groups, int apply_silu, const void* u, const void* bias, int bias_img_stride,
                                  const void* residual, void* y, int N, int H, int W, int Cin, int Cout, float* stat_part)
{
    if (!mean_rstd) return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_wide_gn: mean_rstd is NULL");
    return launch_wide(stream, x, mean_rstd, gamma, beta, groups, apply_silu, u, bias, bias_img_stride, residual, y, N, H, W,
                       Cin, Cout, stat_part);
}

#ifdef GD_NN_EXPERIMENTAL_REGW   // tools/regw_variants.sh only: a measured negative, not in libgd_nn.so
#include "nn_conv_regw_api.h"
#endif

// persistent workgroups of the matrix-core first convolution: two per CU (8 waves of ~170 VGPRs), fewer for small inputs
static int first_mfma_grid(int N, int H, int W)
{
    const int64_t tiles = (int64_t)N * H * ((W + 31) / 32);
    int64_t grid = (tiles + 3) / 4;
    const int64_t cap = 512;   // 2 per CU of an MI355X; a constant, so that gd_nn_conv3x3_first_stat_rows is a pure function
    if (grid > cap) grid = cap;
    return (int)(grid < 1 ? 1 : grid);
}

size_t gd_nn_conv3x3_first_stat_rows(int N, int H, int W, int Cin, int Cout)
{
    if (N <= 0 || H <= 0 || W <= 0 || Cin < 1 || Cin > 4 || Cout != 128 || g_force_variant >= 0) return 0;
    return (size_t)first_mfma_grid(N, H, W) * 8;
}

static int first_forward(void* stream, const void* x, const void* weight, const void* bias, void* y, int N, int H, int W,
                         int Cin, int Cout, float* stat_part);

int gd_nn_conv3x3_first_forward(void* stream, const void* x, const void* weight, const void* bias, void* y, int N, int H,
                                int W, int Cin, int Cout)
{
    return first_forward(stream, x, weight, bias, y, N, H, W, Cin, Cout, nullptr);
}

int gd_nn_conv3x3_first_forward_stats(void* stream, const void* x, const void* weight, const void* bias, void* y, int N,
                                      int H, int W, int Cin, int Cout, float* stat_part)
{
    if (!stat_part) return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_first_forward_stats: stat_part is NULL");
    return first_forward(stream, x, weight, bias, y, N, H, W, Cin, Cout, stat_part);
}

static int first_forward(void* stream, const void* x, const void* weight, const void* bias, void* y, int N, int H, int W,
                         int Cin, int Cout, float* stat_part)
{
    if (!x || !weight || !y) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || Cin < 1 || Cin > 4 || Cout % 8 || Cout <= 0 || (size_t)9 * Cin * Cout * 4 > 65536)
        return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_first: need 1 <= Cin <= 4, Cout % 8 == 0 and 36*Cin*Cout <= 64 KiB of LDS");
    if (stat_part && Cout != 128) return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_first: statistics need Cout == 128");
    if (Cout == 128 && g_force_variant < 0) {     // VAE encoder: the matrix-core kernel
        if ((double)N * H * W * Cin >= 2147483648.0) return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_first: tensor too large");
        const int grid = first_mfma_grid(N, H, W);
#define GD_FIRST_M(C_)                                                                                             \
    hipLaunchKernelGGL(conv3x3_first_mfma_kernel<C_>, dim3(grid), dim3(256), 0, (hipStream_t)stream,                \
                       (const uint16_t*)x, (const uint16_t*)weight, (const uint16_t*)bias, (uint16_t*)y, N, H, W, stat_part)
        if (Cin == 1) GD_FIRST_M(1);
        else if (Cin == 2) GD_FIRST_M(2);
        else if (Cin == 3) GD_FIRST_M(3);
        else GD_FIRST_M(4);
#undef GD_FIRST_M
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
        return GD_NN_OK;
    }
    const int octs = Cout / 8, gpb = 256 / octs;
    const int64_t groups = (int64_t)N * H * ((W + 3) / 4);
    const int64_t blocks = (groups + gpb - 1) / gpb;
    if (blocks > 2147483647LL) return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_first: tensor too large");
    const size_t lds = (size_t)9 * Cin * Cout * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    if (groups > 2147483647LL) return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_first: tensor too large");
    if (const char* e = getenv("GD_NN_FIRST_GRID")) g_first_grid = atoi(e) > 0 ? atoi(e) : g_first_grid;
#define GD_FIRST(C_)                                                                                               \
    hipLaunchKernelGGL(conv3x3_first_kernel<C_>, dim3((unsigned)(blocks < g_first_grid ? blocks : g_first_grid)), dim3(256), lds, s, (const uint16_t*)x,    \
                       (const uint16_t*)weight, (const uint16_t*)bias, (uint16_t*)y, N, H, W, Cout)
    if (Cin == 1) GD_FIRST(1);
    else if (Cin == 2) GD_FIRST(2);
    else if (Cin == 3) GD_FIRST(3);
    else GD_FIRST(4);
#undef GD_FIRST
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

static ConvGeom s2_forward_geom(int Hin, int Win, int pad_lo)
{
    ConvGeom g = {};
    g.Hin = Hin; g.Win = Win;
    g.Hg = g.Hout = (Hin + pad_lo - 2) / 2 + 1;      // pad (pad_lo, 1): floor((Hin + pad_lo + 1 - 3) / 2) + 1
    g.Wg = g.Wout = (Win + pad_lo - 2) / 2 + 1;
    g.sy = g.sx = 2;
    g.osy = g.osx = 1;
    for (int ky = 0; ky < 3; ky++)
        for (int kx = 0; kx < 3; kx++) add_tap(g, ky - pad_lo, kx - pad_lo, ky * 3 + kx);
    return g;
}

// parity class (py, px) of the stride-2 input gradient; false if the class has no pixels
static bool s2_dgrad_geom(ConvGeom& g, int Hin, int Win, int pad_lo, int py, int px)
{
    const int Ho = (Hin + pad_lo - 2) / 2 + 1, Wo = (Win + pad_lo - 2) / 2 + 1;
    g = ConvGeom{};
    g.Hin = Ho; g.Win = Wo;
    g.Hg = (Hin - py + 1) / 2; g.Wg = (Win - px + 1) / 2;
    g.sy = g.sx = 1;
    g.Hout = Hin; g.Wout = Win;
    g.osy = g.osx = 2; g.ooy = py; g.oox = px;
    for (int ky = 0; ky < 3; ky++) {
        if ((py + pad_lo - ky) & 1) continue;
        for (int kx = 0; kx < 3; kx++) {
            if ((px + pad_lo - kx) & 1) continue;
            add_tap(g, (py + pad_lo - ky) / 2, (px + pad_lo - kx) / 2, 8 - (ky * 3 + kx));
        }
    }
    return g.Hg > 0 && g.Wg > 0;
}

static size_t geom_ws_bytes(int N, const ConvGeom& g, int Cin, int Cout)
{
    const int64_t M = (int64_t)N * g.Hg * g.Wg;
    const int split = choose_split(M, Cout, g.ntaps, Cin);
    return split > 1 ? split_ws_bytes(split, M, Cout) : 0;
}

// fp32 scratch the split-K form of the stride-2 layers wants (0: the layer fills the chip unsplit); forward, or the
// input gradient of the same layer when dgrad != 0 (the largest of its parity classes)
size_t gd_nn_conv3x3_s2_ws_bytes(int N, int Hin, int Win, int Cin, int Cout, int pad_lo, int dgrad)
{
    if (N <= 0 || Hin < 2 || Win < 2 || Cin <= 0 || Cout <= 0 || Cin % 4 || Cout % 4) return 0;
    if (!dgrad) return Cin % BK ? 0 : geom_ws_bytes(N, s2_forward_geom(Hin, Win, pad_lo), Cin, Cout);
    if (Cout % BK) return 0;
    size_t need = 0;
    for (int py = 0; py < 2; py++)
        for (int px = 0; px < 2; px++) {
            ConvGeom g;
            if (!s2_dgrad_geom(g, Hin, Win, pad_lo, py, px)) continue;
            const size_t b = geom_ws_bytes(N, g, Cout, Cin);
            need = b > need ? b : need;
        }
    return need;
}

int gd_nn_conv3x3_s2_forward_ws(void* stream, const void* x, const void* weight, const void* bias, void* y, int N,
                                int Hin, int Win, int Cin, int Cout, int pad_lo, void* ws, size_t ws_bytes)
{
    if (!x || !weight || !y) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (N <= 0 || Hin < 2 || Win < 2 || Cin % BK || Cout % 4 || Cin <= 0 || Cout <= 0 || (pad_lo != 0 && pad_lo != 1))
        return fail(GD_NN_ERR_INVALID_ARG, "conv3x3 s2: need Cin % 64 == 0, Cout % 4 == 0, pad_lo in {0, 1}");
    return launch_conv((hipStream_t)stream, x, weight, bias, 0, nullptr, y, N, s2_forward_geom(Hin, Win, pad_lo), Cin, Cout,
                       ws, ws_bytes);
}

int gd_nn_conv3x3_s2_forward(void* stream, const void* x, const void* weight, const void* bias, void* y, int N, int Hin,
                             int Win, int Cin, int Cout, int pad_lo)
{
    return gd_nn_conv3x3_s2_forward_ws(stream, x, weight, bias, y, N, Hin, Win, Cin, Cout, pad_lo, nullptr, 0);
}

int gd_nn_conv3x3_s2_dgrad_ws(void* stream, const void* dy, const void* weight_flipped, void* dx, int N, int Hin, int Win,
                              int Cin, int Cout, int pad_lo, void* ws, size_t ws_bytes)
{
    if (!dy || !weight_flipped || !dx) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (N <= 0 || Hin < 2 || Win < 2 || Cout % BK || Cin % 4 || Cin <= 0 || Cout <= 0 || (pad_lo != 0 && pad_lo != 1))
        return fail(GD_NN_ERR_INVALID_ARG, "conv3x3 s2 dgrad: need Cout % 64 == 0, Cin % 4 == 0, pad_lo in {0, 1}");
    // dx[2a+py, 2b+px] = sum over (ky, kx) with (py + pad - ky), (px + pad - kx) even of
    //                    dy[a + (py + pad - ky)/2, b + (px + pad - kx)/2] . w[:, ky, kx, :]
    // weight_flipped[ci][8 - (3 ky + kx)][co] = w[co][ky][kx][ci]  (gd_nn_conv3x3_flip_weights)
    // One launch for the four classes where the maps are large (round 5); else back to back on one stream, sharing the scratch.
    {
        ConvGeom gs4[4];
        const void* ws4[4];
        int n = 0;
        for (int py = 0; py < 2; py++)
            for (int px = 0; px < 2; px++)
                if (s2_dgrad_geom(gs4[n], Hin, Win, pad_lo, py, px)) ws4[n++] = weight_flipped;
        const int r = launch_conv_multi((hipStream_t)stream, dy, ws4, nullptr, 0, nullptr, dx, N, gs4, n, Cout, Cin, ws != nullptr);
        if (r != 0) return r < 0 ? r : GD_NN_OK;
    }
    for (int py = 0; py < 2; py++)
        for (int px = 0; px < 2; px++) {
            ConvGeom g;
            if (!s2_dgrad_geom(g, Hin, Win, pad_lo, py, px)) continue;
            const int r = launch_conv((hipStream_t)stream, dy, weight_flipped, nullptr, 0, nullptr, dx, N, g, Cout, Cin, ws,
                                      ws_bytes);
            if (r < 0) return r;
        }
    return GD_NN_OK;
}

int gd_nn_conv3x3_s2_dgrad(void* stream, const void* dy, const void* weight_flipped, void* dx, int N, int Hin, int Win,
                           int Cin, int Cout, int pad_lo)
{
    return gd_nn_conv3x3_s2_dgrad_ws(stream, dy, weight_flipped, dx, N, Hin, Win, Cin, Cout, pad_lo, nullptr, 0);
}

int gd_nn_conv3x3_up2_forward(void* stream, const void* x, const void* w_even_rows, const void* w_odd_rows,
                              const void* bias, void* y, int N, int H, int W, int Cin, int Cout)
{
    if (!x || !w_even_rows || !w_odd_rows || !y) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (N <= 0 || H < 1 || W < 1 || Cin % BK || Cout % 4 || Cin <= 0 || Cout <= 0)
        return fail(GD_NN_ERR_INVALID_ARG, "conv3x3 up2: need Cin % 64 == 0, Cout % 4 == 0");
    // y[2i+py, 2j+px] = sum over (ty, tx) in {0,1}^2 of x[i + ty + py - 1, j + tx + px - 1] . Wc[py][px][ty][tx]
    // where Wc sums the 3x3 taps that land on the same source pixel of the nearest-neighbour upsampled image
    // (rows: py = 0 -> {ky=0}, {ky=1,2};  py = 1 -> {ky=0,1}, {ky=2}; columns alike).  Tap (px, ty, tx) of row
    // parity py is filter slot px*4 + ty*2 + tx of w_even_rows / w_odd_rows ([Cout][9][Cin], slot 8 unused).
    ConvGeom gs4[4];
    const void* ws4[4];
    for (int py = 0; py < 2; py++)
        for (int px = 0; px < 2; px++) {
            ConvGeom g = {};
            g.Hin = H; g.Win = W;
            g.Hg = H; g.Wg = W;
            g.sy = g.sx = 1;
            g.Hout = 2 * H; g.Wout = 2 * W;
            g.osy = g.osx = 2; g.ooy = py; g.oox = px;
            for (int ty = 0; ty < 2; ty++)
                for (int tx = 0; tx < 2; tx++) add_tap(g, ty + py - 1, tx + px - 1, px * 4 + ty * 2 + tx);
            gs4[py * 2 + px] = g;
            ws4[py * 2 + px] = py ? w_odd_rows : w_even_rows;
        }
    {   // the four parity classes as one launch (round 5)
        ConvGeom tmp[4] = {gs4[0], gs4[1], gs4[2], gs4[3]};
        const int r = launch_conv_multi((hipStream_t)stream, x, ws4, bias, 0, nullptr, y, N, tmp, 4, Cin, Cout, false);
        if (r != 0) return r < 0 ? r : GD_NN_OK;
    }
    for (int c = 0; c < 4; c++) {
        const int r = launch_conv((hipStream_t)stream, x, ws4[c], bias, 0, nullptr, y, N, gs4[c], Cin, Cout);
        if (r < 0) return r;
    }
    return GD_NN_OK;
}

/* nn.Linear as a one-tap implicit GEMM: y[M][Nout] = x[M][K] . w[Nout][K]^T + bias[Nout] + residual[M][Nout] (bf16). */
int gd_nn_linear_forward(void* stream, const void* x, const void* weight, const void* bias, const void* residual, void* y,
                         int64_t M, int K, int Nout)
{
    if (!x || !weight || !y) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (M <= 0 || M > 0x7fffffff || K <= 0 || K % BK || Nout <= 0 || Nout % 4)
        return fail(GD_NN_ERR_INVALID_ARG, "linear: need K % 64 == 0 and Nout % 4 == 0");
    ConvGeom g = {};
    g.Hin = g.Hg = g.Hout = 1;
    g.Win = g.Wg = g.Wout = (int)M;
    g.sy = g.sx = g.osy = g.osx = 1;
    g.wtaps = 1;
    add_tap(g, 0, 0, 0);
    return launch_conv((hipStream_t)stream, x, weight, bias, 0, residual, y, 1, g, K, Nout);
}

int gd_nn_conv_force_variant(int v)
{
    g_force_variant = v;
    return GD_NN_OK;
}

int gd_nn_conv_profile_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_cprof.mu);
    g_cprof.on = on != 0;
    return GD_NN_OK;
}

int gd_nn_conv_profile_reset(void)
{
    std::lock_guard<std::mutex> lk(g_cprof.mu);
    g_cprof.total_ms = g_cprof.total_flops = g_cprof.total_bytes = 0;
    g_cprof.launches = 0;
    return GD_NN_OK;
}

/* Waits for the recorded events; returns summed kernel time, launches and algorithmic FLOPs
 * (2 * N*H*W * Cout * 9*Cin per launch) since the last reset. */
int gd_nn_conv_profile_read(double* total_ms, int64_t* launches, double* total_flops)
{
    std::lock_guard<std::mutex> lk(g_cprof.mu);
    for (auto& r : g_cprof.pending) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            g_cprof.total_ms += ms;
            g_cprof.launches += 1;
        }
        g_cprof.pool.push_back(r.a);
        g_cprof.pool.push_back(r.b);
    }
    g_cprof.pending.clear();
    if (total_ms) *total_ms = g_cprof.total_ms;
    if (launches) *launches = g_cprof.launches;
    if (total_flops) *total_flops = g_cprof.total_flops;
    return GD_NN_OK;
}

/* Algorithmic HBM bytes (activations in + weights + activations out, each once) summed like total_flops. */
int gd_nn_conv_profile_read_bytes(double* total_bytes)
{
    std::lock_guard<std::mutex> lk(g_cprof.mu);
    if (total_bytes) *total_bytes = g_cprof.total_bytes;
    return GD_NN_OK;
}

// Input gradient of the first convolution (Cin <= 3, Cout = 128; nn_conv_first_dgrad.h).  `wpack`: 32 x 128 bf16 from
// gd_nn_conv3x3_first_dgrad_weights (once per frozen weight); dx4: [N, H, W, 4] bf16 (channel Cin.. = 0).
int gd_nn_conv3x3_first_dgrad_supported(int N, int H, int W, int Cin, int Cout)
{
    return N > 0 && H > 0 && W > 0 && Cin >= 1 && Cin <= 3 && Cout == 128 && (double)H * W * 256.0 < 2147483648.0 &&
           (double)N * ((H + 15) / 16) * ((W + 15) / 16) < 2147483648.0;
}

int gd_nn_conv3x3_first_dgrad_weights(void* stream, const void* weight, void* wpack, int Cout, int Cin)
{
    if (!weight || !wpack) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (Cout != 128 || Cin < 1 || Cin > 3) return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_first_dgrad: need Cout == 128 and 1 <= Cin <= 3");
    hipLaunchKernelGGL(conv3x3_first_dgrad_weights_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)weight,
                       (uint16_t*)wpack, Cin);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

int gd_nn_conv3x3_first_dgrad(void* stream, const void* dy, const void* wpack, void* dx4, int N, int H, int W, int Cin, int Cout)
{
    if (!dy || !wpack || !dx4) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (!gd_nn_conv3x3_first_dgrad_supported(N, H, W, Cin, Cout))
        return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_first_dgrad: need Cout == 128, 1 <= Cin <= 3 and an image below 2 GiB");
    const int tx = (W + 15) / 16, ty = (H + 15) / 16;
    hipLaunchKernelGGL(conv3x3_first_dgrad_kernel, dim3((unsigned)(N * tx * ty)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)dy, (const uint16_t*)wpack, (uint16_t*)dx4, H, W, tx, ty);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

int gd_nn_conv3x3_flip_weights(void* stream, const void* weight, void* flipped, int Cout, int Cin)
{
    if (!weight || !flipped) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    hipLaunchKernelGGL(conv3x3_flip_weights_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)weight, (uint16_t*)flipped, Cout, Cin);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

}  // extern "C"
