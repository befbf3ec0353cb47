// nn_conv_wino.h -- 3x3 / stride 1 / pad 1 convolution with the Winograd F(2,3) minimal-filtering transform
// along x, on the gfx950 matrix cores.  Included by nn_conv3x3.hip inside its anonymous namespace (shares the
// LDS-DMA helpers, the epilogue helpers and the profiling hooks of that file).
//
// WHY.  The direct patch-staged kernels (conv3x3_patch_stream_kernel / conv3x3_gn_patch_kernel) sit at 0.35 of
// the bf16 MFMA roof, 65-80 % of what the matrix pipe sustains on this part at all: an MFMA-only stream runs at
// 1.2-1.45 PFLOP/s because the chip clocks down to its power budget (profiles/r01_conv_ablation.txt).  Feeding
// the same MFMAs faster has been worth 1-3 % per idea for two rounds; the lever left is issuing FEWER of them.
// F(2,3): two neighbouring outputs of a 3-tap filter from 4 multiplications instead of 6,
//     m0 = (d0 - d2) g0            y0 = m0 + m1 + m2
//     m1 = (d1 + d2) (g0+g1+g2)/2  y1 = m1 - m2 - m3
//     m2 = (d2 - d1) (g0-g1+g2)/2
//     m3 = (d1 - d3) g2
// applied along x only: the three kernel rows (ky) stay an ordinary accumulation, so per pair of output pixels
// and input channel the matrix pipe does 3 x 4 instead of 3 x 6 multiply-adds -- 2/3 of the MFMAs, for twice the
// accumulators.  (The 2-D form F(2x2,3x3), 4/9 of the MFMAs, needs FOUR times the accumulators -- 16 position
// planes of a 64 x 64 block fill the whole register file of a CU -- and 2 KB of LDS fragment reads per MFMA at
// the block sizes that fit, above what the LDS delivers per MFMA slot: costed in DESIGN.md, not built.)
//
// DATA FLOW per workgroup (8 waves, tile = 128 output channels x 16x16 pixels, K chunk = 32 input channels):
//   P   raw 18x18 halo'd patch [324 px][32 ch] bf16 by LDS-DMA (out-of-image lanes: hardware zero fill)
//   V   P -> V[pos 0..3][row 0..17][x-pair t 0..7][32 ch]: d0-d2, d1+d2, d2-d1, d1-d3 of the four patch columns
//       2t .. 2t+3 (fp32 arithmetic on the bf16 inputs -- exact -- and ONE rounding to bf16)
//   W   per (ky, chunk) step the slice U[pos][128 co][32 ch] of the transformed filter bank (precomputed once per
//       frozen weight by conv3x3_wino_weights_kernel, cached by the caller like the flipped dgrad weights)
//   M[pos][co][row y][t] += sum_ky sum_ch U[ky][pos][co][ch] * V[pos][y + ky][t][ch]      (32x32x16 bf16 MFMA)
// A wave owns TWO position planes (pp = 0: m0, m1; pp = 1: m2, m3) of a 64-channel x 8-row block: 128 accumulator
// registers, 1 KB of LDS fragment reads per MFMA like the direct 128-channel kernel.  The inverse transform needs
// one plane of the partner wave: y0 = (m0 + m1) + m2, y1 = m1 - (m2 + m3), so wave pp = 0 hands m1 over, pp = 1
// hands m2 over (fp32, through the LDS that held V / W), and each finishes one pixel parity.
//
// LDS images (all 64-byte rows = 32 channels; 16-byte chunk c of a row is stored at chunk c ^ key):
//   V: pos * 9216 + row * 512 + t * 64, key = ((row & 1) << 1) | (t >> 2)   -- an MFMA B fragment is 4 rows x 8 pairs
//      = 2 KB contiguous; the key makes the four lanes of a 16-lane ds_read_b128 service group that share a bank
//      quarter read four different chunks, for every ky shift
//   W: (pos * 128 + co) * 64, key = (co >> 2) & 3; filled by LDS-DMA with the key applied on the source side.
#pragma once

// Timing-only switches of tools/wino_ablate.sh (never defined in a product build; results are wrong by construction):
//   1 no input transform   2 no filter DMA after the first slice   3 no patch DMA after the first chunk
//   4 no MFMAs (loads, transform, fragment reads stay)   5 no partner exchange   6 no output stores
#ifndef GD_WINO_ABLATE
#define GD_WINO_ABLATE 0
#endif

constexpr int kWinoCK = 32;                                   // input channels per K chunk
constexpr int kWinoVStage = 4 * 18 * 8 * kWinoCK * 2;         // 36864
constexpr int kWinoWStage = 4 * 128 * kWinoCK * 2;            // 32768
constexpr int kWinoPBytes = 21 * 1024;                        // 336 pixel rows of 64 B (324 used)
constexpr int kWinoLds = 2 * kWinoVStage + 2 * kWinoWStage + kWinoPBytes;   // 160768

// w [Cout][3][3][Cin] bf16 -> the transformed filter bank in the ORDER THE KERNEL STREAMS IT: for every (128-channel
// block tn, ky, 32-channel chunk c) one contiguous 32 KB slice = the LDS image of that step,
//     u[((tn * 3 + ky) * kc + c)][pos 0..3][row 0..127][16-byte chunk pc 0..3][8]  =  U[pos][tn * 128 + row][ky][c * 32 + (pc ^ key(row)) * 8 + e]
// (key(row) = (row >> 2) & 3: the bank swizzle of the fragment reads, baked in), rows beyond Cout zero.  An LDS-DMA
// instruction of the kernel then moves 1 KB of CONSECUTIVE global memory (8 full cache lines) instead of sixteen
// 64-byte half lines of sixteen different filter rows.  One thread per 16-byte piece.
__global__ __launch_bounds__(256) void conv3x3_wino_weights_kernel(const uint16_t* __restrict__ w,
                                                                   uint16_t* __restrict__ u, int Cout, int Cin)
{
    const int kc = Cin / kWinoCK;
    const int tiles_n = (Cout + 127) / 128;
    const size_t total = (size_t)tiles_n * 3 * kc * 2048;     // 16-byte pieces
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i & 2047);
        const size_t sl = i >> 11;
        const int c = (int)(sl % kc);
        const int ky = (int)((sl / kc) % 3);
        const int tn = (int)(sl / kc / 3);
        const int pc = q & 3, row = (q >> 2) & 127, pos = q >> 9;
        const int co = tn * 128 + row;
        const int ci0 = c * kWinoCK + ((pc ^ ((row >> 2) & 3)) << 3);
        uint16_t o[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            uint16_t v = 0;
            if (co < Cout) {
                const uint16_t* g = w + ((size_t)co * 3 + ky) * 3 * Cin + ci0 + e;
                const float g0 = bf2f(g[0]), g1 = bf2f(g[Cin]), g2 = bf2f(g[2 * Cin]);
                v = pos == 0 ? g[0] : pos == 3 ? g[2 * Cin]
                    : pos == 1 ? f2bf(0.5f * ((g0 + g2) + g1)) : f2bf(0.5f * ((g0 + g2) - g1));
            }
            o[e] = v;
        }
        *(uint4*)(u + i * 8) = *(const uint4*)o;
    }
}

// GN = true: diffusers' ResnetBlock2D front half conv(silu(GroupNorm(x))) in one kernel -- the raw patch chunk goes
// global -> registers -> x * a[n, c] + b[n, c] -> SiLU -> bf16 -> P (a = gamma * rstd, b = beta - mean * a; zero padding
// applied after the transform) instead of the LDS-DMA.
//
// One tile per workgroup, the compiler's own schedule of the step.  Measured and NOT kept (same-box A/B against this
// form, tools/wino_conv_bench.py): persistent workgroups with the patch loader / transform running across
// tile boundaries, the step's LDS-DMA and transform dealt out behind the MFMA groups, fragment reads as inline asm with
// register double buffering and counted waits, filter slices through registers instead of LDS-DMA, a third filter stage
// fetched two steps ahead -- each 0...-15 %.
template <bool GN>
__global__ __launch_bounds__(512) void conv3x3_wino_kernel(
    const uint16_t* __restrict__ in, const uint16_t* __restrict__ uw, const uint16_t* __restrict__ bias,
    int bias_img_stride, const uint16_t* __restrict__ residual, uint16_t* __restrict__ out, int Nimg, int H, int W,
    int Cin, int Cout, const float* __restrict__ mean_rstd, const uint16_t* __restrict__ gamma,
    const uint16_t* __restrict__ beta, int G, int apply_silu, int tiles_n, int tiles_x, int tiles_y, int nwg,
    float* __restrict__ stat_part)
{
    constexpr int BN = 128;
    constexpr int THREADS = 512;
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sV = smem;
    char* sW = smem + 2 * kWinoVStage;
    char* sP = smem + 2 * kWinoVStage + 2 * kWinoWStage;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);    // XCD-contiguous tile order
    const int tn = bid % tiles_n, tm = bid / tiles_n;
    const int tpi = tiles_x * tiles_y;
    const int nimg = tm / tpi, trem = tm - nimg * tpi;
    const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;
    const int y0 = tyi * 16 - 1, x0 = txi * 16 - 1;
    const int n0 = tn * BN;

    const uint32_t row_bytes = (uint32_t)Cin * 2u;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        (void*)in, 0, (int)((uint32_t)Nimg * (uint32_t)(H * W) * row_bytes), 0x00020000);
    const int kc = Cin / kWinoCK;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)uw, 0, (int)((uint32_t)tiles_n * 3u * (uint32_t)kc * (uint32_t)kWinoWStage), 0x00020000);

    // ---- raw patch loader (LDS-DMA): piece q = tid + 512 i -> pixel q >> 2, 16-byte chunk q & 3
    uint32_t p_goff[3];
    uint32_t p_keep = 0;     // GN: bit i = piece i is a pixel inside the image
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int q = tid + THREADS * i;
        const int pix = q >> 2;
        const int py = pix / kPatch, px = pix - py * kPatch;
        const int gy = y0 + py, gx = x0 + px;
        const bool inimg = pix < kPatchPix && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        p_goff[i] = inimg ? (uint32_t)((nimg * H + gy) * W + gx) * row_bytes + (uint32_t)(q & 3) * 16u : kOOB;
        if (inimg) p_keep |= 1u << i;
    }
    auto issueP = [&](int c) {      // GN = false: LDS-DMA, out-of-image lanes are zero-filled by the hardware
        if (GD_WINO_ABLATE == 3) return;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (i == 2 && wave >= 5) continue;    // 1344 pieces = 21 wave-instructions
            bload_lds16(rs_in, p_goff[i], (uint32_t)c * (kWinoCK * 2), sP + (wave * 64 + THREADS * i) * 16);
        }
    };
    u32x4 p_reg[3];
    auto loadP = [&](int c) {       // GN = true: global -> registers
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (i == 2 && wave >= 5) continue;
            p_reg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)p_goff[i],
                                                                                    (int)(c * (kWinoCK * 2)), 0));
        }
    };
    const int cg = GN ? Cin / G : 1;
    const uint32_t sP_off = (uint32_t)(uintptr_t)sP;
    auto storeP = [&](int c) {      // GN = true: normalise + SiLU + round, registers -> P
        float sc[8], sh[8];
        const int ch0 = c * kWinoCK + (tid & 3) * 8;     // (tid + 512 i) & 3 == tid & 3: one channel octet per thread
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
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (i == 2 && wave >= 5) continue;
            const bool keep = (p_keep >> i) & 1u;
            u32x4 v;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float lo = __uint_as_float(p_reg[i][k] << 16), hi = __uint_as_float(p_reg[i][k] & 0xffff0000u);
                lo = lo * sc[2 * k] + sh[2 * k];
                hi = hi * sc[2 * k + 1] + sh[2 * k + 1];
                if (apply_silu) { lo = silu_fast(lo); hi = silu_fast(hi); }
                v[k] = keep ? pack_bf16(lo, hi) : 0u;
            }
            // inline asm (here and in the transform): the compiler cannot tell a plain LDS store from the target of the
            // LDS-DMA in flight (the next filter slice) and would put s_waitcnt vmcnt(0) in front of it
            asm volatile("ds_write_b128 %0, %1" ::"v"(sP_off + (uint32_t)(tid + THREADS * i) * 16u), "v"(v) : "memory");
        }
    };
    // ---- transformed-filter loader (LDS-DMA): the step's slice is one contiguous 32 KB image (weights kernel above)
    const uint32_t w_voff = (uint32_t)tid * 16u;
    auto issueW = [&](int buf, int ky, int c) {
        char* dst = sW + buf * kWinoWStage;
        const uint32_t soff = (uint32_t)((tn * 3 + ky) * kc + c) * (uint32_t)kWinoWStage;
#pragma unroll
        for (int i = 0; i < 4; i++)
            bload_lds16(rs_w, w_voff + (uint32_t)(THREADS * 16 * i), soff, dst + (wave * 64 + THREADS * i) * 16);
    };
    // ---- input transform P -> V: item = (row r, pair t, chunk c); 576 items, thread tid takes item tid, the wave
    // `extra_wave` the last 64
    auto transform_item = [&](char* dstV, int item) {
        const int c = item & 3, t = (item >> 2) & 7, r = item >> 5;
        const char* src = sP + (r * kPatch + 2 * t) * 64 + c * 16;
        u32x4 d[4];
#pragma unroll
        for (int j = 0; j < 4; j++) d[j] = *(const u32x4*)(src + j * 64);
        uint32_t v[4][4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t w0 = d[0][k], w1 = d[1][k], w2 = d[2][k], w3 = d[3][k];
            const float a0 = __uint_as_float(w0 << 16), b0 = __uint_as_float(w0 & 0xffff0000u);
            const float a1 = __uint_as_float(w1 << 16), b1 = __uint_as_float(w1 & 0xffff0000u);
            const float a2 = __uint_as_float(w2 << 16), b2 = __uint_as_float(w2 & 0xffff0000u);
            const float a3 = __uint_as_float(w3 << 16), b3 = __uint_as_float(w3 & 0xffff0000u);
            v[0][k] = pack_bf16(a0 - a2, b0 - b2);
            v[1][k] = pack_bf16(a1 + a2, b1 + b2);
            v[2][k] = pack_bf16(a2 - a1, b2 - b1);
            v[3][k] = pack_bf16(a1 - a3, b1 - b3);
        }
        const int key = ((r & 1) << 1) | (t >> 2);
        // inline asm: the compiler cannot tell a plain LDS store from the target of the LDS-DMA in flight (the next
        // filter slice) and would put s_waitcnt vmcnt(0) in front of it -- the whole DMA latency, every chunk
        const uint32_t o = (uint32_t)(uintptr_t)(dstV + r * 512 + t * 64 + ((c ^ key) << 4));
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const u32x4 pv = {v[p][0], v[p][1], v[p][2], v[p][3]};
            asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(o), "v"(pv), "n"(p * 9216) : "memory");
        }
    };

    // ---- MFMA roles: wave = (pp: position pair, wc: 64-channel half, wq: 8-row half)
    const int pp = wave & 1, wc = (wave >> 1) & 1, wq = wave >> 2;
    const int fk = lane >> 5, fn = lane & 31;
    f32x16 acc[2][2][2];    // [position of the pair][a: 32-channel block][b: 4-row block]
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int k = 0; k < 16; k++) acc[p][a][b][k] = 0.f;
    uint32_t a_rd[2], b_rd[2];
#pragma unroll
    for (int a = 0; a < 2; a++) a_rd[a] = (uint32_t)((wc * 64 + a * 32 + fn) * 64);
#pragma unroll
    for (int b = 0; b < 2; b++) b_rd[b] = (uint32_t)((wq * 8 + b * 4 + (fn >> 3)) * 512 + (fn & 7) * 64);
    const uint32_t a_key = (uint32_t)((fn >> 2) & 3);
    const uint32_t b_keylo = (uint32_t)((fn & 7) >> 2);

    const int nsteps = 3 * kc;
    // prologue
    if (GN) { loadP(0); storeP(0); } else issueP(0);
    issueW(0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    transform_item(sV, tid);
    if (wave == 0) transform_item(sV, 512 + lane);
    __syncthreads();                 // P(0) consumed, V(0) complete
    if (kc > 1) { if (GN) { loadP(1); storeP(1); } else issueP(1); }
    int s = 0;
    for (int c = 0; c < kc; c++) {
        const char* pv = sV + (c & 1) * kWinoVStage;
        char* nv = sV + ((c + 1) & 1) * kWinoVStage;
        for (int ky = 0; ky < 3; ky++, s++) {
            const int bufW = s & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (s + 1 < nsteps && GD_WINO_ABLATE != 2) issueW(bufW ^ 1, ky == 2 ? 0 : ky + 1, ky == 2 ? c + 1 : c);
            if (c + 1 < kc) {
                if (ky == 0) {
                    if (GD_WINO_ABLATE != 1) transform_item(nv, tid);
                    if (GN && c + 2 < kc) loadP(c + 2);
                } else if (ky == 1) {
                    if (wave == (c & 7) && GD_WINO_ABLATE != 1) transform_item(nv, 512 + lane);
                } else if (c + 2 < kc) {       // every wave is past its reads of P(c + 1)
                    if (GN) storeP(c + 2); else issueP(c + 2);
                }
            }
            const char* pw = sW + bufW * kWinoWStage;
            const uint32_t b_key = ((((uint32_t)(fn >> 3) + (uint32_t)ky) & 1u) << 1) | b_keylo;
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                const uint32_t ch = (uint32_t)(kk * 2 + fk);
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    const int pos = 2 * pp + p;
                    bf16x8_t wf[2], vf[2];
#pragma unroll
                    for (int a = 0; a < 2; a++)
                        wf[a] = *(const bf16x8_t*)(pw + pos * (128 * 64) + a_rd[a] + ((ch ^ a_key) << 4));
#pragma unroll
                    for (int b = 0; b < 2; b++)
                        vf[b] = *(const bf16x8_t*)(pv + pos * 9216 + b_rd[b] + ky * 512 + ((ch ^ b_key) << 4));
                    if (GD_WINO_ABLATE == 4) {
#pragma unroll
                        for (int a = 0; a < 2; a++)
#pragma unroll
                            for (int b = 0; b < 2; b++) acc[p][a][b][0] += (float)wf[a][0] * (float)vf[b][1];
                    } else {
#pragma unroll
                        for (int a = 0; a < 2; a++)
#pragma unroll
                            for (int b = 0; b < 2; b++)
                                acc[p][a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[a], vf[b], acc[p][a][b], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- inverse transform.  pp = 0 holds (m0, m1), pp = 1 holds (m2, m3): y0 = (m0 + m1) + m2, y1 = m1 - (m2 + m3).
    // Each wave hands ONE plane to its partner (wave ^ 1) through LDS and keeps the sum of its two.
    __syncthreads();
    if (GD_WINO_ABLATE != 5) {
        char* mine = smem + wave * 16384;
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    f32x4 sd;
#pragma unroll
                    for (int e = 0; e < 4; e++) sd[e] = pp ? acc[0][a][b][4 * q + e] : acc[1][a][b][4 * q + e];
                    *(f32x4*)(mine + ((a * 2 + b) * 4 + q) * 1024 + lane * 16) = sd;
                }
            }
    }
    __syncthreads();
    {
        const char* theirs = smem + (wave ^ 1) * 16384;
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const f32x4 rv = GD_WINO_ABLATE == 5 ? f32x4{0.f, 1.f, 2.f, 3.f}
                                                         : *(const f32x4*)(theirs + ((a * 2 + b) * 4 + q) * 1024 + lane * 16);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float sm = acc[0][a][b][4 * q + e] + acc[1][a][b][4 * q + e];
                        acc[0][a][b][4 * q + e] = pp ? rv[e] - sm : sm + rv[e];
                    }
                }
    }
    __syncthreads();      // the exchange slots are dead: their memory becomes the transposition buffers

    // ---- epilogue: bias (+ residual), one rounding, stores through a wave-private LDS transposition buffer
    // (64 contiguous bytes per pixel and instruction), optional GroupNorm partial sums of the stored values
    char* tr = smem + wave * kTrWave;
    bool ok[2];
    size_t opix[2];
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const int oy = tyi * 16 + wq * 8 + b * 4 + (fn >> 3), ox = txi * 16 + 2 * (fn & 7) + pp;
        ok[b] = oy < H && ox < W;
        opix[b] = ((size_t)nimg * H + oy) * W + ox;
    }
    const uint16_t* bias_n = bias ? bias + (size_t)nimg * bias_img_stride : nullptr;
    const int stat_rows = tpi * 8;
    const int stat_row = (tyi * tiles_x + txi) * 8 + (pp * 2 + wq) * 2 + ((lane >> 4) & 1);
    const bool wide = (Cout & 7) == 0;
#pragma unroll
    for (int a = 0; a < 2; a++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int co = n0 + wc * 64 + a * 32 + 8 * q + 4 * fk;
            const bool cok = co < Cout;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias_n && cok) {
                const uint2 bb = *(const uint2*)(bias_n + co);
                bv[0] = bf2f((uint16_t)(bb.x & 0xffff)); bv[1] = bf2f((uint16_t)(bb.x >> 16));
                bv[2] = bf2f((uint16_t)(bb.y & 0xffff)); bv[3] = bf2f((uint16_t)(bb.y >> 16));
            }
            float2 st = make_float2(0.f, 0.f);
#pragma unroll
            for (int b = 0; b < 2; b++) {
                if (!ok[b] || !cok) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = acc[0][a][b][4 * q + e] + bv[e];
                if (residual) {
                    const uint2 rv = *(const uint2*)(residual + opix[b] * Cout + co);
                    v[0] += bf2f((uint16_t)(rv.x & 0xffff)); v[1] += bf2f((uint16_t)(rv.x >> 16));
                    v[2] += bf2f((uint16_t)(rv.y & 0xffff)); v[3] += bf2f((uint16_t)(rv.y >> 16));
                }
                uint2 o;
                o.x = pack_bf16(v[0], v[1]);
                o.y = pack_bf16(v[2], v[3]);
                if (wide) *(uint2*)(tr + (b * 32 + fn) * kTrRow + 16 * q + 8 * fk) = o;
                else *(uint2*)(out + opix[b] * Cout + co) = o;
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
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int pr = (lane >> 2) + 16 * i, ch = lane & 3;
                const int oy = tyi * 16 + wq * 8 + (pr >> 5) * 4 + ((pr & 31) >> 3), ox = txi * 16 + 2 * (pr & 7) + pp;
                const int co8 = n0 + wc * 64 + a * 32 + 8 * ch;
                if (oy < H && ox < W && co8 < Cout) {
                    const uint4 v = *(const uint4*)(tr + pr * kTrRow + ch * 16);
                    if (GD_WINO_ABLATE != 6 || H < 0) *(uint4*)(out + (((size_t)nimg * H + oy) * W + ox) * Cout + co8) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}
