// nn_conv_wide.h -- direct 3x3 / stride 1 / pad 1 convolution for layers with FEW output channels (Cout <= 128 per
// tile): tile = 128 output channels x (16 rows x 32 columns of pixels).  Included by nn_conv3x3.hip inside its anonymous
// namespace (shares its LDS-DMA helpers, epilogue helpers and profiling hooks).
//
// WHY.  The 128-channel layers of the VAE encoder at 512^2 (eight plain + four GroupNorm-fused launches per SDS step,
// a fifth of the step) run on the 128 x 256-pixel patch kernels at 0.28-0.31 of the bf16 roof, while the same input depth
// (Cin = 128) on the 256-channel x 256-pixel tile reaches 0.42: that tile does 32 MFMAs per wave and barrier instead
// of 16, and moves half the filter bytes per MFMA.  A layer with 128 output channels cannot use a 256-channel tile --
// but it can use a 512-PIXEL one, with the same 8 x 16 accumulator registers per lane (2 channel blocks x 4 pixel blocks):
//   * K chunk = 32 input channels, one step = one kernel ROW (ky) of a chunk: its three taps (kx) share the patch rows they
//     read, 48 MFMAs per wave and barrier;
//   * per step a 24 KB filter slice [kx][128 co][32 ch] (packed in streaming order by conv3x3_wide_weights_kernel: one
//     LDS-DMA instruction moves 1 KB of consecutive memory) + a third of the 39 KB patch chunk: 0.77 KB of LDS-DMA per
//     MFMA and wave against 1.3 KB (128 x 256 direct tile) and 2.4 KB (Winograd form, nn_conv_wino.h);
//   * fragment reads 6 per 8 MFMAs (filter fragment reused by 4 pixel blocks, pixel fragment by 2 channel blocks).
// LDS: 2 patch stages [18 x 34 px][32 ch] (40 KB each incl. padding) + 2 filter stages (24 KB) = 128 KB.
// 64-byte rows: 16-byte chunk c of patch pixel p at chunk c ^ ((p >> 2) & 3), of filter row r at c ^ ((r >> 2) & 3) -- the
// sixteen lanes of a ds_read_b128 service group read sixteen CONSECUTIVE pixels of a patch row (patch_col), whose
// (p & 3, (p >> 2) & 3) pairs are all different whatever the tap shift: conflict free.
//
// GN = true: conv(silu(GroupNorm(x))) of diffusers' ResnetBlock2D -- the raw patch chunk goes global -> registers ->
// x * a[n, c] + b[n, c] -> SiLU -> bf16 -> LDS (a = gamma * rstd, b = beta - mean * a; zero padding after the transform).
#pragma once

constexpr int kWideCK = 32;
constexpr int kWidePW = 34, kWidePix = 18 * kWidePW;          // 16 x 32 tile + halo = 612 patch pixels
constexpr int kWidePStage = 2560 * 16;                        // 40960: five rounds of 512 16-byte pieces (2448 used)
constexpr int kWideWStage = 3 * 128 * kWideCK * 2;            // 24576
constexpr int kWideLds = 2 * kWidePStage + 2 * kWideWStage;   // 131072

// w [Cout][3][3][Cin] bf16 -> per (128-channel block tn, ky, 32-channel chunk c) one contiguous 24 KB slice = the LDS image
// of that step: [kx 0..2][row 0..127][16-byte chunk pc 0..3][8] = w[tn * 128 + row][ky][kx][c * 32 + (pc ^ key(row)) * 8 + e],
// rows beyond Cout zero.  One thread per 16-byte piece.
__global__ __launch_bounds__(256) void conv3x3_wide_weights_kernel(const uint16_t* __restrict__ w, uint16_t* __restrict__ u,
                                                                   int Cout, int Cin)
{
    const int kc = Cin / kWideCK;
    const int tiles_n = (Cout + 127) / 128;
    const size_t total = (size_t)tiles_n * 3 * kc * 1536;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % 1536);
        const size_t sl = i / 1536;
        const int c = (int)(sl % kc);
        const int ky = (int)((sl / kc) % 3);
        const int tn = (int)(sl / kc / 3);
        const int pc = q & 3, row = (q >> 2) & 127, kx = q >> 9;
        const int co = tn * 128 + row;
        const int ci0 = c * kWideCK + ((pc ^ ((row >> 2) & 3)) << 3);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (co < Cout) v = *(const uint4*)(w + (((size_t)co * 3 + ky) * 3 + kx) * Cin + ci0);
        *(uint4*)(u + i * 8) = v;
    }
}

template <bool GN>
__global__ __launch_bounds__(512) void conv3x3_wide_kernel(
    const uint16_t* __restrict__ in, const uint16_t* __restrict__ uw, const uint16_t* __restrict__ bias,
    int bias_img_stride, const uint16_t* __restrict__ residual, uint16_t* __restrict__ out, int Nimg, int H, int W,
    int Cin, int Cout, const float* __restrict__ mean_rstd, const uint16_t* __restrict__ gamma,
    const uint16_t* __restrict__ beta, int G, int apply_silu, int tiles_n, int tiles_x, int tiles_y, int nwg,
    float* __restrict__ stat_part)
{
    constexpr int THREADS = 512;
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sP = smem;
    char* sW = smem + 2 * kWidePStage;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);    // XCD-contiguous tile order
    const int tn = bid % tiles_n, tm = bid / tiles_n;
    const int tpi = tiles_x * tiles_y;
    const int nimg = tm / tpi, trem = tm - nimg * tpi;
    const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;
    const int y0 = tyi * 16 - 1, x0 = txi * 32 - 1;
    const int n0 = tn * 128;

    const uint32_t row_bytes = (uint32_t)Cin * 2u;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        (void*)in, 0, (int)((uint32_t)Nimg * (uint32_t)(H * W) * row_bytes), 0x00020000);
    const int kc = Cin / kWideCK;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)uw, 0, (int)((uint32_t)tiles_n * 3u * (uint32_t)kc * (uint32_t)kWideWStage), 0x00020000);

    // ---- patch loader: piece q = tid + 512 i -> patch pixel q >> 2, 16-byte chunk slot q & 3 (the lane fetches the channel
    // octet that belongs there: swizzle on the source side, the LDS image of an LDS-DMA is lane linear)
    uint32_t p_goff[5];
    uint32_t p_keep = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const int q = tid + THREADS * i;
        const int pix = q >> 2;
        const int py = (pix * 1928) >> 16, px = pix - py * kWidePW;      // pix / 34 for pix < 700, without a 64-bit multiply
        const int gy = y0 + py, gx = x0 + px;
        const bool inimg = (pix < kWidePix) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
        // LDS-DMA: the LDS image is lane linear, so the swizzle picks WHICH octet the lane fetches; GN (registers): the
        // thread keeps octet tid & 3 for all its pieces (one set of scale / shift) and swizzles the LDS address instead
        const int oct = GN ? (q & 3) : (q & 3) ^ ((pix >> 2) & 3);
        p_goff[i] = inimg ? (uint32_t)((nimg * H + gy) * W + gx) * row_bytes + (uint32_t)oct * 16u : kOOB;
        if (inimg) p_keep |= 1u << i;
    }
    auto issueP = [&](int buf, int c) {
        char* dst = sP + buf * kWidePStage;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            if (i == 4 && wave == 7) continue;      // pieces 2496.. are beyond the patch
            bload_lds16(rs_in, p_goff[i], (uint32_t)c * (kWideCK * 2), dst + (wave * 64 + THREADS * i) * 16);
        }
    };
    u32x4 p_reg[3];
    auto loadP = [&](int c, int i0, int i1) {       // pieces i0 .. i1 - 1 (at most three in flight: registers)
#pragma unroll
        for (int i = i0; i < i1; i++) {
            if (i == 4 && wave == 7) continue;
            p_reg[i - i0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)p_goff[i],
                                                                                         (int)(c * (kWideCK * 2)), 0));
        }
    };
    const int cg = GN ? Cin / G : 1;
    const uint32_t sP_off = (uint32_t)(uintptr_t)sP;
    auto storeP = [&](int buf, int c, int i0, int i1) {
        float sc[8], sh[8];
        const int ch0 = c * kWideCK + (tid & 3) * 8;
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
        for (int i = i0; i < i1; i++) {
            if (i == 4 && wave == 7) continue;
            const int q = tid + THREADS * i, pix = q >> 2;
            const bool keep = (p_keep >> i) & 1u;
            u32x4 v;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float lo = __uint_as_float(p_reg[i - i0][k] << 16), hi = __uint_as_float(p_reg[i - i0][k] & 0xffff0000u);
                lo = lo * sc[2 * k] + sh[2 * k];
                hi = hi * sc[2 * k + 1] + sh[2 * k + 1];
                if (apply_silu) { lo = silu_fast(lo); hi = silu_fast(hi); }
                v[k] = keep ? pack_bf16(lo, hi) : 0u;
            }
            // inline asm: the compiler cannot tell a plain LDS store from the target of the LDS-DMA in flight and would put
            // s_waitcnt vmcnt(0) in front of it
            asm volatile("ds_write_b128 %0, %1" ::"v"(sP_off + (uint32_t)(buf * kWidePStage) + (uint32_t)pix * 64u +
                                                       (uint32_t)(((q & 3) ^ ((pix >> 2) & 3)) << 4)), "v"(v) : "memory");
        }
    };
    // ---- filter loader (LDS-DMA): the step's slice is one contiguous 24 KB image
    const uint32_t w_voff = (uint32_t)tid * 16u;
    auto issueW = [&](int buf, int ky, int c) {
        char* dst = sW + buf * kWideWStage;
        const uint32_t soff = (uint32_t)((tn * 3 + ky) * kc + c) * (uint32_t)kWideWStage;
#pragma unroll
        for (int i = 0; i < 3; i++)
            bload_lds16(rs_w, w_voff + (uint32_t)(THREADS * 16 * i), soff, dst + (wave * 64 + THREADS * i) * 16);
    };

    // ---- MFMA roles: wave = (wc: 64-channel half, wr: 4-row group); pixel block b = (row pair b & 1, column half b >> 1)
    const int wc = wave & 1, wr = wave >> 1;
    const int fk = lane >> 5, fn = lane & 31;
    int rr, fx;
    patch_col(fn, rr, fx);
    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc[a][b][k] = 0.f;
    int p_base[4];      // patch pixel of the lane's pixel of block b for tap (0, 0)
#pragma unroll
    for (int b = 0; b < 4; b++) p_base[b] = (4 * wr + 2 * (b & 1) + rr) * kWidePW + 16 * (b >> 1) + fx;
    uint32_t a_rd[2];
#pragma unroll
    for (int a = 0; a < 2; a++) a_rd[a] = (uint32_t)((wc * 64 + a * 32 + fn) * 64);
    const uint32_t a_key = (uint32_t)((fn >> 2) & 3);

    const int nsteps = 3 * kc;
    if (GN) { loadP(0, 0, 3); storeP(0, 0, 0, 3); loadP(0, 3, 5); storeP(0, 0, 3, 5); } else issueP(0, 0);
    issueW(0, 0, 0);
    int s = 0;
    for (int c = 0; c < kc; c++) {
        const char* pa = sP + (c & 1) * kWidePStage;
        for (int ky = 0; ky < 3; ky++, s++) {
            const int bufW = s & 1;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            if (s + 1 < nsteps) issueW(bufW ^ 1, ky == 2 ? 0 : ky + 1, ky == 2 ? c + 1 : c);
            if (c + 1 < kc) {       // the other patch stage was last read in chunk c - 1
                if (GN) {       // three pieces, then two: twelve registers in flight
                    if (ky == 0) loadP(c + 1, 0, 3);
                    else if (ky == 1) { storeP((c + 1) & 1, c + 1, 0, 3); loadP(c + 1, 3, 5); }
                    else storeP((c + 1) & 1, c + 1, 3, 5);
                } else if (ky == 0) {
                    issueP((c + 1) & 1, c + 1);
                }
            }
            const char* pw = sW + bufW * kWideWStage;
#pragma unroll 1
            for (int kx = 0; kx < 3; kx++) {      // (not unrolled: unrolled, the LDS-DMA variant spills 38 registers)
                uint32_t p_rd[4], p_sw[4];
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int p = p_base[b] + ky * kWidePW + kx;
                    p_rd[b] = (uint32_t)p * 64u;
                    p_sw[b] = (uint32_t)((p >> 2) & 3);
                }
#pragma unroll
                for (int kk = 0; kk < 2; kk++) {
                    const uint32_t ch = (uint32_t)(kk * 2 + fk);
                    bf16x8_t wf[2], pf[4];
#pragma unroll
                    for (int a = 0; a < 2; a++)
                        wf[a] = *(const bf16x8_t*)(pw + kx * (128 * 64) + a_rd[a] + ((ch ^ a_key) << 4));
#pragma unroll
                    for (int b = 0; b < 4; b++) pf[b] = *(const bf16x8_t*)(pa + p_rd[b] + ((ch ^ p_sw[b]) << 4));
#pragma unroll
                    for (int a = 0; a < 2; a++)
#pragma unroll
                        for (int b = 0; b < 4; b++)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[a], pf[b], acc[a][b], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: bias (+ residual), one rounding, stores through a wave-private LDS transposition buffer (64 contiguous
    // bytes per pixel and instruction), optional GroupNorm partial sums of the stored values: per (image, channel quad)
    // tiles * 16 rows of {sum, sum of squares} over 32 pixels each -- (H / 16) * (W / 32) * 16 = the (H / 16) * (W / 16) * 8 rows
    // of the 16 x 16-pixel kernels, so gd_nn_groupnorm_finish_partials and the callers' buffers are unchanged
    __syncthreads();
    char* tr = smem + wave * kTrWave;
    const uint16_t* bias_n = bias ? bias + (size_t)nimg * bias_img_stride : nullptr;
    const int stat_rows = ((H + 15) / 16) * ((W + 15) / 16) * 8;
#pragma unroll
    for (int bh = 0; bh < 2; bh++) {        // column half: pixel blocks b = 2 bh, 2 bh + 1 (the two row pairs)
        bool ok[2];
        size_t opix[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int oy = tyi * 16 + 4 * wr + 2 * j + rr, ox = txi * 32 + 16 * bh + fx;
            ok[j] = oy < H && ox < W;
            opix[j] = ((size_t)nimg * H + oy) * W + ox;
        }
        // the 16-column half (txi, bh) IS tile 2 txi + bh of the 16 x 16-pixel kernels' grid: same eight rows per tile
        const int t16x = (W + 15) / 16, cx = 2 * txi + bh;
        const int stat_row = (tyi * t16x + cx) * 8 + wr * 2 + ((lane >> 4) & 1);
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
                for (int j = 0; j < 2; j++) {
                    if (!ok[j] || !cok) continue;
                    const int b = 2 * bh + j;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = acc[a][b][4 * q + e] + bv[e];
                    if (residual) {
                        const uint2 rv = *(const uint2*)(residual + opix[j] * Cout + co);
                        v[0] += bf2f((uint16_t)(rv.x & 0xffff)); v[1] += bf2f((uint16_t)(rv.x >> 16));
                        v[2] += bf2f((uint16_t)(rv.y & 0xffff)); v[3] += bf2f((uint16_t)(rv.y >> 16));
                    }
                    uint2 o;
                    o.x = pack_bf16(v[0], v[1]);
                    o.y = pack_bf16(v[2], v[3]);
                    *(uint2*)(tr + (j * 32 + fn) * kTrRow + 16 * q + 8 * fk) = o;
                    if (stat_part) stat_accumulate(st, o);
                }
                if (stat_part) {
                    const float sx = row16_sum(st.x), sy = row16_sum(st.y);
                    if ((lane & 15) == 0 && cok && cx < t16x)
                        *(float2*)(stat_part + (((size_t)nimg * (Cout >> 2) + (co >> 2)) * stat_rows + stat_row) * 2) =
                            make_float2(sx, sy);
                }
            }
            // LDS operations of one wave execute in order: the reads below see the writes above
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int pr = (lane >> 2) + 16 * i, ch = lane & 3;
                int rr2, fx2;
                patch_col(pr & 31, rr2, fx2);
                const int oy = tyi * 16 + 4 * wr + 2 * (pr >> 5) + rr2, ox = txi * 32 + 16 * bh + fx2;
                const int co8 = n0 + wc * 64 + a * 32 + 8 * ch;
                if (oy < H && ox < W && co8 < Cout) {
                    const uint4 v = *(const uint4*)(tr + pr * kTrRow + ch * 16);
                    *(uint4*)(out + (((size_t)nimg * H + oy) * W + ox) * Cout + co8) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}
