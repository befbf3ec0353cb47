// nn_conv_regw.h -- 3x3 / stride 1 / pad 1 convolution for 128 -> 128 channels with the WHOLE FILTER BANK IN REGISTERS.
// Included by nn_conv3x3.hip inside its anonymous namespace (shares its LDS-DMA helpers and profiling hooks).
//
// WHY.  The 128 -> 128 layers of the VAE encoder at 512^2 (eight launches per SDS step, 6.7 ms) are the slowest big
// shape of the step: 0.33 of the bf16 roof on the wide tile (nn_conv_wide.h), at the socket's power cap.  Every tile
// form so far streams the 295 KB filter bank through LDS once per TILE (LDS-DMA + one filter-fragment read per two
// MFMAs) because it does not fit the 160 KB of LDS -- but it does fit the CU's 512 KB REGISTER file: eight waves (two per
// SIMD, 256 registers each) hold 32 output channels x 576 K = 36 MFMA A-fragments = 144 registers apiece (128 of them AGPRs), loaded once.
//   wave = (nt: output channels 32 nt .., kh: input channels 64 kh .. of all nine taps)
// What is left per output row is the input: a workgroup walks DOWN a 32-pixel-wide column strip; each new output row
// needs ONE new input row (34 px x 128 ch = 8.5 KB into a six-row LDS ring), and every MFMA takes its B operand
// (32 pixels x 16 channels) from the ring with one ds_read_b128 whose address is a per-row base + an immediate:
//   * per output row and wave 36 v_mfma_f32_32x32x16_bf16 + 36 ds_read_b128 (half the LDS's 256 B/clk per CU), ONE
//     workgroup barrier, no filter traffic at all, no address arithmetic;
//   * the two waves of a SIMD split K, so one can issue while the other waits for the matrix pipe or the LDS (a lone
//     wave per SIMD with all 72 fragments measured 1.5x slower: MFMA issue, fragment reads and epilogue ADD UP in one
//     in-order instruction stream).  The K halves meet through LDS: the kh = 1 wave writes its 32 x 32 fp32 partial tile,
//     the kh = 0 wave adds it one row later, in the shadow of the next row's MFMAs, and runs the epilogue; the kh = 1
//     waves fetch the look-ahead row instead.  Only they wait on vmcnt, so no wave ever waits for a store acknowledgement.
// Ring layout: pixel stride 272 B (256 of data + 16 of padding): the sixteen lanes of a ds_read_b128 service group read
// sixteen consecutive pixels, 17 x 16 B apart = sixteen different bank quads whatever the tap shift -> conflict free, and
// the tap column (kx * 272) and the K step (cs * 32) are instruction immediates.  One LDS-DMA instruction (4 B per lane)
// moves one pixel's 256 contiguous bytes.
#pragma once

#ifndef GD_REGW_ABLATE     // timing-only builds (tools/regw_variants.sh), bit mask: 1 no global stores, 2 no workgroup barrier,
#define GD_REGW_ABLATE 0   // 4 no look-ahead row, 8 no MFMAs, 16 no fragment reads, 32 no epilogue
#endif

// The compiler splits a 256-register wave evenly, 128 VGPRs + 128 AGPRs: 32 of the wave's 36 filter fragments are AGPR
// operands of their MFMAs, four live in VGPRs.
constexpr int kRegwAgprFrags = 32;
constexpr int kRegwRing = 6;
constexpr int kRegwPixB = 272;                            // bytes per ring pixel
constexpr int kRegwRowB = 36 * kRegwPixB;                 // 9792: 34 patch pixels + the two pad pixels the loader's last round covers
constexpr int kRegwTrB = 32 * 80;                         // epilogue transposition, per (row parity, nt): 32 pixels x (64 B + pad)
constexpr int kRegwPartB = 4 * 64 * 16;                   // a partial tile: [4 register quads][64 lanes][16 B]
constexpr int kRegwOffTr = kRegwRing * kRegwRowB;         // 58752
constexpr int kRegwOffPart = kRegwOffTr + 2 * 4 * kRegwTrB;   // 79232: [2 row parities][4 nt] partial tiles
constexpr int kRegwLds = kRegwOffPart + 2 * 4 * kRegwPartB;      // 112000

// w [128][3][3][128] bf16 -> u [wave 0..7][f 0..35][lane 0..63][8]: A fragment f = tap * 4 + c (32 output channels x 16 K)
// of wave (nt = wave & 3, kh = wave >> 2): lane l holds w[nt * 32 + (l & 31)][tap][kh * 64 + c * 16 + (l >> 5) * 8 + e]
__global__ __launch_bounds__(256) void conv3x3_regw_weights_kernel(const uint16_t* __restrict__ w, uint16_t* __restrict__ u)
{
    const int i = blockIdx.x * 256 + threadIdx.x;      // one 16-byte piece
    if (i >= 8 * 36 * 64) return;
    const int lane = i & 63, f = (i >> 6) % 36, wv = (i >> 6) / 36;
    const int tap = f >> 2, c = f & 3, nt = wv & 3, kh = wv >> 2;
    const int co = nt * 32 + (lane & 31), ci0 = kh * 64 + c * 16 + (lane >> 5) * 8;
    *(uint4*)(u + (size_t)i * 8) = *(const uint4*)(w + ((size_t)co * 9 + tap) * 128 + ci0);
}

template <bool STAT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_regw_kernel(
    const uint16_t* __restrict__ in, const uint16_t* __restrict__ uw, const uint16_t* __restrict__ bias,
    int bias_img_stride, uint16_t* __restrict__ out, int Nimg, int H, int W,
    int strips, int segs, int seg_rows, int nwg, float* __restrict__ stat_part)
{
    constexpr int C = 128;
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = wave & 3, kh = wave >> 2;
    int bid = blockIdx.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);    // XCD-contiguous order: neighbours share halo columns
    const int sx = bid % strips, rest = bid / strips;
    const int sg = rest % segs, nimg = rest / segs;
    const int x0 = sx * 32 - 1;                 // image x of patch pixel 0
    const int yb = sg * seg_rows;               // first output row of this segment
    const int rows = min(seg_rows, H - yb);
    if (rows <= 0) return;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        (void*)in, 0, (int)((uint32_t)Nimg * (uint32_t)(H * W) * (uint32_t)(C * 2)), 0x00020000);

    // ---- the wave's share of the filter bank: 36 A fragments, resident for the whole launch
    bf16x8_t wreg[36];
    {
        const uint4* src = (const uint4*)uw + ((size_t)wave * 36) * 64 + lane;
#pragma unroll
        for (int f = 0; f < 36; f++) wreg[f] = __builtin_bit_cast(bf16x8_t, src[f * 64]);
    }

    // ---- input-row loader (the kh = 1 waves; in the prologue all eight).  Plain form: one LDS-DMA instruction (4 bytes per
    // lane) moves ONE pixel's 256 bytes; loader wave lw fetches patch pixels lw, lw + 4, ..., lw + 32 of a row (nine
    // instructions; pixels 34, 35 are padding and, like everything outside the image, read as zeros: buffer range check)
    const int lw = nt;
    uint32_t l_first, l_mid, l_last;     // the lane's byte offset inside an image row for round 0, rounds 1..7, round 8 (or kOOB)
    {
        const int g0 = x0 + lw, g8 = x0 + lw + 32;
        // rounds 1..7 (always inside the image): pixel x0 + lw + 4 (>= 3: the vector offset must not wrap, it is what the range
        // check sees) + 1024 (i - 1) through the scalar offset
        l_mid = (uint32_t)(x0 + lw + 4) * (C * 2) + (uint32_t)lane * 4u;
        l_first = (unsigned)g0 < (unsigned)W ? (uint32_t)g0 * (C * 2) + (uint32_t)lane * 4u : kOOB;
        l_last = (lw + 32 < 34 && (unsigned)g8 < (unsigned)W) ? (uint32_t)g8 * (C * 2) + (uint32_t)lane * 4u : kOOB;
    }
    auto row_goff = [&](int prow) -> uint32_t {            // patch row prow <-> image row yb - 1 + prow
        const int gy = yb - 1 + prow;
        return (unsigned)gy < (unsigned)H ? (uint32_t)((nimg * H + gy) * W) * (C * 2) : kOOB;
    };
    auto issue_row = [&](int prow) {
        const uint32_t base = row_goff(prow);
        char* dst = smem + (prow % kRegwRing) * kRegwRowB + lw * kRegwPixB;
        const bool rok = base != kOOB;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const uint32_t v = i == 0 ? l_first : i == 8 ? l_last : l_mid;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void*)(dst + 4 * i * kRegwPixB), 4,
                                                     rok ? v : kOOB, rok ? base + (i >= 1 && i <= 7 ? 1024u * (i - 1) : 0u) : 0u, 0, 0);
        }
    };
    // ---- B-fragment addressing: lane -> pixel column fn = lane & 31 (patch pixel fn + kx), K half fk = lane >> 5
    const int fn = lane & 31, fk = lane >> 5;
    const uint32_t b_lane = (uint32_t)fn * kRegwPixB + (uint32_t)fk * 16u + (uint32_t)kh * 128u;   // + ring row base; kx, cs: immediates

    // bias of the lane's 16 output channels (co = nt * 32 + 8 a + 4 fk + e), kept packed: two bf16 per register (kh = 0 waves)
    uint2 bq[4];
#pragma unroll
    for (int a = 0; a < 4; a++) bq[a] = make_uint2(0u, 0u);
    if (bias && kh == 0) {
        const uint16_t* bias_n = bias + (size_t)nimg * bias_img_stride;
#pragma unroll
        for (int a = 0; a < 4; a++) bq[a] = *(const uint2*)(bias_n + nt * 32 + 8 * a + 4 * fk);
    }

    // ---- prologue: patch rows 0 .. 3 (rows 0-2 feed output row 0; row 3 is the first look-ahead), two rows per wave set
    issue_row(2 * kh);
    issue_row(2 * kh + 1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    const int t16x = (W + 15) / 16;
    const int stat_rows = ((H + 15) / 16) * t16x * 8;
    const bool xok = sx * 32 + fn < W;
    f32x16 res;
    // ---- epilogue of output row r, spread over the two waves of a SIMD and over the MFMA groups of the NEXT rows:
    //   iteration r + 1, kh = 0: res (own half) += the kh = 1 half; part a (0..3): eight output channels -- bias, one
    //                            rounding, into transposition buffer r & 1 (+ statistics)
    //   iteration r + 2, kh = 0: store i (0..1): sixteen pixels x 64 bytes read back transposed, global stores (one row later
    //                            so that the transposition writes have long landed; buffer r & 1 is written again in iteration r + 3)
    // The kh = 1 waves issue no stores: the look-ahead row's LDS-DMA instructions stay the youngest vector-memory operations
    // of their wave, so their vmcnt wait never covers this iteration's traffic (with the stores behind the look-ahead in the
    // same wave the row time was bounded below by the DMA latency: 623 -> 556 us without them, tools/regw_variants.sh)
    auto tr_ptr = [&](int r) { return smem + kRegwOffTr + ((r & 1) * 4 + nt) * kRegwTrB; };
    const size_t row_elems = (size_t)W * C;
    uint16_t* out_p = out + (((size_t)nimg * H + yb) * W + sx * 32 + (lane >> 2)) * C + nt * 32 + 8 * (lane & 3);
    auto epi_part = [&](int r, int a) {          // r = row inside the segment
        float v[4];
        v[0] = res[4 * a + 0] + __uint_as_float(bq[a].x << 16);
        v[1] = res[4 * a + 1] + __uint_as_float(bq[a].x & 0xffff0000u);
        v[2] = res[4 * a + 2] + __uint_as_float(bq[a].y << 16);
        v[3] = res[4 * a + 3] + __uint_as_float(bq[a].y & 0xffff0000u);
        uint2 o;
        o.x = pack_bf16(v[0], v[1]);
        o.y = pack_bf16(v[2], v[3]);
        *(uint2*)(tr_ptr(r) + fn * 80 + 16 * a + 8 * fk) = o;
        if (STAT) {
            // {sum, sum of squares} of the stored values over the 16 pixels of a half row, per channel quad.  The 16 x 16-pixel
            // kernels' partial layout has 8 rows per tile = pixel-row PAIRS; this kernel finishes one pixel row at a time, so
            // the odd row of a pair adds to what the even row (same lane, previous iteration) wrote
            const int oy = yb + r;
            float2 st = make_float2(0.f, 0.f);
            if (xok) stat_accumulate(st, o);
            const float s0 = row16_sum(st.x), s1 = row16_sum(st.y);
            if ((lane & 15) == 0) {
                const int cx = 2 * sx + ((lane >> 4) & 1);
                if (cx < t16x) {
                    const int co = nt * 32 + 8 * a + 4 * fk;
                    float2* dst = (float2*)(stat_part + (((size_t)nimg * (C >> 2) + (co >> 2)) * stat_rows +
                                                         ((oy >> 4) * t16x + cx) * 8 + ((oy & 15) >> 1)) * 2);
                    if (oy & 1) { const float2 pv = *dst; *dst = make_float2(pv.x + s0, pv.y + s1); }
                    else *dst = make_float2(s0, s1);
                }
            }
        }
    };
    auto epi_store = [&](int r, int i) {
        const int pr = (lane >> 2) + 16 * i, ch = lane & 3;
        if (sx * 32 + pr < W) {
            const uint4 v = *(const uint4*)(tr_ptr(r) + pr * 80 + ch * 16);
            if (!(GD_REGW_ABLATE & 1) || v.x == 0x12345678u) *(uint4*)(out_p + (size_t)r * row_elems + (size_t)(16 * i) * C) = v;
        }
    };
    // the other K half of row r: [parity r & 1][nt][register quad][lane] -- written by the kh = 1 wave at the end of iteration r,
    // read by the kh = 0 wave in iteration r + 1 (the row barrier lies between; the next write to this parity is two barriers on)
    auto part_ptr = [&](int r) { return smem + kRegwOffPart + ((r & 1) * 4 + nt) * kRegwPartB + lane * 16; };

    for (int y = 0; y < rows + 2; y++) {
        const bool have_part = y >= 1 && y <= rows;          // kh = 0: epilogue parts of row y - 1
        const bool have_store = y >= 2;                       // kh = 1: stores of row y - 2
        if (have_part && kh == 0) {          // res = own half (kept from the last iteration) + the kh = 1 wave's half
            const char* pp = part_ptr(y - 1);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 t = *(const float4*)(pp + q * 1024);
                res[4 * q + 0] += t.x; res[4 * q + 1] += t.y; res[4 * q + 2] += t.z; res[4 * q + 3] += t.w;
            }
        }
        if (y >= rows) {        // the last rows' epilogue stages have no MFMAs to hide behind
            if (kh == 0 && have_part) {
#pragma unroll
                for (int a = 0; a < 4; a++) epi_part(y - 1, a);
            }
            if (kh == 0 && have_store) { epi_store(y - 2, 0); epi_store(y - 2, 1); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            continue;
        }
        const bool ahead = y + 4 < rows + 2 && !(GD_REGW_ABLATE & 4);     // look-ahead: patch row y + 4 (read from iteration y + 2 on)
        // output row y: 36 MFMAs per wave on patch rows y, y + 1, y + 2, in nine groups (= taps) of four K steps.  Hand-pipelined:
        // the B fragments of tap t + 1 are in flight while the MFMAs of tap t run (LDS returns in order: lgkmcnt(4) =
        // everything but the four youngest LDS operations has landed; the SIMD's other wave covers the rest of the latency; operations the compiler adds in between only make
        // that wait stricter).  Inline asm because the compiler serialises ds_read -> s_waitcnt lgkmcnt(0) -> MFMA on one
        // fragment register, and keeps the filter fragments in VGPRs / copies them from AGPRs before every use.
        {
            const uint32_t rb[3] = {lds0 + (uint32_t)((y % kRegwRing) * kRegwRowB) + b_lane,
                                    lds0 + (uint32_t)(((y + 1) % kRegwRing) * kRegwRowB) + b_lane,
                                    lds0 + (uint32_t)(((y + 2) % kRegwRing) * kRegwRowB) + b_lane};
            bf16x8_t pf[2][4];
            f32x16 acc;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (!(GD_REGW_ABLATE & 16))
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pf[0][j]) : "v"(rb[0]), "n"(32 * j) : "memory");
#pragma unroll
            for (int t = 0; t < 9; t++) {
                if (t + 1 < 9) {
                    const int t1 = t + 1;
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (!(GD_REGW_ABLATE & 16))
                            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pf[t1 & 1][j]) : "v"(rb[t1 / 3]), "n"((t1 % 3) * kRegwPixB + 32 * j) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int f = t * 4 + j;
                    if ((GD_REGW_ABLATE & 8) && f > 0) continue;
                    // "=&v": an MFMA reads its sources over several passes -- the result must not share registers with them
                    if (f == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(wreg[f]), "v"(pf[t & 1][j]));
                    else if (f < kRegwAgprFrags) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(wreg[f]), "v"(pf[t & 1][j]));
                    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wreg[f]), "v"(pf[t & 1][j]));
                }
                // side jobs in the shadow of the tap's MFMAs
                if (!(GD_REGW_ABLATE & 32)) {
                    if (kh == 0) {
                        if (have_part && t < 4) epi_part(y - 1, t);
                        if (have_store && t == 5) epi_store(y - 2, 0);
                        if (have_store && t == 6) epi_store(y - 2, 1);
                    } else if (t == 1 && ahead) {
                        issue_row(y + 4);
                    }
                } else if (kh == 1 && t == 1 && ahead) {
                    issue_row(y + 4);
                }
            }
            // the accumulator is read by VALU / LDS instructions next: the matrix pipe needs its passes + write-back first
            asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc));
            if (kh == 0) {
#pragma unroll
                for (int k = 0; k < 16; k++) res[k] = acc[k];
            } else {
                char* pp = part_ptr(y);
#pragma unroll
                for (int q = 0; q < 4; q++)
                    *(float4*)(pp + q * 1024) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            }
        }
        if (kh == 1) {
            // patch row y + 3 (issued one iteration ago) must have landed before anyone passes the barrier; this iteration's
            // look-ahead (the nine youngest vector-memory instructions of the wave; loads return in order) may stay in flight
            // (<= 9 outstanding cannot include a piece of row y + 3 AND the nine younger loads)
            if (!ahead) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(GD_REGW_ABLATE & 2)) __syncthreads();
    }
}
