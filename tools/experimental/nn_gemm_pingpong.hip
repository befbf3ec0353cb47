// MEASURED, NOT ROUTED (round 6): csrc/nn_gemm.hip with its K loop as a ping-pong of the two waves of each SIMD (memory segment /
// compute segment, four workgroup barriers per K tile, waves 4..7 one barrier late).  Correct (the 14 parity cases), and within
// +-3 % of the single-phase loop on every shape of the step (profiles/r06_gemm_shapes.txt): with all of a wave's DMA issue in its
// memory segment that segment is ~870 cycles against 512 for the 16 MFMAs of the compute segment, so the loop becomes
// memory-segment bound instead of sum-bound -- the same time.  Build: tools/gemm_variants.sh "pp:-DGD_GEMM_PIPE=1" (from this file).
// nn_gemm.hip -- y[M][N] = x[M][K] . W[N][K]^T (+ bias) (+ residual | GEGLU) in bf16 with fp32 accumulation: a GEMM
// designed as a GEMM for gfx950 (the transformer linears of the SD-2.1 UNet that diffusers hands to cuBLAS and PyTorch-ROCm
// to hipBLASLt; call site Garment_3DGS/threestudio/models/guidance/stable_diffusion_guidance.py:153-157 -> diffusers
// BasicTransformerBlock: attn to_q/to_k/to_v/to_out, FeedForward GEGLU projection and output projection).
//
// Rounds 1-5 ran these products on the library because the one-tap form of the convolution kernel (two LDS stages,
// vmcnt(0) + __syncthreads() per 64-deep K step) reaches 70-90 % of it.  This kernel is a different structure:
//
//   * PERSISTENT workgroups (one per CU, 8 wave64) walk 256 x 256 output tiles; the K loop of a tile is a sequence of
//     64-deep K tiles and the sequence simply continues into the next output tile, so the first operands of tile i + 1
//     are in flight under the last MFMAs and the epilogue stores of tile i.
//   * BOTH operands stream through LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR staging) in HALF tiles of
//     128 rows x 64 K (16 KiB): a ring of TEN slots = all 160 KiB of the CU -- W (the MFMA A operand: output channels)
//     double-buffered (2 K tiles x 2 halves), x (the B operand: rows) TRIPLE-buffered (3 x 2).  One half tile is issued
//     per 16-deep sub-step:  kk = 0, 1 -> W halves of K tile t + 1;  kk = 2, 3 -> x halves of K tile t + 2.  Every wait is
//     a counted `s_waitcnt vmcnt(4)` (gfx9 retires VMEM operations in issue order through one counter): "at most the two
//     x half tiles issued last are still in flight" == K tile t has landed.  ONE bare s_barrier per K tile (never
//     __syncthreads(), whose release fence lowers to vmcnt(0) and would drain the ring).
//   * wave (wc, wp) of the 2 x 4 wave grid owns 128 channels x 64 rows = 4 x 2 MFMA tiles of v_mfma_f32_32x32x16_bf16
//     (128 accumulator registers), 6 ds_read_b128 per 8 MFMAs; the LDS image is the convolution kernels' 256-byte-line
//     XOR swizzle applied on the SOURCE side of the DMA (conflict-free fragment reads).
//   * XCD-aware order: the 32 tiles a pass gives one XCD are consecutive in (row tile, channel tile) order, so a row tile
//     of x is fetched from HBM once per XCD and the weights stay in that XCD's L2.
//   * epilogues the library cannot fuse: bias, bias + residual (second rounding exactly where the eager bf16 op sequence
//     `linear` then `add` has it), and diffusers' GEGLU (hidden * gelu(gate), nn_math.h: the arithmetic and rounding
//     points of the separate geglu_kernel) with hidden / gate rows interleaved per wave so that both halves of a pair
//     sit in the same lane -- the [M][2 inner] projection output is neither written nor read.
//   * the summation order of an output element depends on K only -- never on M, on the tile a row falls in or on how
//     many workgroups run -- so a rank holding 1/k of the rows reproduces the single-rank bits without the k-fold padded
//     row set the library needs for that (nn_ops.route_rows).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gd_nn.h"
#include "../../garmentdreamer_amd/csrc/nn_math.h"

// Timing-only switches of tools/gemm_variants.sh (never defined in a product build; results are wrong with any of them):
//   GD_GEMM_ABLATE=1  no LDS-DMA inside the K loop      2  no fragment reads (stale registers)      3  no MFMAs
//   4  no per-K-tile wait + barrier      5  barrier but no vmcnt wait (is the loop waiting for operands to land?)
#ifndef GD_GEMM_ABLATE
#define GD_GEMM_ABLATE 0
#endif
//   GD_GEMM_PIPE=0    the round's first K loop (one phase per 16-deep step, every wave issues DMA, reads and MFMAs in turn);
//                     correct results: the A/B of the ping-pong loop
#ifndef GD_GEMM_PIPE
#define GD_GEMM_PIPE 1
#endif
//   GD_GEMM_ORDER=1   consecutive tiles share the CHANNEL tile and walk the row tiles (default 0: share the row tile)
#ifndef GD_GEMM_ORDER
#define GD_GEMM_ORDER 0
#endif

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

constexpr int kBK = 64;                    // K tile (128-byte rows)
constexpr int kHalf = 128;                 // rows of a half tile
constexpr int kSlot = kHalf * kBK * 2;     // 16384 B
constexpr int kSlots = 10;                 // W: slots 0..3 = [K-tile parity][half]; x: slots 4..9 = [K tile mod 3][half]
constexpr int kLds = kSlots * kSlot;       // 163840 B = the CU's LDS
constexpr int kThreads = 512;
constexpr uint32_t kOOB = 0x80000000u;     // voffset that fails the buffer range check (every tensor here is < 2 GiB)

__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi)
{
    f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void bload_lds16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff, char* lds_wave_base)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff,
                                             0, 0);
}
// byte offset of logical (row, 16-B chunk j) inside a swizzled [128][64] bf16 half-tile image (nn_conv3x3.hip's layout)
__device__ __forceinline__ int swz(int row, int j)
{
    return (row >> 1) * 256 + (((((row & 1) << 3) | j) ^ ((row >> 1) & 15)) << 4);
}

// row offset + channel offset, out of range if either part is (kOOB + kOOB would wrap around to a valid address)
__device__ __forceinline__ uint32_t addr2(uint32_t row_off, uint32_t ch_off)
{
    return ((row_off | ch_off) & kOOB) ? kOOB : row_off + ch_off;
}

enum { kModeBias = 0, kModeGeglu = 1 };

// N: output channels (MODE 0) / inner width of the GEGLU (MODE 1: W has 2 N rows, hidden then gate).  ldy: row pitch of y
// and of residual, in elements.  tiles_n: channel tiles per row tile (256 channels; GEGLU: 128 output channels).
template <int MODE>
__global__ __launch_bounds__(kThreads) void gemm256_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                           const uint16_t* __restrict__ bias,
                                                           const uint16_t* __restrict__ residual, uint16_t* __restrict__ y,
                                                           int M, int K, int N, int ldy, int tiles_n, int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 1, wp = wave >> 1;           // W half; x quarter (half wp >> 1, rows (wp & 1) * 64 .. + 63 of it)
    const int frow = lane & 31, fk = lane >> 5;
    const int T = K / kBK;
    constexpr int kChTile = MODE == kModeGeglu ? 128 : 256;     // output channels of a tile

    // this workgroup's tile sequence: pass p -> logical tile p * G + perm(b); the 32 tiles a pass deals to one XCD
    // (workgroup b runs on XCD b % 8) are consecutive.  G is a multiple of 8 (launch).
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    const int perm = (b & 7) * (G >> 3) + (b >> 3);
    auto tile_of = [&](int pass) { return pass * G + perm; };
#if GD_GEMM_ORDER == 0
    auto tile_tn = [&](int tile) { return tile % tiles_n; };
    auto tile_tm = [&](int tile) { return tile / tiles_n; };
#else
    const int tiles_m = ntiles / tiles_n;
    auto tile_tn = [&](int tile) { return tile / tiles_m; };
    auto tile_tm = [&](int tile) { return tile % tiles_m; };
#endif

    const uint32_t row_bytes = (uint32_t)K * 2u;
    const int w_rows = MODE == kModeGeglu ? 2 * N : N;
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)((uint32_t)w_rows * row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((uint32_t)M * row_bytes), 0x00020000);

    // loader: a half tile is 1024 16-byte chunks = 2 per thread; chunk q = tid + 512 i sits at LDS byte 16 q (lane-linear
    // DMA), and holds logical (row r, chunk c & 7) with line = q >> 4, c = (q & 15) ^ (line & 15), r = 2 line + (c >> 3)
    int ld_r[2], ld_c[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int q = tid + kThreads * i;
        const int line = q >> 4, c = (q & 15) ^ (line & 15);
        ld_r[i] = 2 * line + (c >> 3);
        ld_c[i] = (c & 7) * 16;
    }
    // per-thread source offsets of the tile the W / x loader is working on (kOOB = row outside the matrix: zero fill)
    uint32_t w_voff[2][2], x_voff[2][2];
    int w_pass = 0, w_t = 0, x_pass = 0, x_t = 0;
    auto set_w_tile = [&](int pass) {
        const int tile = tile_of(pass);
        const bool live = tile < ntiles;
        const int n0 = tile_tn(tile) * kChTile;
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                int n;
                if (MODE == kModeGeglu) {
                    // half h = 64 hidden rows, then the 64 gate rows of the same channels: a wave (which owns a whole half)
                    // ends up with hidden and gate of a channel in the same lane and register position
                    const int r = ld_r[i], ch = n0 + h * 64 + (r & 63);
                    n = ch < N ? (r < 64 ? ch : N + ch) : -1;
                } else {
                    n = n0 + h * kHalf + ld_r[i];
                    if (n >= N) n = -1;
                }
                w_voff[h][i] = (live && n >= 0) ? (uint32_t)n * row_bytes + (uint32_t)ld_c[i] : kOOB;
            }
    };
    auto set_x_tile = [&](int pass) {
        const int tile = tile_of(pass);
        const bool live = tile < ntiles;
        const int m0 = tile_tm(tile) * 256;
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int m = m0 + h * kHalf + ld_r[i];
                x_voff[h][i] = (live && m < M) ? (uint32_t)m * row_bytes + (uint32_t)ld_c[i] : kOOB;
            }
    };
    // exactly TWO DMA instructions per call and thread, whatever the state (the vmcnt immediates count instructions)
    auto issue_w = [&](int h) {
        char* dst = smem + (((w_pass * T + w_t) & 1) * 2 + h) * kSlot + wave * 1024;
        const uint32_t soff = (uint32_t)w_t * (kBK * 2);
#pragma unroll
        for (int i = 0; i < 2; i++) bload_lds16(rs_w, w_voff[h][i], soff, dst + kThreads * 16 * i);
    };
    auto issue_x = [&](int h, int third) {
        char* dst = smem + (4 + third * 2 + h) * kSlot + wave * 1024;
        const uint32_t soff = (uint32_t)x_t * (kBK * 2);
#pragma unroll
        for (int i = 0; i < 2; i++) bload_lds16(rs_x, x_voff[h][i], soff, dst + kThreads * 16 * i);
    };
    auto advance_w = [&]() {
        if (++w_t == T) { w_t = 0; set_w_tile(++w_pass); }
    };
    auto advance_x = [&]() {
        if (++x_t == T) { x_t = 0; set_x_tile(++x_pass); }
    };

    // fragment read offsets inside a half-tile slot for kk = 0 (kk > 0: XOR with kk << 5)
    uint32_t w_rd[4], x_rd[2];
#pragma unroll
    for (int a = 0; a < 4; a++) w_rd[a] = (uint32_t)swz(a * 32 + frow, fk);
#pragma unroll
    for (int bb = 0; bb < 2; bb++) x_rd[bb] = (uint32_t)swz((wp & 1) * 64 + bb * 32 + frow, fk);

    // The accumulators of a tile START as bias (+ residual): y = bf16(residual + bias + x W^T), one rounding, like addmm.  The
    // bias comes through SCALAR loads (wave-uniform address: 8 consecutive channels = one s_load_dwordx4, the lane keeps the
    // half its fk selects) -- an ordinary vector load beside LDS-DMA in flight is waited for with vmcnt(0) by the compiler,
    // i.e. it would drain the operand ring once per tile; only the residual form pays that.
    f32x16 acc[4][2];
    auto init_acc = [&](int pass) {
        typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
        const int tile = tile_of(pass);
        const int n0 = tile_tn(tile) * kChTile + wc * (kChTile / 2);
        const bool live = tile < ntiles;
        // residual first, ALL 32 loads before anything else of the new tile is live (the accumulators are dead here, so
        // their registers hold the 64 dwords in flight): one memory round trip per tile, not one per channel quad
        u32x2 rr[2][4][4];
        const bool with_res = MODE == kModeBias && residual != nullptr;
        if (with_res) {
            const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(
                (void*)residual, 0, live ? (int)(uint32_t)((size_t)M * (size_t)ldy * 2u) : 0, 0x00020000);
            const int m0 = tile_tm(tile) * 256 + (wp >> 1) * kHalf + (wp & 1) * 64;
#pragma unroll
            for (int bb = 0; bb < 2; bb++) {
                const int m = m0 + bb * 32 + frow;
                const uint32_t row_off = m < M ? (uint32_t)m * (uint32_t)ldy * 2u : kOOB;
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int ch = n0 + a * 32 + 8 * q + 4 * fk;
                        rr[bb][a][q] = __builtin_amdgcn_raw_buffer_load_b64(rs_r, (int)addr2(row_off, ch < N ? (uint32_t)ch * 2u : kOOB), 0, 0);
                    }
            }
        }
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                // first of the 8 consecutive channels of (a, q) -- wave-uniform
                const int c8 = MODE == kModeGeglu ? n0 + (a & 1) * 32 + 8 * q : n0 + a * 32 + 8 * q;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (bias != nullptr && live && c8 < N) v = *(const uint4*)(bias + (MODE == kModeGeglu && a >= 2 ? N : 0) + c8);
                const uint32_t lo = fk ? v.z : v.x, hi = fk ? v.w : v.y;
#pragma unroll
                for (int bb = 0; bb < 2; bb++) {
                    float f0 = __uint_as_float(lo << 16), f1 = __uint_as_float(lo & 0xffff0000u);
                    float f2 = __uint_as_float(hi << 16), f3 = __uint_as_float(hi & 0xffff0000u);
                    if (with_res) {
                        const u32x2 r2 = rr[bb][a][q];
                        f0 += __uint_as_float(r2.x << 16);
                        f1 += __uint_as_float(r2.x & 0xffff0000u);
                        f2 += __uint_as_float(r2.y << 16);
                        f3 += __uint_as_float(r2.y & 0xffff0000u);
                    }
                    acc[a][bb][4 * q] = f0;
                    acc[a][bb][4 * q + 1] = f1;
                    acc[a][bb][4 * q + 2] = f2;
                    acc[a][bb][4 * q + 3] = f3;
                }
            }
    };
    init_acc(0);

    // ---- prologue: W(0), x(0), x(1) of the first tile --------------------------------------------------------------
    set_w_tile(0);
    set_x_tile(0);
    issue_w(0);
    issue_w(1);
    advance_w();
    int x_third = 0;                 // ring position (K-tile count mod 3) of the NEXT x K tile to load
    issue_x(0, x_third);
    issue_x(1, x_third);
    advance_x();
    x_third = 1;
    issue_x(0, x_third);
    issue_x(1, x_third);
    advance_x();
    x_third = 2;

#if GD_GEMM_ABLATE == 2
    bf16x8_t abl_w[4], abl_x[2];
#pragma unroll
    for (int a = 0; a < 4; a++) abl_w[a] = *(const bf16x8_t*)(smem + w_rd[a]);
#pragma unroll
    for (int bb = 0; bb < 2; bb++) abl_x[bb] = *(const bf16x8_t*)(smem + x_rd[bb]);
#endif
    int c_par = 0, c_third = 0;      // ring positions of the K tile being multiplied
#if GD_GEMM_PIPE == 0
    int first_after_epilogue = 0;
    for (int pass = 0; tile_of(pass) < ntiles; pass++) {
        const int tile = tile_of(pass);
        for (int t = 0; t < T; t++) {
            // K tile (pass, t) has landed: everything but the two x half tiles issued last (4 DMA instructions) -- and, right
            // after an epilogue, that tile's stores, which are younger than this K tile's operands
#if GD_GEMM_ABLATE != 4
#if GD_GEMM_ABLATE != 5
            if (first_after_epilogue) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + (MODE == kModeGeglu ? 16 : 32)) : "memory");
                first_after_epilogue = 0;
            } else {
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
#endif
            // ... for every wave; and every wave is done reading the K tile before (whose slots the loads below overwrite)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#endif
            const char* sw = smem + (c_par * 2 + wc) * kSlot;
            const char* sx = smem + (4 + c_third * 2 + (wp >> 1)) * kSlot;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
#if GD_GEMM_ABLATE != 1
                if (kk == 0) issue_w(0);
                if (kk == 1) { issue_w(1); advance_w(); }
                if (kk == 2) issue_x(0, x_third);
                if (kk == 3) { issue_x(1, x_third); advance_x(); x_third = x_third == 2 ? 0 : x_third + 1; }
#endif
#if GD_GEMM_ABLATE == 2
                bf16x8_t wf[4], xf[2];
#pragma unroll
                for (int a = 0; a < 4; a++) wf[a] = abl_w[a];
#pragma unroll
                for (int bb = 0; bb < 2; bb++) xf[bb] = abl_x[bb];
#else
                bf16x8_t wf[4], xf[2];
#pragma unroll
                for (int a = 0; a < 4; a++) wf[a] = *(const bf16x8_t*)(sw + (w_rd[a] ^ (uint32_t)(kk << 5)));
#pragma unroll
                for (int bb = 0; bb < 2; bb++) xf[bb] = *(const bf16x8_t*)(sx + (x_rd[bb] ^ (uint32_t)(kk << 5)));
#endif
#if GD_GEMM_ABLATE == 3
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int bb = 0; bb < 2; bb++) acc[a][bb][0] += __builtin_bit_cast(float, (int)wf[a][0] ^ (int)xf[bb][1]);
#else
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int bb = 0; bb < 2; bb++)
                        acc[a][bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[a], xf[bb], acc[a][bb], 0, 0, 0);
#endif
            }
            c_par ^= 1;
            c_third = c_third == 2 ? 0 : c_third + 1;
        }

        // ---- epilogue: D[i = channel][j = row]; lane: row column lane & 31, channels (reg & 3) + 8 (reg >> 2) + 4 fk ----
        // Stores only (bias and residual went into the accumulators before the first MFMA, init_acc): no load, hence no wait,
        // sits between the operands in flight for the next tile and the first K tile that needs them.  Exactly 32 (GEGLU: 16)
        // store instructions per lane, out-of-range lanes masked by the buffer range check -- the vmcnt immediate counts them.
        {
            typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
            const int m0 = tile_tm(tile) * 256 + (wp >> 1) * kHalf + (wp & 1) * 64;
            const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
                (void*)y, 0, (int)(uint32_t)((size_t)M * (size_t)ldy * 2u), 0x00020000);
            uint32_t row_off[2];
#pragma unroll
            for (int bb = 0; bb < 2; bb++) {
                const int m = m0 + bb * 32 + frow;
                row_off[bb] = m < M ? (uint32_t)m * (uint32_t)ldy * 2u : kOOB;
            }
            const int n0 = tile_tn(tile) * kChTile + wc * (kChTile / 2);     // this wave's first output channel
#pragma unroll
            for (int a = 0; a < (MODE == kModeGeglu ? 2 : 4); a++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int ch = n0 + a * 32 + 8 * q + 4 * fk;
                    const uint32_t ch_off = ch < N ? (uint32_t)ch * 2u : kOOB;
#pragma unroll
                    for (int bb = 0; bb < 2; bb++) {
                        const f32x16& ah = acc[a][bb];
                        u32x2 o = {pack_bf16(ah[4 * q], ah[4 * q + 1]), pack_bf16(ah[4 * q + 2], ah[4 * q + 3])};
                        if (MODE == kModeGeglu) {
                            // hidden and gate rounded to bf16 (the unfused projection's rounding point), then the GEGLU
                            const f32x16& ag = acc[a + 2][bb];
                            o.x = gdnn::geglu2(o.x, pack_bf16(ag[4 * q], ag[4 * q + 1]));
                            o.y = gdnn::geglu2(o.y, pack_bf16(ag[4 * q + 2], ag[4 * q + 3]));
                        }
                        __builtin_amdgcn_raw_buffer_store_b64(o, rs_y, (int)addr2(row_off[bb], ch_off), 0, 0);
                    }
                }
        }
        init_acc(pass + 1);
        first_after_epilogue = 1;
    }
#else
    // ---- the K loop as a PING-PONG of the two waves of each SIMD (waves w and w + 4 share SIMD w % 4) -----------------------
    // A K tile is two half steps (kk = 0, 1 | kk = 2, 3), each a MEMORY segment (this wave's 4 DMA instructions of the half
    // tiles due now, the 12 fragment reads of the half step, wait for them) and a COMPUTE segment (16 MFMAs on those
    // registers), with a workgroup barrier after every segment.  Waves 4..7 start ONE barrier late: between any two barriers
    // one wave of a SIMD issues nothing but MFMAs while the other issues nothing but memory operations -- the in-order
    // instruction stream of a wave no longer serialises its own DMA issue (~100 cycles a piece), LDS latency and matrix work
    // (measured on the single-phase loop: removing ANY one of DMA / fragment reads / MFMAs bought 15-25 %, i.e. each wave
    // paid their SUM).  Ring safety with the groups one segment apart: K tile f is read in the four intervals 4f .. 4f + 3
    // (first group 4f, 4f + 2; second group 4f + 1, 4f + 3), so its slots are free from interval 4f + 4 on -- W(f + 1)
    // (slots of W(f - 1)) is issued in 4f / 4f + 1, x(f + 2) (slots of x(f - 1)) in 4f + 2 / 4f + 3 -- and K tile f + 1 has
    // landed for everybody before interval 4f + 4 because every wave's `vmcnt(4)` ("all but the x pieces just issued")
    // stands in front of its barrier at the end of its second memory segment, which is at or before that interval's start.
    const int late = wave >> 2;                         // second group
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // K tile 0 (prologue) has landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (late) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    auto seg_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int pass = 0; tile_of(pass) < ntiles; pass++) {
        const int tile = tile_of(pass);
        for (int t = 0; t < T; t++) {
            const char* sw = smem + (c_par * 2 + wc) * kSlot;
            const char* sx = smem + (4 + c_third * 2 + (wp >> 1)) * kSlot;
#pragma unroll
            for (int hs = 0; hs < 2; hs++) {
                // ---- memory segment ----
#if GD_GEMM_ABLATE != 1
                if (hs == 0) {
                    issue_w(0);
                    issue_w(1);
                    advance_w();
                } else {
                    issue_x(0, x_third);
                    issue_x(1, x_third);
                    advance_x();
                    x_third = x_third == 2 ? 0 : x_third + 1;
                }
#endif
                bf16x8_t wf[2][4], xf[2][2];
#pragma unroll
                for (int k2 = 0; k2 < 2; k2++) {
                    const uint32_t kx = (uint32_t)((2 * hs + k2) << 5);
#if GD_GEMM_ABLATE == 2
#pragma unroll
                    for (int a = 0; a < 4; a++) wf[k2][a] = abl_w[a];
#pragma unroll
                    for (int bb = 0; bb < 2; bb++) xf[k2][bb] = abl_x[bb];
#else
#pragma unroll
                    for (int a = 0; a < 4; a++) wf[k2][a] = *(const bf16x8_t*)(sw + (w_rd[a] ^ kx));
#pragma unroll
                    for (int bb = 0; bb < 2; bb++) xf[k2][bb] = *(const bf16x8_t*)(sx + (x_rd[bb] ^ kx));
#endif
                }
#if GD_GEMM_ABLATE != 4
                if (hs == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                seg_barrier();
#endif
                // ---- compute segment ----
#pragma unroll
                for (int k2 = 0; k2 < 2; k2++)
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int bb = 0; bb < 2; bb++)
#if GD_GEMM_ABLATE == 3
                            acc[a][bb][0] += __builtin_bit_cast(float, (int)wf[k2][a][0] ^ (int)xf[k2][bb][1]);
#else
                            acc[a][bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[k2][a], xf[k2][bb], acc[a][bb], 0, 0, 0);
#endif
                if (hs == 0) {
#if GD_GEMM_ABLATE != 4
                    seg_barrier();
#endif
                }
            }
            c_par ^= 1;
            c_third = c_third == 2 ? 0 : c_third + 1;
            if (t + 1 < T) {
#if GD_GEMM_ABLATE != 4
                seg_barrier();
#endif
            }
        }
        // ---- epilogue: D[i = channel][j = row]; lane: row column lane & 31, channels (reg & 3) + 8 (reg >> 2) + 4 fk ----
        // Stores only (bias and residual went into the accumulators before the first MFMA, init_acc): no load, hence no wait,
        // sits between the operands in flight for the next tile and the first K tile that needs them.  Exactly 32 (GEGLU: 16)
        // store instructions per lane, out-of-range lanes masked by the buffer range check -- the vmcnt immediate counts them.
        {
            typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
            const int m0 = tile_tm(tile) * 256 + (wp >> 1) * kHalf + (wp & 1) * 64;
            const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
                (void*)y, 0, (int)(uint32_t)((size_t)M * (size_t)ldy * 2u), 0x00020000);
            uint32_t row_off[2];
#pragma unroll
            for (int bb = 0; bb < 2; bb++) {
                const int m = m0 + bb * 32 + frow;
                row_off[bb] = m < M ? (uint32_t)m * (uint32_t)ldy * 2u : kOOB;
            }
            const int n0 = tile_tn(tile) * kChTile + wc * (kChTile / 2);     // this wave's first output channel
#pragma unroll
            for (int a = 0; a < (MODE == kModeGeglu ? 2 : 4); a++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int ch = n0 + a * 32 + 8 * q + 4 * fk;
                    const uint32_t ch_off = ch < N ? (uint32_t)ch * 2u : kOOB;
#pragma unroll
                    for (int bb = 0; bb < 2; bb++) {
                        const f32x16& ah = acc[a][bb];
                        u32x2 o = {pack_bf16(ah[4 * q], ah[4 * q + 1]), pack_bf16(ah[4 * q + 2], ah[4 * q + 3])};
                        if (MODE == kModeGeglu) {
                            // hidden and gate rounded to bf16 (the unfused projection's rounding point), then the GEGLU
                            const f32x16& ag = acc[a + 2][bb];
                            o.x = gdnn::geglu2(o.x, pack_bf16(ag[4 * q], ag[4 * q + 1]));
                            o.y = gdnn::geglu2(o.y, pack_bf16(ag[4 * q + 2], ag[4 * q + 3]));
                        }
                        __builtin_amdgcn_raw_buffer_store_b64(o, rs_y, (int)addr2(row_off[bb], ch_off), 0, 0);
                    }
                }
        }
        init_acc(pass + 1);
#if GD_GEMM_ABLATE != 4
        seg_barrier();          // the barrier behind the tile's last compute segment (epilogue and next tile's bias in front of it)
#endif
    }
    if (!late) {                // the first group is one barrier ahead: pair the second group's last one
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
#endif
    // the ring's tail: DMA instructions issued for tiles past the end were out of range (no LDS write pending that matters),
    // but the wave must not end with DMA outstanding into LDS another workgroup may be given
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

thread_local char g_err[256] = "";
int fail(int code, const char* msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int num_cus()
{
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

template <int MODE>
int launch(void* stream, const void* x, const void* w, const void* bias, const void* residual, void* y, int64_t M, int K,
           int N, int ldy)
{
    static bool attr_set = false;
    auto kfn = gemm256_kernel<MODE>;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess)
            return fail(GD_NN_ERR_HIP, "gd_nn_gemm: cannot reserve 160 KiB of LDS");
        attr_set = true;
    }
    const int ch_tile = MODE == kModeGeglu ? 128 : 256;
    const int tiles_n = (N + ch_tile - 1) / ch_tile;
    const int64_t tiles_m = (M + 255) / 256;
    const int64_t ntiles = tiles_m * tiles_n;
    int grid = num_cus() & ~7;
    if (grid < 8) grid = 8;
    if (ntiles < grid) grid = (int)((ntiles + 7) & ~7);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(kThreads), kLds, (hipStream_t)stream, (const uint16_t*)x, (const uint16_t*)w,
                       (const uint16_t*)bias, (const uint16_t*)residual, (uint16_t*)y, (int)M, K, N, ldy, tiles_n, (int)ntiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

}  // namespace

extern "C" {

int gd_nn_gemm_supported(int64_t M, int K, int N, int geglu)
{
    const int64_t w_rows = geglu ? 2 * (int64_t)N : N;
    if (M <= 0 || K <= 0 || N <= 0 || K % kBK || N % 8) return 0;
    if (M * (int64_t)K * 2 >= (1ll << 31) || w_rows * K * 2 >= (1ll << 31) || M * (int64_t)N * 2 >= (1ll << 31)) return 0;
    return 1;
}

int gd_nn_gemm_forward(void* stream, const void* x, const void* weight, const void* bias, const void* residual, void* y,
                       int64_t M, int K, int N)
{
    if (!x || !weight || !y) return fail(GD_NN_ERR_INVALID_ARG, "gd_nn_gemm_forward: null pointer");
    if (!gd_nn_gemm_supported(M, K, N, 0)) return fail(GD_NN_ERR_INVALID_ARG, "gd_nn_gemm_forward: K % 64, N % 8 or a tensor >= 2 GiB");
    return launch<kModeBias>(stream, x, weight, bias, residual, y, M, K, N, N);
}

int gd_nn_gemm_geglu_forward(void* stream, const void* x, const void* weight, const void* bias, void* y, int64_t M, int K,
                             int inner)
{
    if (!x || !weight || !y) return fail(GD_NN_ERR_INVALID_ARG, "gd_nn_gemm_geglu_forward: null pointer");
    if (!gd_nn_gemm_supported(M, K, inner, 1)) return fail(GD_NN_ERR_INVALID_ARG, "gd_nn_gemm_geglu_forward: K % 64, inner % 8 or a tensor >= 2 GiB");
    return launch<kModeGeglu>(stream, x, weight, bias, nullptr, y, M, K, inner, inner);
}

const char* gd_nn_gemm_last_error(void) { return g_err; }

}  // extern "C"
