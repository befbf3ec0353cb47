// nn.Linear with K = 320 of the UNet's 64x64-token transformer blocks -- to_q of the cross-attention, to_out.0, proj_in,
// proj_out (N = 320), the fused q | k projection of the self-attention (N = 640) and the GEGLU projection (N = 2560,
// optionally with the GEGLU itself as the epilogue) of diffusers' Transformer2DModel / Attention / FeedForward;
// GarmentDreamer reaches them through threestudio's StableDiffusionGuidance.forward_unet
// (stable_diffusion_guidance.py:106-118) -- for MI355X.
//
// y[M][N] = x[M][320] . W[N][320]^T + bias, bf16 in / out, fp32 accumulation.  With M = 16 * 4096 rows the 320 -> 320
// product moves 42 MB in and 42 MB out for 13 GFLOP: it is an HBM stream, and a tiled library GEMM spends its time in
// the five-step K pipelines of 512 short-lived workgroups (hipBLASLt: 38 us = 2.2 TB/s).  Here the WEIGHTS LIVE IN
// REGISTERS: a workgroup is ten waves and owns one block of 320 output channels; wave w keeps channels
// [32 w, 32 w + 32) x all 320 inputs as its twenty MFMA A-operand fragments (80 VGPRs) for the life of the kernel, and
// the (persistent) workgroup streams 32-row tiles of x through FOUR LDS stages by LDS-DMA -- three tiles in flight
// behind the one being multiplied, each wait an `s_waitcnt vmcnt(n)` with n counted per instruction (gfx9 retires
// loads and stores through one in-order counter), never vmcnt(0).  The result goes through an LDS tile so that it
// leaves as whole rows, and those stores are issued at the top of the NEXT pass, under its MFMAs: one bare s_barrier
// per pass.  N = 640 / 2560: 2 / 8 column blocks, the workgroups that share a row tile on one XCD.
//
// LDS: x stage [32 rows][768 B] (640 used): the 16-byte chunk c of row r sits at slot (c & ~15) | ((c & 15) ^ (r & 15)),
// applied on the SOURCE side of the DMA (the LDS side of a DMA piece is lane-linear); the row pitch is a multiple of
// 256 B, so the 16 lanes of a ds_read_b128 service group (16 rows with distinct r & 15) read 16 distinct bank quads.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gd_nn.h"
#include "nn_math.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

constexpr int kK = 320, kN = 320, kBM = 32, kWaves = kN / 32, kThreads = 64 * kWaves;
constexpr int kChunks = kK / 8;                    // 16-byte chunks per row of x (40)
constexpr int kPitch = 768, kSlots = kPitch / 16;  // padded row: 48 chunk slots
constexpr int kStage = kBM * kPitch;               // 24576 B
constexpr int kStages = 4;                         // three tiles in flight behind the one being multiplied
constexpr int kOutPitch = 656;                     // result tile row pitch (16-B aligned, 164 banks: rows spread)
constexpr int kOutOff = kStages * kStage;
constexpr int kOutTile = kBM * kOutPitch;           // 20992 B, two of them (the stores of a tile overlap the next one's MFMAs)
constexpr int kLds = kOutOff + 2 * kOutTile;        // 140288 B
constexpr uint32_t kOOB = 0x80000000u;

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

// GEGLU = true: weight is diffusers' GEGLU projection [2 * 160 nb][320] (hidden rows, then gate rows) and y[M][160 nb] =
// hidden * gelu(gate): a workgroup owns 160 output channels -- waves 0..4 their hidden rows, waves 5..9 their gate rows --
// and the GEGLU arithmetic (nn_math.h, the separate geglu_kernel's) runs in the store phase on the bf16-rounded tile,
// i.e. at the rounding points of the unfused path: the [M][2 * 160 nb] intermediate (335 MB at 65536 rows) is neither
// written nor read.
template <bool GEGLU>
__global__ __launch_bounds__(kThreads) void linear_320_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                              const uint16_t* __restrict__ bias, uint16_t* __restrict__ y,
                                                              int M, int ntiles, int nb)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sOut = smem + kOutOff;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fn = lane & 31, fk = lane >> 5;

    // N = 320 nb outputs: a workgroup owns ONE block of 320 output channels (cb) for its whole life and walks the row
    // tiles; the nb workgroups that share a row tile sit on the same XCD (blockIdx & 7) and run at the same time, so the
    // tile comes from HBM once and from that XCD's L2 for the others.
    const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
    const int cb = rr % nb, tslot = rr / nb;
    const int tiles_per_pass = (int)(gridDim.x / nb);
    constexpr int kOutCh = GEGLU ? kN / 2 : kN;            // output channels of a workgroup
    const int ldy = kOutCh * nb;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
        (void*)y, 0, (int)(uint32_t)((size_t)M * (size_t)ldy * 2u), 0x00020000);      // < 4 GiB (gd_nn_linear_320_supported)
    // first of this wave's 32 weight rows
    const int wrow = GEGLU ? (wave < kWaves / 2 ? kOutCh * cb + 32 * wave : kOutCh * nb + kOutCh * cb + 32 * (wave - kWaves / 2))
                           : kN * cb + 32 * wave;
    // this wave's 32 output channels x 320 inputs as MFMA A fragments: row wrow + fn, k = 16 s + 8 fk .. + 7
    bf16x8_t wf[kK / 16];
#pragma unroll
    for (int s = 0; s < kK / 16; s++) wf[s] = *(const bf16x8_t*)(w + (size_t)(wrow + fn) * kK + 16 * s + 8 * fk);
    // the lane's 16 bias values as 8 packed bf16 pairs (an LDS copy would make the compiler order its reads behind the
    // LDS-DMA in flight -- s_waitcnt vmcnt(0) in the middle of the pipeline)
    uint2 bq2[4];
#pragma unroll
    for (int q = 0; q < 4; q++) bq2[q] = bias ? *(const uint2*)(bias + wrow + 8 * q + 4 * fk) : make_uint2(0u, 0u);

    const __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((uint32_t)M * (uint32_t)(kK * 2)), 0x00020000);
    // DMA pieces of one stage: 32 rows x 48 slots = 24 pieces of 64 lanes: waves 0..7 issue kDma = 3 each per tile
    // (pieces w, w + 8, w + 16), waves 8 and 9 none -- the s_waitcnt immediates below count instructions per wave.
    constexpr int kPieces = kBM * kSlots / 64, kDmaWaves = 8, kDma = kPieces / kDmaWaves;
    constexpr int kOutChunks = kOutCh / 8;                 // 16-byte chunks per output row
    constexpr int kStores = kBM * kOutChunks / kThreads;   // 16-byte stores per thread and (full) tile: 2, GEGLU 1
    static_assert(kDma * kDmaWaves == kPieces && kStores * kThreads == kBM * kOutChunks, "instruction counts per wave");
    const bool dma_wave = wave < kDmaWaves;
    uint32_t a_off[kDma];     // byte offset inside the tile's rows of x, or kOOB (padding slot)
#pragma unroll
    for (int i = 0; i < kDma; i++) {
        const int p = (wave & (kDmaWaves - 1)) + kDmaWaves * i;
        const int qd = p * 64 + lane, row = qd / kSlots, slot = qd - row * kSlots;
        const int c = (slot & ~15) | ((slot & 15) ^ (row & 15));
        a_off[i] = c < kChunks ? (uint32_t)(row * (kK * 2) + c * 16) : kOOB;
    }
    auto issue = [&](int buf, int tile) {
        if (!dma_wave) return;
        char* dst = smem + buf * kStage + wave * 1024;
        // the tile's base goes into the scalar offset, which the buffer range check does not see: rows past M (last
        // tile) and tiles past the end (the pipeline's tail still issues its instructions) are masked here
        const bool live = tile < ntiles;
        const uint32_t soff = live ? (uint32_t)tile * (uint32_t)(kBM * kK * 2) : 0u;
        const uint32_t bytes_valid = live ? (uint32_t)min(M - tile * kBM, kBM) * (uint32_t)(kK * 2) : 0u;
#pragma unroll
        for (int i = 0; i < kDma; i++)
            bload_lds16(rs_x, a_off[i] < bytes_valid ? a_off[i] : kOOB, soff, dst + i * (kDmaWaves * 1024));
    };
    const uint32_t rd_row = (uint32_t)(fn * kPitch), rsw = (uint32_t)(fn & 15);
    const uint32_t out_wr = (uint32_t)(uintptr_t)(sOut + fn * kOutPitch + 64 * wave + 8 * fk);   // LDS byte address
    // the thread's kStores chunks of a result tile: (row, 16-byte chunk) -> LDS address and offset in y
    uint32_t st_lds[kStores], st_row[kStores];
    size_t st_y[kStores];
#pragma unroll
    for (int i = 0; i < kStores; i++) {
        const int j = tid + kThreads * i, row = j / kOutChunks, c = j - row * kOutChunks;
        st_lds[i] = (uint32_t)(uintptr_t)(sOut + row * kOutPitch + c * 16);
        st_row[i] = (uint32_t)row;
        st_y[i] = (size_t)row * ldy + kOutCh * cb + c * 8;
    }
    // result tile `t` (already complete in LDS half `half`, all waves past a barrier) -> whole 640-byte rows of y.
    // (LDS accesses of the result tile are inline asm: the compiler orders every LDS access it cannot tell apart from
    // the input stages behind ALL LDS-DMA in flight -- s_waitcnt vmcnt(0) in the middle of the pipeline.)  Every wave
    // issues exactly kStores store instructions on a full tile (the vmcnt immediates rely on it); the one partial tile
    // is the last pass of its workgroup.
    auto store_tile = [&](int t, int half) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const int m0 = t * kBM;
        u32x4 v[kStores], g[kStores];
#pragma unroll
        for (int i = 0; i < kStores; i++) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(v[i]) : "v"(st_lds[i] + (uint32_t)(half * kOutTile)) : "memory");
            if (GEGLU)            // the gate chunk: same row, 320 bytes on (waves 5..9 wrote it)
                asm volatile("ds_read_b128 %0, %1" : "=v"(g[i]) : "v"(st_lds[i] + (uint32_t)(half * kOutTile + kN)) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < kStores; i++) {
            if (GEGLU) {
#pragma unroll
                for (int k = 0; k < 4; k++) v[i][k] = gdnn::geglu2(v[i][k], g[i][k]);
            }
            // ONE buffer_store_dwordx4 per chunk by construction -- the vmcnt immediates below count instructions -- with
            // rows beyond M sent to an out-of-range offset, which the hardware drops (no branch around the store)
            const uint32_t so = m0 + (int)st_row[i] < M ? (uint32_t)(((size_t)m0 * ldy + st_y[i]) * 2u) : 0xfffffff0u;
            __builtin_amdgcn_raw_buffer_store_b128(v[i], rs_y, (int)so, 0, 0);
        }
    };

    // One bare s_barrier (+ LDS wait) per pass: __syncthreads() carries a release fence that the compiler lowers to
    // s_waitcnt vmcnt(0), which would wait for the tiles in flight and for the stores of the previous pass.
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    const int grid = tiles_per_pass;
    int tile = tslot * 8 + xcd;
#pragma unroll
    for (int k = 0; k < kStages - 1; k++) issue(k, tile + k * grid);
    int it = 0;
    for (; tile < ntiles; it++, tile += grid) {
        const int buf = it & (kStages - 1);
        // vmcnt retires in issue order on gfx9 (one counter for loads and stores): "at most n younger instructions
        // outstanding" means this pass's DMA has landed.  Issue order per DMA wave: D0 D1 D2 | D3 | D4 S0 | D5 S1 | ...
        // (D = kDma DMA instructions of a tile, S = kStores stores of the tile one pass back), so behind D(it) there are
        // 2 D, 2 D, 2 D + S, 2 D + 2 S, then 2 D + 3 S of them.
        if (dma_wave) {           // (waves 8, 9 issue no DMA: the barrier below is all they need)
            if (it <= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kDma) : "memory");
            else if (it == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kDma + kStores) : "memory");
            else if (it == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kDma + 2 * kStores) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kDma + 3 * kStores) : "memory");
        }
        // stage `buf` has landed for everyone; everyone has written the previous result tile and is done READING the
        // one before it (whose LDS half this pass overwrites)
        lds_barrier();
        issue((it + kStages - 1) & (kStages - 1), tile + (kStages - 1) * grid);
        if (it > 0) store_tile(tile - grid, (it - 1) & 1);       // overlaps the MFMAs below
        const char* pa = smem + buf * kStage + rd_row;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < kK / 16; s++) {
            const uint32_t c = (uint32_t)(2 * s) + (uint32_t)fk;
            const uint32_t slot = (c & ~15u) | ((c & 15u) ^ rsw);
            const bf16x8_t pf = *(const bf16x8_t*)(pa + (slot << 4));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s], pf, acc, 0, 0, 0);
        }
        // result tile [32 rows][320 ch] bf16 in LDS: lane owns 4 consecutive channels of row fn per quad q
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint2 o;
            o.x = pack_bf16(acc[4 * q] + __uint_as_float(bq2[q].x << 16), acc[4 * q + 1] + __uint_as_float(bq2[q].x & 0xffff0000u));
            o.y = pack_bf16(acc[4 * q + 2] + __uint_as_float(bq2[q].y << 16), acc[4 * q + 3] + __uint_as_float(bq2[q].y & 0xffff0000u));
            asm volatile("ds_write_b64 %0, %1" ::"v"(out_wr + (uint32_t)((it & 1) * kOutTile + 16 * q)),
                         "v"((unsigned long long)o.x | ((unsigned long long)o.y << 32))
                         : "memory");
        }
    }
    if (it > 0) {                 // the last result tile
        lds_barrier();
        store_tile(tile - grid, (it - 1) & 1);
    }
}

thread_local char g_err[256] = "";
int fail(int code, const char* msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

template <bool GEGLU>
int launch_k320(void* stream, const void* x, const void* weight, const void* bias, void* y, int64_t M, int nb)
{
    static_assert((kStages & (kStages - 1)) == 0, "stage index by mask");
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(GD_NN_ERR_HIP, "hipGetDevice failed");
    static bool attr_set[16] = {false};
    if (dev >= 0 && dev < 16 && !attr_set[dev]) {
        (void)hipFuncSetAttribute((const void*)linear_320_kernel<GEGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        attr_set[dev] = true;
    }
    const int ntiles = (int)((M + kBM - 1) / kBM);
    // 256 workgroups (one per CU of an MI355X; a multiple of 8 nb so that blockIdx -> (XCD, column block, tile slot) is
    // exact), fewer for short row sets
    int grid = 256;
    while (grid > 8 * nb && (grid / nb) / 2 >= ntiles) grid /= 2;
    hipLaunchKernelGGL(linear_320_kernel<GEGLU>, dim3(grid), dim3(kThreads), kLds, (hipStream_t)stream, (const uint16_t*)x,
                       (const uint16_t*)weight, (const uint16_t*)bias, (uint16_t*)y, (int)M, ntiles, nb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

}  // namespace

extern "C" {

int gd_nn_linear_320_supported(int64_t M, int K, int N)
{
    // K = 320 and N = 320, 640 or 2560 (one, two or eight column blocks: the grid of 256 workgroups splits evenly and the
    // workgroups of a row tile share an XCD), enough rows to fill the chip
    return (K == kK && (N == kN || N == 2 * kN || N == 8 * kN) && M >= 4096 && M * kK * 2 < 2147483648LL &&
            M * N * 2 < 4294967296LL) ? 1 : 0;
}

int gd_nn_linear_k320_geglu_forward(void* stream, const void* x, const void* weight, const void* bias, void* y, int64_t M,
                                    int inner)
{
    if (!x || !weight || !y) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (M <= 0 || M * kK * 2 >= 2147483648LL) return fail(GD_NN_ERR_INVALID_ARG, "linear_k320_geglu: need 0 < M and x < 2 GiB");
    if (inner != 4 * kN) return fail(GD_NN_ERR_INVALID_ARG, "linear_k320_geglu: inner must be 1280 (weight [2560][320])");
    return launch_k320<true>(stream, x, weight, bias, y, M, inner / (kN / 2));
}

int gd_nn_linear_k320_forward(void* stream, const void* x, const void* weight, const void* bias, void* y, int64_t M, int N)
{
    if (!x || !weight || !y) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (M <= 0 || M * kK * 2 >= 2147483648LL) return fail(GD_NN_ERR_INVALID_ARG, "linear_k320: need 0 < M and x < 2 GiB");
    if (N != kN && N != 2 * kN && N != 8 * kN) return fail(GD_NN_ERR_INVALID_ARG, "linear_k320: N must be 320, 640 or 2560");
    return launch_k320<false>(stream, x, weight, bias, y, M, N / kN);
}

int gd_nn_linear_320_forward(void* stream, const void* x, const void* weight, const void* bias, void* y, int64_t M)
{
    return gd_nn_linear_k320_forward(stream, x, weight, bias, y, M, kN);
}

const char* gd_nn_linear_320_last_error(void) { return g_err; }

}  // extern "C"
