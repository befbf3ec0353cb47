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
#include <stdint.h>
#include <stdio.h>

#include <mutex>
#include <vector>

#include "../../include/gd_nn.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int BM = 128;   // pixels per tile
constexpr int BN = 128;   // output channels per tile
constexpr int BK = 64;    // channels of one tap per K-step (128-byte rows)
constexpr int kStageBytes = (BM + BN) * BK * 2;  // 32 KiB

__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f)
{
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// bias_img_stride: 0 -> bias[co]; Cout -> bias[n][co] (time-embedding projection folded in).
__global__ __launch_bounds__(256) void conv3x3_nhwc_bf16_kernel(
    const uint16_t* __restrict__ in, const uint16_t* __restrict__ wt, const uint16_t* __restrict__ bias,
    int bias_img_stride, const uint16_t* __restrict__ residual, uint16_t* __restrict__ out, int Nimg, int H, int W,
    int Cin, int Cout, const uint16_t* __restrict__ zeros, int tiles_n, int nwg)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware tile order: consecutive logical tiles (same pixel tile, different Cout tile, then
    // the next pixel tile) stay on one XCD's L2.
    int bid = blockIdx.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int tn = bid % tiles_n, tm = bid / tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int HW = H * W;
    const int64_t M = (int64_t)Nimg * HW;

    // ---- per-thread load descriptors: 4 pixel rows + 4 weight rows, one 16-B chunk each ----
    const int sp = tid & 7;          // stored chunk position inside the 128-B row
    const int r_lo = tid >> 3;       // row 0..31 (+32*i)
    const uint16_t* a_src[4];        // pixel row base (tap 0,0; channel 0), or nullptr if m >= M
    int a_y[4], a_x[4], a_j[4];
    const uint16_t* b_src[4];
    int b_j[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = r_lo + 32 * i;
        const int64_t m = (int64_t)m0 + r;
        a_j[i] = (sp ^ (r & 7)) * 8;
        if (m < M) {
            const int nimg = (int)(m / HW);
            const int rem = (int)(m - (int64_t)nimg * HW);
            a_y[i] = rem / W;
            a_x[i] = rem - a_y[i] * W;
            a_src[i] = in + (size_t)m * Cin;
        } else {
            a_y[i] = -100000; a_x[i] = 0; a_src[i] = zeros;
        }
        const int co = n0 + r;
        b_j[i] = a_j[i];
        b_src[i] = co < Cout ? wt + (size_t)co * 9 * Cin : nullptr;
    }
    const int kc = Cin / BK;         // K-steps per tap
    const int nsteps = 9 * kc;

    auto issue = [&](int s, int buf) {
        const int tap = s / kc, c0 = (s - tap * kc) * BK;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        char* sA = smem + buf * kStageBytes;                 // pixel tile  [128][64] bf16
        char* sB = sA + BM * BK * 2;                         // weight tile [128][64] bf16
        const int tap_off = (dy * W + dx) * Cin + c0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int yy = a_y[i] + dy, xx = a_x[i] + dx;
            const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            const uint16_t* src = ok ? a_src[i] + tap_off + a_j[i] : zeros;
            glds16(src, sA + (wave * 64 + 256 * i) * 16);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint16_t* src = b_src[i] ? b_src[i] + tap * Cin + c0 + b_j[i] : zeros;
            glds16(src, sB + (wave * 64 + 256 * i) * 16);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc[a][b][k] = 0.f;

    const int wc = wave & 1, wp = wave >> 1;   // wave's 64-channel / 64-pixel quadrant
    const int frow = lane & 31, fk = lane >> 5;

    issue(0, 0);
    for (int s = 0; s < nsteps; s++) {
        const int buf = s & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                       // stage `buf` landed for everyone; stage buf^1 free
        if (s + 1 < nsteps) issue(s + 1, buf ^ 1);
        const char* sA = smem + buf * kStageBytes;
        const char* sB = sA + BM * BK * 2;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            bf16x8_t wf[2], pf[2];
            const int j = kk * 2 + fk;
#pragma unroll
            for (int a = 0; a < 2; a++) {
                const int row = wc * 64 + a * 32 + frow;
                wf[a] = *(const bf16x8_t*)(sB + (row * 8 + (j ^ (row & 7))) * 16);
            }
#pragma unroll
            for (int b = 0; b < 2; b++) {
                const int row = wp * 64 + b * 32 + frow;
                pf[b] = *(const bf16x8_t*)(sA + (row * 8 + (j ^ (row & 7))) * 16);
            }
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[a], pf[b], acc[a][b], 0, 0, 0);
        }
    }

    // ---- epilogue: D[i = channel][j = pixel]; lane: pixel column lane&31, rows (reg&3)+8*(reg>>2)+4*(lane>>5)
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const int64_t m = (int64_t)m0 + wp * 64 + b * 32 + (lane & 31);
        if (m >= M) continue;
        const int nimg = (int)(m / HW);
        const uint16_t* bias_n = bias ? bias + (size_t)nimg * bias_img_stride : nullptr;
#pragma unroll
        for (int a = 0; a < 2; a++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int co = n0 + wc * 64 + a * 32 + 8 * q + 4 * fk;
                if (co >= Cout) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = acc[a][b][4 * q + e];
                if (bias_n) {
                    const uint2 bb = *(const uint2*)(bias_n + co);
                    v[0] += bf2f((uint16_t)(bb.x & 0xffff)); v[1] += bf2f((uint16_t)(bb.x >> 16));
                    v[2] += bf2f((uint16_t)(bb.y & 0xffff)); v[3] += bf2f((uint16_t)(bb.y >> 16));
                }
                if (residual) {
                    const uint2 rr = *(const uint2*)(residual + (size_t)m * Cout + co);
                    v[0] += bf2f((uint16_t)(rr.x & 0xffff)); v[1] += bf2f((uint16_t)(rr.x >> 16));
                    v[2] += bf2f((uint16_t)(rr.y & 0xffff)); v[3] += bf2f((uint16_t)(rr.y >> 16));
                }
                uint2 o;
                o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
                o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
                *(uint2*)(out + (size_t)m * Cout + co) = o;
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

uint16_t* g_zeros[16] = {nullptr};

// optional event timing of the conv kernel (bench.py's roofline line)
struct ConvProf {
    std::mutex mu;
    bool on = false;
    struct Rec { hipEvent_t a, b; };
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;
    double total_ms = 0, total_flops = 0;
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

int gd_nn_conv3x3_forward(void* stream, const void* x, const void* weight, const void* bias, int bias_img_stride,
                          const void* residual, void* y, int N, int H, int W, int Cin, int Cout)
{
    if (!x || !weight || !y) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || Cin % BK || Cout % 4 || Cin <= 0 || Cout <= 0)
        return fail(GD_NN_ERR_INVALID_ARG, "conv3x3: need Cin % 64 == 0 and Cout % 4 == 0");
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return fail(GD_NN_ERR_HIP, "hipGetDevice failed");
    hipStream_t s = (hipStream_t)stream;
    if (!g_zeros[dev]) {
        if (hipMalloc((void**)&g_zeros[dev], 256) != hipSuccess) return fail(GD_NN_ERR_HIP, "hipMalloc failed");
        if (hipMemset(g_zeros[dev], 0, 256) != hipSuccess) return fail(GD_NN_ERR_HIP, "hipMemset failed");
    }
    const int64_t M = (int64_t)N * H * W;
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (Cout + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    static bool attr_set[16] = {false};
    if (!attr_set[dev]) {
        (void)hipFuncSetAttribute((const void*)conv3x3_nhwc_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  2 * kStageBytes);
        attr_set[dev] = true;
    }
    hipEvent_t ea = nullptr, eb = nullptr;
    if (g_cprof.on) {
        std::lock_guard<std::mutex> lk(g_cprof.mu);
        ea = g_cprof.get(); eb = g_cprof.get();
        if (ea && eb) (void)hipEventRecord(ea, s);
    }
    hipLaunchKernelGGL(conv3x3_nhwc_bf16_kernel, dim3(nwg), dim3(256), 2 * kStageBytes, s, (const uint16_t*)x,
                       (const uint16_t*)weight, (const uint16_t*)bias, bias_img_stride, (const uint16_t*)residual,
                       (uint16_t*)y, N, H, W, Cin, Cout, g_zeros[dev], tiles_n, nwg);
    if (ea && eb) {
        (void)hipEventRecord(eb, s);
        std::lock_guard<std::mutex> lk(g_cprof.mu);
        g_cprof.pending.push_back({ea, eb});
        g_cprof.total_flops += 2.0 * (double)M * Cout * 9.0 * Cin;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
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
    g_cprof.total_ms = g_cprof.total_flops = 0;
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
