// nn_fp8.hip -- fp8 (OCP e4m3) implicit-GEMM on gfx950's block-scaled matrix instruction, for the NO-GRAD part of the
// SDS step: the UNet forward on 2V samples (threestudio stable_diffusion_guidance.py:235-245; three UNet forwards per
// iteration in the NeTF VSD step, netf/guidance/sd_vsd_utils.py:176-207).
//
//   out[m][co] = dq * sum_{tap, ci} x8[pixel m shifted by tap][ci] * w8[co][tap][ci]  (+ bias[n][co]) (+ residual)
//
// x8 / w8 are e4m3 bytes with ONE fp32 scale per tensor (x = sx * x8, w = sw * w8, dq = sx * sw), accumulation is fp32,
// the output bf16.  The same kernel serves the transformer blocks' linear layers (one tap, an [M][K] x [N][K]^T GEMM)
// and the 3x3 / stride 1 / pad 1 convolutions (nine taps, halo by the buffer descriptor's range check).
//
// v_mfma_scale_f32_32x32x64_f8f6f4 is the only 2x-rate low-precision path of this chip (the plain fp8 MFMAs run at the
// bf16 rate, MI355X_MICROARCH.md): K = 64 per instruction, lane l holds row l & 31 and the 32 consecutive K bytes
// 32 (l >> 5) ... (probed on the hardware, tools/probes/mfma_scale_probe.hip); its per-32-element block scales are set
// to 1.0 (E8M0 0x7f) -- the per-tensor scales are applied once in the epilogue.
//
// Structure = the bf16 implicit-GEMM kernel of nn_conv3x3.hip with the element width halved: a K-step is still one
// 128-BYTE row per pixel / per output channel (now 128 channels of one tap), staged by `buffer_load ... lds` with the
// same source-side XOR swizzle and two LDS stages; per K-step a wave issues two K = 64 instructions per 32x32 tile
// instead of four K = 16 ones, i.e. the same matrix-pipe time per byte staged and twice the FLOPs.  Channel counts
// that are not multiples of 128 (320, 960 ...): the WEIGHT rows are zero-padded to CinP = ceil(Cin / 128) * 128, the
// activation rows are not -- the tail K-step then multiplies bytes of the neighbouring pixel (finite e4m3 values, or
// zeros beyond the tensor) by zero weights.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gd_nn.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) int v4i;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int BKB = 128;                 // bytes (= e4m3 channels) of one tap per K-step
constexpr uint32_t kOOB = 0x80000000u;   // voffset that fails the buffer range check (tensors are < 2 GiB)
constexpr int kUnitScale = 0x7f7f7f7f;   // E8M0 1.0 in every byte

thread_local char g_err[256] = "";
int fail(int code, const char* msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi)
{
    f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// byte offset of logical (row, 16-B chunk j) inside a swizzled [rows][128 B] tile image (as nn_conv3x3.hip)
__device__ __forceinline__ int swz(int row, int j)
{
    return (row >> 1) * 256 + (((((row & 1) << 3) | j) ^ ((row >> 1) & 15)) << 4);
}

__device__ __forceinline__ void bload_lds16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff, char* lds_wave_base)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff,
                                             0, 0);
}

struct TapGeom {          // subset of nn_conv3x3.hip's ConvGeom: stride 1, output grid = input grid
    int H, W, ntaps, back, wtaps;
    uint64_t ty4, tx4, w4;
};

template <int BN, int BM, int WN, int WM>
__global__ __launch_bounds__(64 * WN * WM) void gemm_taps_fp8_kernel(
    const uint8_t* __restrict__ in, const uint8_t* __restrict__ wt, const uint16_t* __restrict__ bias, int bias_img_stride,
    const uint16_t* __restrict__ residual, uint16_t* __restrict__ out, int Nimg, const TapGeom g, int Cin, int CinP,
    int Cout, int tiles_n, int nwg, float dq)
{
    constexpr int THREADS = 64 * WN * WM;
    constexpr int NA = BM * 8 / THREADS;      // 16-B chunks of the pixel tile per thread per K-step
    constexpr int NB = BN * 8 / THREADS;      // ... of the weight tile
    constexpr int FA = BN / WN / 32;          // MFMA tiles per wave along channels
    constexpr int FB = BM / WM / 32;          // ... along pixels
    constexpr int kStage = (BM + BN) * BKB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);   // XCD-contiguous tile order
    const int tn = bid % tiles_n, tm = bid / tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int H = g.H, W = g.W;
    const int HW = H * W;
    const int64_t M = (int64_t)Nimg * HW;

    const uint32_t row_a = (uint32_t)Cin, row_w = (uint32_t)CinP;
    const uint32_t back = (uint32_t)g.back * row_a;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)in - back), 0, (int)((uint32_t)Nimg * (uint32_t)HW * row_a + back), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)wt, 0, (int)((uint32_t)Cout * (uint32_t)g.wtaps * row_w), 0x00020000);

    uint32_t a_off[NA];
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
            a_y[i] = rem / W;
            a_x[i] = rem - a_y[i] * W;
            a_off[i] = (uint32_t)((nimg * H + a_y[i]) * W + a_x[i]) * row_a + (uint32_t)(c & 7) * 16u;
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
        b_off[i] = co < Cout ? (uint32_t)co * (uint32_t)g.wtaps * row_w + (uint32_t)(c & 7) * 16u : kOOB;
    }
    const int kc = CinP / BKB;       // K-steps per tap
    const int nsteps = g.ntaps * kc;

    int ld_tap = 0, ld_c = 0;
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
    set_tap(0);
    auto issue = [&](int buf) {
        char* sA = smem + buf * kStage;                      // pixel tile  [BM][128 B]
        char* sB = sA + BM * BKB;                            // weight tile [BN][128 B]
        const int dy = (int)((g.ty4 >> (4 * ld_tap)) & 15u) - 8, dx = (int)((g.tx4 >> (4 * ld_tap)) & 15u) - 8;
        const uint32_t soff_a = (uint32_t)(dy * W + dx + g.back) * row_a + (uint32_t)ld_c * BKB;
        const uint32_t soff_b = (uint32_t)((g.w4 >> (4 * ld_tap)) & 15u) * row_w + (uint32_t)ld_c * BKB;
#pragma unroll
        for (int i = 0; i < NA; i++) bload_lds16(rs_in, a_voff[i], soff_a, sA + (wave * 64 + THREADS * i) * 16);
#pragma unroll
        for (int i = 0; i < NB; i++) bload_lds16(rs_w, b_off[i], soff_b, sB + (wave * 64 + THREADS * i) * 16);
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
    // fragment = bytes [64 kk + 32 fk, +32) of a row = chunks j0 = 4 kk + 2 fk and j0 + 1; chunk j of a row lives at
    // swz(row, 0) ^ (j << 4)
    uint32_t w_rd[FA], p_rd[FB];
#pragma unroll
    for (int a = 0; a < FA; a++) w_rd[a] = (uint32_t)(BM * BKB + swz(wc * (BN / WN) + a * 32 + frow, 0)) ^ (uint32_t)(fk << 5);
#pragma unroll
    for (int b = 0; b < FB; b++) p_rd[b] = (uint32_t)swz(wp * (BM / WM) + b * 32 + frow, 0) ^ (uint32_t)(fk << 5);

    issue(0);
    for (int s = 0; s < nsteps; s++) {
        const int buf = s & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                       // stage `buf` landed for everyone; stage buf^1 free
        if (s + 1 < nsteps) issue(buf ^ 1);
        const char* st = smem + buf * kStage;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            v8i wf[FA], pf[FB];
#pragma unroll
            for (int a = 0; a < FA; a++) {
                const v4i lo = *(const v4i*)(st + (w_rd[a] ^ (uint32_t)(kk << 6)));
                const v4i hi = *(const v4i*)(st + (w_rd[a] ^ (uint32_t)((kk << 6) | 16)));
                wf[a] = (v8i){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int b = 0; b < FB; b++) {
                const v4i lo = *(const v4i*)(st + (p_rd[b] ^ (uint32_t)(kk << 6)));
                const v4i hi = *(const v4i*)(st + (p_rd[b] ^ (uint32_t)((kk << 6) | 16)));
                pf[b] = (v8i){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int a = 0; a < FA; a++)
#pragma unroll
                for (int b = 0; b < FB; b++)
                    acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[a], pf[b], acc[a][b], 0, 0, 0, kUnitScale,
                                                                                0, kUnitScale);
        }
    }

    // ---- epilogue: D[i = channel][j = pixel]; lane: pixel column lane&31, rows (reg&3)+8*(reg>>2)+4*(lane>>5)
#pragma unroll
    for (int b = 0; b < FB; b++) {
        const int64_t m = (int64_t)m0 + wp * (BM / WM) + b * 32 + (lane & 31);
        if (m >= M) continue;
        const int nimg = (int)(m / HW);
        const uint16_t* bias_n = bias ? bias + (size_t)nimg * bias_img_stride : nullptr;
#pragma unroll
        for (int a = 0; a < FA; a++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int co = n0 + wc * (BN / WN) + a * 32 + 8 * q + 4 * fk;
                if (co >= Cout) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = acc[a][b][4 * q + e] * dq;
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
                o.x = pack_bf16(v[0], v[1]);
                o.y = pack_bf16(v[2], v[3]);
                *(uint2*)(out + (size_t)m * Cout + co) = o;
            }
        }
    }
}

// ---- quantisation --------------------------------------------------------------------------------------------------
// saturate to the e4m3 range; a NaN stays a NaN (fminf / fmaxf would turn it into -448 and hide a diverged activation)
__device__ __forceinline__ float clamp448(float v) { return v < -448.f ? -448.f : (v > 448.f ? 448.f : v); }

// y8[i] = e4m3(sat(x[i] * inv_scale)), 8 bf16 -> 8 bytes per thread step
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const uint4* __restrict__ x, uint2* __restrict__ y, int64_t nvec,
                                                           float inv_scale)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const uint4 v = x[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        float f[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            f[2 * k] = clamp448(bf2f((uint16_t)(w[k] & 0xffff)) * inv_scale);
            f[2 * k + 1] = clamp448(bf2f((uint16_t)(w[k] >> 16)) * inv_scale);
        }
        int o0 = 0, o1 = 0;
        o0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], o0, false);
        o0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], o0, true);
        o1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], o1, false);
        o1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], o1, true);
        y[i] = make_uint2((uint32_t)o0, (uint32_t)o1);
    }
}

// w8[r][kp] = e4m3(sat(w[r][k] * inv_scale)) for k < K, 0 for K <= kp < Kp
__global__ __launch_bounds__(256) void pack_weights_fp8_kernel(const uint16_t* __restrict__ w, uint8_t* __restrict__ out,
                                                               int64_t rows, int K, int Kp, float inv_scale)
{
    const int64_t total = rows * (Kp / 4);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / (Kp / 4);
        const int k0 = (int)(i - r * (Kp / 4)) * 4;
        float f[4];
#pragma unroll
        for (int e = 0; e < 4; e++) f[e] = k0 + e < K ? clamp448(bf2f(w[r * K + k0 + e]) * inv_scale) : 0.f;
        int o = 0;
        o = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], o, false);
        o = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], o, true);
        *(int*)(out + r * Kp + k0) = o;
    }
}

int launch(hipStream_t s, const void* x, const void* w, const void* bias, int bias_img_stride, const void* residual, void* y,
           int Nimg, TapGeom g, int Cin, int CinP, int Cout, float dq)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return fail(GD_NN_ERR_HIP, "hipGetDevice failed");
    int minlin = 0;
    for (int t = 0; t < g.ntaps; t++) {
        const int dy = (int)((g.ty4 >> (4 * t)) & 15u) - 8, dx = (int)((g.tx4 >> (4 * t)) & 15u) - 8;
        minlin = dy * g.W + dx < minlin ? dy * g.W + dx : minlin;
    }
    g.back = -minlin;
    const int64_t M = (int64_t)Nimg * g.H * g.W;
    if (M <= 0) return GD_NN_OK;
    if (((double)M + g.back) * Cin >= 2147483648.0 || (double)Cout * g.wtaps * CinP >= 2147483648.0)
        return fail(GD_NN_ERR_INVALID_ARG, "fp8 gemm: activation / weight tensor must be < 2 GiB (32-bit buffer offsets)");
    const int64_t t256 = ((M + 255) / 256) * ((Cout + 255) / 256);
    const int64_t t128x256 = ((M + 255) / 256) * ((Cout + 127) / 128);
    int variant = 0;
    if (Cout % 256 == 0 && t256 >= 192) variant = 2;
    else if (t128x256 >= 512 && (CinP * g.ntaps >= 1024 || M >= (1 << 20))) variant = 1;
#define GD_LAUNCH8(BN_, BM_, WN_, WM_)                                                                              \
    do {                                                                                                           \
        auto kern = gemm_taps_fp8_kernel<BN_, BM_, WN_, WM_>;                                                      \
        constexpr int lds = 2 * (BN_ + BM_) * BKB;                                                                 \
        static bool attr_set[16] = {false};                                                                        \
        if (!attr_set[dev]) {                                                                                      \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);         \
            attr_set[dev] = true;                                                                                  \
        }                                                                                                          \
        const int tiles_m = (int)((M + BM_ - 1) / BM_), tiles_n = (Cout + BN_ - 1) / BN_;                          \
        const int nwg = tiles_m * tiles_n;                                                                         \
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(64 * WN_ * WM_), lds, s, (const uint8_t*)x, (const uint8_t*)w,    \
                           (const uint16_t*)bias, bias_img_stride, (const uint16_t*)residual, (uint16_t*)y, Nimg,  \
                           g, Cin, CinP, Cout, tiles_n, nwg, dq);                                                  \
    } while (0)
    if (variant == 2) GD_LAUNCH8(256, 256, 2, 4);
    else if (variant == 1) GD_LAUNCH8(128, 256, 2, 4);
    else GD_LAUNCH8(128, 128, 2, 2);
#undef GD_LAUNCH8
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

void add_tap(TapGeom& g, int dy, int dx, int widx)
{
    g.ty4 |= (uint64_t)(dy + 8) << (4 * g.ntaps);
    g.tx4 |= (uint64_t)(dx + 8) << (4 * g.ntaps);
    g.w4 |= (uint64_t)widx << (4 * g.ntaps);
    g.ntaps++;
}

}  // namespace

extern "C" {

const char* gd_nn_fp8_last_error(void) { return g_err; }

int gd_nn_fp8_quantize(void* stream, const void* x_bf16, void* y_fp8, int64_t n, float inv_scale)
{
    if (!x_bf16 || !y_fp8 || n <= 0 || n % 8) return fail(GD_NN_ERR_INVALID_ARG, "fp8 quantize: need n % 8 == 0");
    const int64_t nvec = n / 8;
    const int64_t blocks = (nvec + 255) / 256;
    hipLaunchKernelGGL(quantize_fp8_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)x_bf16, (uint2*)y_fp8, nvec, inv_scale);
    return hipGetLastError() == hipSuccess ? 0 : fail(GD_NN_ERR_HIP, "fp8 quantize: launch failed");
}

int gd_nn_fp8_pack_weights(void* stream, const void* w_bf16, void* w_fp8, int64_t rows, int K, int Kp, float inv_scale)
{
    if (!w_bf16 || !w_fp8 || rows <= 0 || K <= 0 || Kp < K || Kp % BKB)
        return fail(GD_NN_ERR_INVALID_ARG, "fp8 pack_weights: need Kp >= K, Kp % 128 == 0");
    const int64_t total = rows * (Kp / 4);
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(pack_weights_fp8_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0,
                       (hipStream_t)stream, (const uint16_t*)w_bf16, (uint8_t*)w_fp8, rows, K, Kp, inv_scale);
    return hipGetLastError() == hipSuccess ? 0 : fail(GD_NN_ERR_HIP, "fp8 pack_weights: launch failed");
}

int gd_nn_fp8_linear_forward(void* stream, const void* x_fp8, const void* w_fp8, const void* bias, const void* residual,
                             void* y, int64_t M, int K, int Kp, int Nout, float dq)
{
    if (!x_fp8 || !w_fp8 || !y) return fail(GD_NN_ERR_INVALID_ARG, "fp8 linear: null pointer");
    if (M <= 0 || M > 0x7fffffff || K <= 0 || K % 16 || Kp < K || Kp % BKB || Nout <= 0 || Nout % 4)
        return fail(GD_NN_ERR_INVALID_ARG, "fp8 linear: need K % 16 == 0, Kp % 128 == 0, Nout % 4 == 0");
    TapGeom g = {};
    g.H = 1; g.W = (int)M; g.wtaps = 1;
    add_tap(g, 0, 0, 0);
    return launch((hipStream_t)stream, x_fp8, w_fp8, bias, 0, residual, y, 1, g, K, Kp, Nout, dq);
}

int gd_nn_fp8_conv3x3_forward(void* stream, const void* x_fp8, const void* w_fp8, const void* bias, int bias_img_stride,
                              const void* residual, void* y, int N, int H, int W, int Cin, int CinP, int Cout, float dq)
{
    if (!x_fp8 || !w_fp8 || !y) return fail(GD_NN_ERR_INVALID_ARG, "fp8 conv3x3: null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % 16 || CinP < Cin || CinP % BKB || Cout <= 0 || Cout % 4)
        return fail(GD_NN_ERR_INVALID_ARG, "fp8 conv3x3: need Cin % 16 == 0, CinP % 128 == 0, Cout % 4 == 0");
    TapGeom g = {};
    g.H = H; g.W = W; g.wtaps = 9;
    for (int ky = 0; ky < 3; ky++)
        for (int kx = 0; kx < 3; kx++) add_tap(g, ky - 1, kx - 1, ky * 3 + kx);
    return launch((hipStream_t)stream, x_fp8, w_fp8, bias, bias_img_stride, residual, y, N, g, Cin, CinP, Cout, dq);
}

}  // extern "C"
