// nn_elementwise.hip -- row-wise fused passes of the UNet's transformer blocks (bf16, inference):
//
//   gd_nn_geglu_forward          y = h * gelu(gate),  [h | gate] = the two halves of one GEGLU projection row
//                                (diffusers GEGLU: hidden, gate = proj(x).chunk(2, -1); hidden * F.gelu(gate)).
//                                PyTorch runs chunk -> gelu (read+write) -> mul (2 reads + write): 5 row passes,
//                                the gelu over a strided half at ~2 TB/s; here 3 passes at stream rate.
//   gd_nn_add_layernorm_forward  s = x + r (optional r, optional store of s);  y = LayerNorm(s) * w + b.
//                                BasicTransformerBlock: x = x + attn(norm(x)) followed by the next norm
//                                (threestudio/diffusers attention.py).  PyTorch: add (3 passes) + a LayerNorm
//                                kernel that reaches ~1 TB/s on C = 320 rows; here 4 passes, one wave per row.
//
// Both are pure HBM/MALL streams: 16-byte (8 x bf16) accesses, fp32 math, erf-based (exact) GELU like
// F.gelu's default.  LayerNorm statistics: two-pass over registers (mean, then centred sum of squares) in
// fp32, wave-wide DPP/shuffle reduction, no LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gd_nn.h"
#include "nn_math.h"

namespace {

thread_local char g_err[256] = "";
int fail(int code, const char* msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

struct alignas(16) bf16x8 {
    uint16_t v[8];
};

__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
// Packed fp32 helpers: these row passes are VALU-bound on MI355X (a wave64 VALU instruction holds its SIMD ~4.5 cycles,
// tools/probes/valu_rate_probe.hip; SQ_INSTS_VALU x 4.5 cycles = the whole kernel time in profiles/r02_pmc.json), so two
// channels share every arithmetic instruction (v_pk_*_f32) and bf16 rounding is v_cvt_pk_bf16_f32 (nearest even).
using gdnn::f2;
using gdnn::unpack2;
using gdnn::pack2;
using gdnn::round_bf16;
using gdnn::erf_as2;
struct alignas(16) u32x4 { uint32_t w[4]; };

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// x: [rows][2*inner] (hidden | gate), y: [rows][inner]; one thread per 8 output channels
__global__ __launch_bounds__(256) void geglu_kernel(const bf16x8* __restrict__ x, bf16x8* __restrict__ y, int64_t nvec,
                                                    int vin /* inner / 8 */)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / vin;
        const int c = (int)(i - row * vin);
        const u32x4 h = __builtin_bit_cast(u32x4, x[row * 2 * vin + c]);
        const u32x4 g = __builtin_bit_cast(u32x4, x[row * 2 * vin + vin + c]);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; k++) o.w[k] = gdnn::geglu2(h.w[k], g.w[k]);   // (nn_math.h; also the fused GEMM epilogue's)
        y[i] = __builtin_bit_cast(bf16x8, o);
    }
}

// One wave per row; VPL = 16-byte vectors per lane (C <= 64 * 8 * VPL).
template <int VPL>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const bf16x8* __restrict__ x, const bf16x8* __restrict__ r,
                                                            const bf16x8* __restrict__ w, const bf16x8* __restrict__ b,
                                                            bf16x8* __restrict__ s_out, bf16x8* __restrict__ y,
                                                            int64_t rows, int vpr /* C / 8 */, float eps)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16x8* xr = x + row * vpr;
    f2 v[VPL][4];
    f2 sum2 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        const int c = lane + 64 * i;
        if (c < vpr) {
            const u32x4 a = __builtin_bit_cast(u32x4, xr[c]);
            if (r) {
                const u32x4 rr = __builtin_bit_cast(u32x4, r[row * vpr + c]);
                u32x4 so;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    so.w[k] = pack2(unpack2(a.w[k]) + unpack2(rr.w[k]));   // the residual stream stays bf16, as in eager
                    v[i][k] = unpack2(so.w[k]);
                }
                if (s_out) s_out[row * vpr + c] = __builtin_bit_cast(bf16x8, so);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) v[i][k] = unpack2(a.w[k]);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) sum2 += v[i][k];
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) v[i][k] = f2{0.f, 0.f};
        }
    }
    const float inv_c = 1.0f / (float)(vpr * 8);
    const float mean = wave_sum(sum2.x + sum2.y) * inv_c;
    f2 sq2 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        if (lane + 64 * i < vpr) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const f2 d = v[i][k] - mean;
                sq2 += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq2.x + sq2.y) * inv_c + eps);
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        const int c = lane + 64 * i;
        if (c < vpr) {
            const u32x4 ww = __builtin_bit_cast(u32x4, w[c]), bb = __builtin_bit_cast(u32x4, b[c]);
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; k++) o.w[k] = pack2(((v[i][k] - mean) * rstd) * unpack2(ww.w[k]) + unpack2(bb.w[k]));
            y[row * vpr + c] = __builtin_bit_cast(bf16x8, o);
        }
    }
}

// Backward of GEGLU: x = [h | g] (the projection's output), dy the gradient of h * gelu(g):
//   dh = dy * gelu(g),   dg = dy * h * (Phi(g) + g phi(g)),   Phi(g) = (1 + erf(g / sqrt 2)) / 2,  phi(g) = exp(-g^2 / 2) / sqrt(2 pi)
// written as one [rows][2 * inner] tensor (what the projection's backward consumes); fp32 math, one rounding per output.
__global__ __launch_bounds__(256) void geglu_backward_kernel(const bf16x8* __restrict__ x, const bf16x8* __restrict__ dy,
                                                             bf16x8* __restrict__ dx, int64_t nvec, int vin)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / vin;
        const int c = (int)(i - row * vin);
        const u32x4 h = __builtin_bit_cast(u32x4, x[row * 2 * vin + c]);
        const u32x4 g = __builtin_bit_cast(u32x4, x[row * 2 * vin + vin + c]);
        const u32x4 d = __builtin_bit_cast(u32x4, dy[i]);
        u32x4 oh, og;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const f2 gv = unpack2(g.w[k]), hv = unpack2(h.w[k]), dv = unpack2(d.w[k]);
            const f2 cdf = 0.5f * (1.0f + erf_as2(gv * 0.70710678118654752f));
            const f2 a = (gv * gv) * -0.72134752044448170f;            // -g^2 / 2 * log2(e)
            const f2 pdf = 0.39894228040143268f * f2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
            oh.w[k] = pack2(dv * (gv * cdf));
            og.w[k] = pack2((dv * hv) * (cdf + gv * pdf));
        }
        dx[row * 2 * vin + c] = __builtin_bit_cast(bf16x8, oh);
        dx[row * 2 * vin + vin + c] = __builtin_bit_cast(bf16x8, og);
    }
}

// Backward of y = LayerNorm(s) * w + b w.r.t. s, plus the gradient that reaches s directly (the residual stream):
//   a = w * dy,  dx = rstd * (a - mean(a) - xhat * mean(a * xhat)) + ds,   mean / rstd recomputed from the row (registers).
// One wave per row; VPL = 16-byte vectors per lane.
template <int VPL>
__global__ __launch_bounds__(256) void layernorm_backward_kernel(const bf16x8* __restrict__ s, const bf16x8* __restrict__ dy,
                                                                 const bf16x8* __restrict__ w, const bf16x8* __restrict__ ds,
                                                                 bf16x8* __restrict__ dx, int64_t rows, int vpr, float eps)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    f2 v[VPL][4], a[VPL][4];
    f2 sum2 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        const int c = lane + 64 * i;
        if (c < vpr) {
            const u32x4 q = __builtin_bit_cast(u32x4, s[row * vpr + c]);
#pragma unroll
            for (int k = 0; k < 4; k++) { v[i][k] = unpack2(q.w[k]); sum2 += v[i][k]; }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) v[i][k] = f2{0.f, 0.f};
        }
    }
    const float inv_c = 1.0f / (float)(vpr * 8);
    const float mean = wave_sum(sum2.x + sum2.y) * inv_c;
    f2 sq2 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < VPL; i++)
        if (lane + 64 * i < vpr) {
#pragma unroll
            for (int k = 0; k < 4; k++) { const f2 d = v[i][k] - mean; sq2 += d * d; }
        }
    const float rstd = rsqrtf(wave_sum(sq2.x + sq2.y) * inv_c + eps);
    f2 m1 = {0.f, 0.f}, m2 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        const int c = lane + 64 * i;
        if (c < vpr) {
            const u32x4 g = __builtin_bit_cast(u32x4, dy[row * vpr + c]), ww = __builtin_bit_cast(u32x4, w[c]);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v[i][k] = (v[i][k] - mean) * rstd;                      // xhat
                a[i][k] = unpack2(g.w[k]) * unpack2(ww.w[k]);
                m1 += a[i][k];
                m2 += a[i][k] * v[i][k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) a[i][k] = f2{0.f, 0.f};
        }
    }
    const float ma = wave_sum(m1.x + m1.y) * inv_c, mb = wave_sum(m2.x + m2.y) * inv_c;
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        const int c = lane + 64 * i;
        if (c < vpr) {
            u32x4 o;
            u32x4 e = {{0u, 0u, 0u, 0u}};
            if (ds) e = __builtin_bit_cast(u32x4, ds[row * vpr + c]);
#pragma unroll
            for (int k = 0; k < 4; k++) o.w[k] = pack2((a[i][k] - ma - v[i][k] * mb) * rstd + unpack2(e.w[k]));
            dx[row * vpr + c] = __builtin_bit_cast(bf16x8, o);
        }
    }
}

}  // namespace

// Row softmax of the VAE mid block's single-head attention ([8 images x 4096 queries] rows of 4096 bf16 scores, 268 MB: the
// score matrix is materialised on purpose) and its backward.  One wave per row, the row in registers (R 16-byte
// vectors per lane): forward = read once, write once (in place if y == x); backward ds = p (dp - sum(p dp)) = read p and dp once,
// write once -- torch's softmax backward runs a separate `grad * output` pass over the matrix first (aten::mul, 142 us) and then
// its row kernel (137 us).  fp32 arithmetic, exp through v_exp_f32 on (x - max) log2(e).
template <int R>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const u32x4* x, u32x4* y, int64_t rows, int L8)   // no __restrict__: called in place (y == x)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const u32x4* xr = x + row * L8;
    u32x4 v[R];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int c = lane + 64 * i;
        if (c < L8) {
            v[i] = xr[c];
#pragma unroll
            for (int e = 0; e < 4; e++) { const f2 f = unpack2(v[i].w[e]); m = fmaxf(m, fmaxf(f.x, f.y)); }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    const float k = 1.4426950408889634f;
    const float mk = m == -INFINITY ? 0.f : m * k;
    float sum = 0.f;
    f2 ex[R][4];
#pragma unroll
    for (int i = 0; i < R; i++) {
        if (lane + 64 * i < L8) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const f2 f = unpack2(v[i].w[e]);
                ex[i][e] = f2{__builtin_amdgcn_exp2f(f.x * k - mk), __builtin_amdgcn_exp2f(f.y * k - mk)};
                sum += ex[i][e].x + ex[i][e].y;
            }
        }
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    u32x4* yr = y + row * L8;
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int c = lane + 64 * i;
        if (c < L8) {
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; e++) o.w[e] = pack2(ex[i][e] * inv);
            yr[c] = o;
        }
    }
}

template <int R>
__global__ __launch_bounds__(256) void softmax_rows_backward_kernel(const u32x4* __restrict__ p, const u32x4* dp,
                                                                    u32x4* ds, int64_t rows, int L8)   // ds may be dp (in place)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const u32x4* pr = p + row * L8;
    const u32x4* gr = dp + row * L8;
    u32x4 vp[R], vg[R];
    f2 acc = f2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int c = lane + 64 * i;
        if (c < L8) {
            vp[i] = pr[c];
            vg[i] = gr[c];
        }
    }
#pragma unroll
    for (int i = 0; i < R; i++) {
        if (lane + 64 * i < L8) {
#pragma unroll
            for (int e = 0; e < 4; e++) acc += unpack2(vp[i].w[e]) * unpack2(vg[i].w[e]);
        }
    }
    const float s = wave_sum(acc.x + acc.y);
    u32x4* or_ = ds + row * L8;
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int c = lane + 64 * i;
        if (c < L8) {
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; e++) o.w[e] = pack2(unpack2(vp[i].w[e]) * (unpack2(vg[i].w[e]) - s));
            or_[c] = o;
        }
    }
}

// AutoencoderKL.quant_conv: nn.Conv2d(8, 8, 1) on the encoder's [B, 8, 64, 64] moments (diffusers AutoencoderKL.encode, reached
// through StableDiffusionGuidance.encode_images, stable_diffusion_guidance.py:160-167).  NHWC: a pixel is ONE 16-byte vector;
// thread = pixel, the 8 x 8 weights + bias are wave-uniform (scalar loads), 64 FMAs per pixel in fp32.  TRANSPOSED = the input
// gradient, dx[ci] = sum_co w[co][ci] dy[co] (no bias).  (MIOpen ran this layer on its naive fp64-accumulating kernel, the last
// library convolution of the step.)
template <bool TRANSPOSED>
__global__ __launch_bounds__(256) void conv1x1_c8_kernel(const u32x4* __restrict__ x, const uint16_t* __restrict__ w,
                                                         const uint16_t* __restrict__ bias, u32x4* __restrict__ y, int64_t npix)
{
    float wf[8][8], bf[8];
#pragma unroll
    for (int o = 0; o < 8; o++) {
        bf[o] = (!TRANSPOSED && bias) ? bf2f(bias[o]) : 0.f;
#pragma unroll
        for (int i = 0; i < 8; i++) wf[o][i] = bf2f(TRANSPOSED ? w[i * 8 + o] : w[o * 8 + i]);
    }
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
        const u32x4 q = x[p];
        float v[8], o[8];
#pragma unroll
        for (int e = 0; e < 4; e++) { v[2 * e] = __uint_as_float(q.w[e] << 16); v[2 * e + 1] = __uint_as_float(q.w[e] & 0xffff0000u); }
#pragma unroll
        for (int c = 0; c < 8; c++) {
            float a = bf[c];
#pragma unroll
            for (int i = 0; i < 8; i++) a = fmaf(wf[c][i], v[i], a);
            o[c] = a;
        }
        u32x4 out;
#pragma unroll
        for (int e = 0; e < 4; e++) out.w[e] = pack2(f2{o[2 * e], o[2 * e + 1]});
        y[p] = out;
    }
}

extern "C" {

const char* gd_nn_elementwise_last_error(void) { return g_err; }

int gd_nn_geglu_forward(void* stream, const void* x, void* y, int64_t rows, int inner)
{
    if (!x || !y) return fail(GD_NN_ERR_INVALID_ARG, "geglu: null pointer");
    if (rows <= 0 || inner <= 0 || inner % 8) return fail(GD_NN_ERR_INVALID_ARG, "geglu: need inner % 8 == 0");
    const int64_t nvec = rows * (inner / 8);
    const int64_t blocks = (nvec + 255) / 256;
    const int grid = (int)(blocks < 16384 ? blocks : 16384);
    hipLaunchKernelGGL(geglu_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)x, (bf16x8*)y, nvec,
                       inner / 8);
    return hipGetLastError() == hipSuccess ? 0 : fail(GD_NN_ERR_HIP, "geglu: launch failed");
}

int gd_nn_add_layernorm_forward(void* stream, const void* x, const void* residual, const void* weight, const void* bias,
                                float eps, void* sum_out, void* y, int64_t rows, int C)
{
    if (!x || !weight || !bias || !y) return fail(GD_NN_ERR_INVALID_ARG, "add_layernorm: null pointer");
    if (rows <= 0 || C <= 0 || C % 8 || C > 64 * 8 * 4)
        return fail(GD_NN_ERR_INVALID_ARG, "add_layernorm: need C % 8 == 0 and C <= 2048");
    if (sum_out && !residual) return fail(GD_NN_ERR_INVALID_ARG, "add_layernorm: sum_out without residual");
    const int vpr = C / 8;
    const int vpl = (vpr + 63) / 64;
    const int64_t blocks = (rows + 3) / 4;
    if (blocks > 2147483647LL) return fail(GD_NN_ERR_INVALID_ARG, "add_layernorm: too many rows");
#define GD_LN(V_)                                                                                                      \
    hipLaunchKernelGGL(add_layernorm_kernel<V_>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,            \
                       (const bf16x8*)x, (const bf16x8*)residual, (const bf16x8*)weight, (const bf16x8*)bias,          \
                       (bf16x8*)sum_out, (bf16x8*)y, rows, vpr, eps)
    if (vpl == 1) GD_LN(1);
    else if (vpl == 2) GD_LN(2);
    else if (vpl == 3) GD_LN(3);
    else GD_LN(4);
#undef GD_LN
    return hipGetLastError() == hipSuccess ? 0 : fail(GD_NN_ERR_HIP, "add_layernorm: launch failed");
}

int gd_nn_geglu_backward(void* stream, const void* x, const void* dy, void* dx, int64_t rows, int inner)
{
    if (!x || !dy || !dx) return fail(GD_NN_ERR_INVALID_ARG, "geglu_backward: null pointer");
    if (rows <= 0 || inner <= 0 || inner % 8) return fail(GD_NN_ERR_INVALID_ARG, "geglu_backward: need inner % 8 == 0");
    const int64_t nvec = rows * (inner / 8);
    const int64_t blocks = (nvec + 255) / 256;
    const int grid = (int)(blocks < 16384 ? blocks : 16384);
    hipLaunchKernelGGL(geglu_backward_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)x, (const bf16x8*)dy,
                       (bf16x8*)dx, nvec, inner / 8);
    return hipGetLastError() == hipSuccess ? 0 : fail(GD_NN_ERR_HIP, "geglu_backward: launch failed");
}

int gd_nn_layernorm_backward(void* stream, const void* s, const void* dy, const void* weight, const void* ds, void* dx,
                             int64_t rows, int C, float eps)
{
    if (!s || !dy || !weight || !dx) return fail(GD_NN_ERR_INVALID_ARG, "layernorm_backward: null pointer");
    if (rows <= 0 || C <= 0 || C % 8 || C > 2048) return fail(GD_NN_ERR_INVALID_ARG, "layernorm_backward: need C % 8 == 0, C <= 2048");
    const int vpr = C / 8, vpl = (vpr + 63) / 64;
    const int64_t blocks = (rows + 3) / 4;
    if (blocks > 2147483647LL) return fail(GD_NN_ERR_INVALID_ARG, "layernorm_backward: too many rows");
#define GD_LNB(V_)                                                                                                     \
    hipLaunchKernelGGL(layernorm_backward_kernel<V_>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,       \
                       (const bf16x8*)s, (const bf16x8*)dy, (const bf16x8*)weight, (const bf16x8*)ds, (bf16x8*)dx, rows, vpr, eps)
    if (vpl == 1) GD_LNB(1);
    else if (vpl == 2) GD_LNB(2);
    else if (vpl == 3) GD_LNB(3);
    else GD_LNB(4);
#undef GD_LNB
    return hipGetLastError() == hipSuccess ? 0 : fail(GD_NN_ERR_HIP, "layernorm_backward: launch failed");
}

int gd_nn_conv1x1_c8(void* stream, const void* x, const void* weight, const void* bias, void* y, int64_t npix, int transposed)
{
    if (!x || !weight || !y) return fail(GD_NN_ERR_INVALID_ARG, "conv1x1_c8: null pointer");
    if (npix <= 0) return fail(GD_NN_ERR_INVALID_ARG, "conv1x1_c8: need npix > 0");
    const int64_t blocks = (npix + 255) / 256;
    const int grid = (int)(blocks < 16384 ? blocks : 16384);
    if (transposed)
        hipLaunchKernelGGL(conv1x1_c8_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x,
                           (const uint16_t*)weight, (const uint16_t*)nullptr, (u32x4*)y, npix);
    else
        hipLaunchKernelGGL(conv1x1_c8_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x,
                           (const uint16_t*)weight, (const uint16_t*)bias, (u32x4*)y, npix);
    return hipGetLastError() == hipSuccess ? 0 : fail(GD_NN_ERR_HIP, "conv1x1_c8: launch failed");
}

int gd_nn_softmax_rows_forward(void* stream, const void* x, void* y, int64_t rows, int L)
{
    if (!x || !y) return fail(GD_NN_ERR_INVALID_ARG, "softmax_rows: null pointer");
    if (rows <= 0 || L <= 0 || L % 8 || L > 8192) return fail(GD_NN_ERR_INVALID_ARG, "softmax_rows: need rows > 0, L % 8 == 0, L <= 8192");
    const int64_t blocks = (rows + 3) / 4;
    if (blocks > 2147483647LL) return fail(GD_NN_ERR_INVALID_ARG, "softmax_rows: too many rows");
    const int L8 = L / 8, R = (L8 + 63) / 64;
#define GD_SM(R_) hipLaunchKernelGGL(softmax_rows_kernel<R_>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x, (u32x4*)y, rows, L8)
    if (R <= 2) GD_SM(2);
    else if (R <= 4) GD_SM(4);
    else if (R <= 8) GD_SM(8);
    else GD_SM(16);
#undef GD_SM
    return hipGetLastError() == hipSuccess ? 0 : fail(GD_NN_ERR_HIP, "softmax_rows: launch failed");
}

int gd_nn_softmax_rows_backward(void* stream, const void* p, const void* dp, void* ds, int64_t rows, int L)
{
    if (!p || !dp || !ds) return fail(GD_NN_ERR_INVALID_ARG, "softmax_rows_backward: null pointer");
    if (rows <= 0 || L <= 0 || L % 8 || L > 8192) return fail(GD_NN_ERR_INVALID_ARG, "softmax_rows_backward: need rows > 0, L % 8 == 0, L <= 8192");
    const int64_t blocks = (rows + 3) / 4;
    if (blocks > 2147483647LL) return fail(GD_NN_ERR_INVALID_ARG, "softmax_rows_backward: too many rows");
    const int L8 = L / 8, R = (L8 + 63) / 64;
#define GD_SMB(R_) hipLaunchKernelGGL(softmax_rows_backward_kernel<R_>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)p, (const u32x4*)dp, (u32x4*)ds, rows, L8)
    if (R <= 2) GD_SMB(2);
    else if (R <= 4) GD_SMB(4);
    else if (R <= 8) GD_SMB(8);
    else GD_SMB(16);
#undef GD_SMB
    return hipGetLastError() == hipSuccess ? 0 : fail(GD_NN_ERR_HIP, "softmax_rows_backward: launch failed");
}

}  // extern "C"
