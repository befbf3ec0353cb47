// nn_prologue.hip -- the two small fused passes either side of the rasterizer in the SDS iteration:
//
//   gd_nn_vae_prologue_forward / _backward
//       rgb_BCHW_512 = F.interpolate(rgb_BCHW, (512, 512), mode="bilinear", align_corners=False)
//       imgs = rgb_BCHW_512 * 2 - 1                              (threestudio stable_diffusion_guidance.py:394-396, :164)
//       + the cast to the VAE's bf16 / NHWC layout that its first convolution (conv3x3_first_kernel) reads.
//     PyTorch: upsample_bilinear2d (a full pass even at 512 -> 512), mul/add, dtype copy, layout copy and their four
//     backward kernels -- about ten launches and ten passes over the largest fp32 images of the step.  Here: one
//     kernel each way.  Input: planar fp32 [N,3,H,W] exactly as the rasterizer writes it; output NHWC bf16
//     [N,OH,OW,3]; backward takes the NHWC bf16 gradient with CG >= 3 channels per pixel (the first conv's dgrad is
//     computed on 4 zero-padded channels) and GATHERS, per source pixel, the output pixels whose bilinear footprint
//     covers it (no atomics; 1:1 at 512 -> 512, 2x2 at 1024 -> 512).
//
//   gd_nn_sparsity_forward / _backward
//       opacity = depth / (depth.max() + 1e-5);  loss_sparsity = mean(sqrt(opacity^2 + 0.01))
//                                                                (threestudio systems/GaussianDreamer.py:215,253)
//     The maximum (a batch-wide -- with view sharding a GLOBAL -- reduction) stays a tensor input so that its
//     gradient keeps flowing through torch's max / the all-reduce wrapper; everything after it is one kernel forward
//     (the value and d loss / d max) and one backward, instead of div, pow, add, sqrt, mean and their backward nodes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gd_nn.h"

namespace {

thread_local char g_err[256] = "";
int fail(int code, const char* msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

__device__ __forceinline__ uint16_t f2bf(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

// source coordinate of output index o (PyTorch area_pixel_compute_source_index, align_corners = False, bilinear)
__device__ __forceinline__ void src_index(int o, float scale, int in_size, int& i0, int& i1, float& w1)
{
    float s = ((float)o + 0.5f) * scale - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    w1 = s - (float)i0;
}

__global__ __launch_bounds__(256) void vae_prologue_fwd_kernel(const float* __restrict__ x, uint16_t* __restrict__ y,
                                                               int N, int H, int W, int OH, int OW, float sh, float sw)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)N * OH * OW;
    if (idx >= total) return;
    const int ox = (int)(idx % OW);
    const int oy = (int)((idx / OW) % OH);
    const int n = (int)(idx / ((int64_t)OW * OH));
    int y0, y1, x0, x1;
    float wy, wx;
    src_index(oy, sh, H, y0, y1, wy);
    src_index(ox, sw, W, x0, x1, wx);
    const float* p = x + (size_t)n * 3 * H * W;
    uint16_t o[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float* pc = p + (size_t)c * H * W;
        const float v00 = pc[(size_t)y0 * W + x0], v01 = pc[(size_t)y0 * W + x1];
        const float v10 = pc[(size_t)y1 * W + x0], v11 = pc[(size_t)y1 * W + x1];
        // upsample_bilinear2d's own expression order: rows blended first, then the two rows
        const float v = (1.f - wy) * ((1.f - wx) * v00 + wx * v01) + wy * ((1.f - wx) * v10 + wx * v11);
        o[c] = f2bf(v * 2.0f - 1.0f);
    }
    uint16_t* q = y + (size_t)idx * 3;
    q[0] = o[0]; q[1] = o[1]; q[2] = o[2];
}

// One thread per SOURCE pixel and image: adds dY over the output pixels that read it (transpose of the forward).
__global__ __launch_bounds__(256) void vae_prologue_bwd_kernel(const uint16_t* __restrict__ dy, float* __restrict__ dx,
                                                               int N, int H, int W, int OH, int OW, int CG, float sh,
                                                               float sw)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)N * H * W;
    if (idx >= total) return;
    const int sx = (int)(idx % W);
    const int sy = (int)((idx / W) % H);
    const int n = (int)(idx / ((int64_t)W * H));
    // output rows / columns o with floor(src(o)) in {s - 1, s}:  (s - 1 + 0.5) / scale - 0.5 < o + ... (conservative)
    const int oy_lo = max(0, (int)floorf(((float)sy - 0.5f) / sh - 0.5f) - 1);
    const int oy_hi = min(OH - 1, (int)ceilf(((float)sy + 1.5f) / sh - 0.5f) + 1);
    const int ox_lo = max(0, (int)floorf(((float)sx - 0.5f) / sw - 0.5f) - 1);
    const int ox_hi = min(OW - 1, (int)ceilf(((float)sx + 1.5f) / sw - 0.5f) + 1);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; oy++) {
        int y0, y1;
        float wy;
        src_index(oy, sh, H, y0, y1, wy);
        const float cy = (y0 == sy ? 1.f - wy : 0.f) + (y1 == sy ? wy : 0.f);
        if (cy == 0.f) continue;
        for (int ox = ox_lo; ox <= ox_hi; ox++) {
            int x0, x1;
            float wx;
            src_index(ox, sw, W, x0, x1, wx);
            const float cx = (x0 == sx ? 1.f - wx : 0.f) + (x1 == sx ? wx : 0.f);
            if (cx == 0.f) continue;
            const uint16_t* q = dy + ((size_t)((size_t)n * OH + oy) * OW + ox) * CG;
            const float wgt = 2.0f * cy * cx;        // d(2 v - 1)/dv = 2
            a0 += wgt * bf2f(q[0]); a1 += wgt * bf2f(q[1]); a2 += wgt * bf2f(q[2]);
        }
    }
    float* p = dx + (size_t)n * 3 * H * W + (size_t)sy * W + sx;
    p[0] = a0; p[(size_t)H * W] = a1; p[2 * (size_t)H * W] = a2;
}

// ---- sparsity head -----------------------------------------------------------------------------------------------
// out[0] = sum f(x), out[1] = sum f'(x) x  (fp64 accumulators, zeroed by the caller), x = depth / (max + 1e-5),
// f = sqrt(x^2 + 0.01)
__global__ __launch_bounds__(256) void sparsity_fwd_kernel(const float* __restrict__ depth, const float* __restrict__ dmax,
                                                           int64_t n, double* __restrict__ out)
{
    __shared__ double s0[4], s1[4];
    const float inv = 1.0f / (dmax[0] + 1e-5f);
    double a = 0.0, b = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float x = depth[i] * inv;
        const float f = sqrtf(x * x + 0.01f);
        a += (double)f;
        b += (double)(x * x / f);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off, 64);
        b += __shfl_xor(b, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { s0[threadIdx.x >> 6] = a; s1[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(out, s0[0] + s0[1] + s0[2] + s0[3]);
        atomicAdd(out + 1, s1[0] + s1[1] + s1[2] + s1[3]);
    }
}

// d_depth[i] = g / n * f'(x_i) / (max + eps)
__global__ __launch_bounds__(256) void sparsity_bwd_kernel(const float* __restrict__ depth, const float* __restrict__ dmax,
                                                           const float* __restrict__ gout, int64_t n,
                                                           float* __restrict__ d_depth)
{
    const float inv = 1.0f / (dmax[0] + 1e-5f);
    const float k = gout[0] / (float)n * inv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float x = depth[i] * inv;
        d_depth[i] = k * x / sqrtf(x * x + 0.01f);
    }
}

}  // namespace

extern "C" {

const char* gd_nn_prologue_last_error(void) { return g_err; }

int gd_nn_vae_prologue_forward(void* stream, const float* x, void* y, int N, int H, int W, int OH, int OW)
{
    if (!x || !y) return fail(GD_NN_ERR_INVALID_ARG, "vae_prologue: null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return fail(GD_NN_ERR_INVALID_ARG, "vae_prologue: bad shape");
    const int64_t total = (int64_t)N * OH * OW;
    hipLaunchKernelGGL(vae_prologue_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, (uint16_t*)y, N, H, W, OH, OW, (float)H / (float)OH, (float)W / (float)OW);
    return hipGetLastError() == hipSuccess ? 0 : fail(GD_NN_ERR_HIP, "vae_prologue: launch failed");
}

int gd_nn_vae_prologue_backward(void* stream, const void* dy, float* dx, int N, int H, int W, int OH, int OW, int CG)
{
    if (!dy || !dx) return fail(GD_NN_ERR_INVALID_ARG, "vae_prologue backward: null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || CG < 3)
        return fail(GD_NN_ERR_INVALID_ARG, "vae_prologue backward: bad shape");
    const int64_t total = (int64_t)N * H * W;
    hipLaunchKernelGGL(vae_prologue_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)dy, dx, N, H, W, OH, OW, CG, (float)H / (float)OH, (float)W / (float)OW);
    return hipGetLastError() == hipSuccess ? 0 : fail(GD_NN_ERR_HIP, "vae_prologue backward: launch failed");
}

int gd_nn_sparsity_forward(void* stream, const float* depth, const float* dmax, int64_t n, double* sums2)
{
    if (!depth || !dmax || !sums2 || n <= 0) return fail(GD_NN_ERR_INVALID_ARG, "sparsity: bad argument");
    if (hipMemsetAsync(sums2, 0, 2 * sizeof(double), (hipStream_t)stream) != hipSuccess)
        return fail(GD_NN_ERR_HIP, "sparsity: memset failed");
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(sparsity_fwd_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0,
                       (hipStream_t)stream, depth, dmax, n, sums2);
    return hipGetLastError() == hipSuccess ? 0 : fail(GD_NN_ERR_HIP, "sparsity: launch failed");
}

int gd_nn_sparsity_backward(void* stream, const float* depth, const float* dmax, const float* grad_out, int64_t n,
                            float* d_depth)
{
    if (!depth || !dmax || !grad_out || !d_depth || n <= 0) return fail(GD_NN_ERR_INVALID_ARG, "sparsity backward: bad argument");
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(sparsity_bwd_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                       (hipStream_t)stream, depth, dmax, grad_out, n, d_depth);
    return hipGetLastError() == hipSuccess ? 0 : fail(GD_NN_ERR_HIP, "sparsity backward: launch failed");
}

}  // extern "C"
