// nn_lora.hip -- the rank-4 adapter branch of the NeTF stage's LoRA UNet, forward and backward, fp32 accumulation throughout.
//
//   y = base + s * up(down(x))         diffusers 0.19 LoRALinearLayer behind LoRAAttnProcessor
//                                      (Garment_Deformer_NeTF/netf/vsd/lora_unet.py:119-160, 415-422; trained by
//                                      netf/trainer.py:215-256)
// x: [M][K] bf16, down: [4][K] fp32, up: [N][4] fp32, base / y: [M][N] bf16.  With rank 4 neither product is a matrix-core
// problem: `down` is four dot products per row (a skinny GEMM the library runs at 30 us per call on 16-wide tiles), `up` is
// four FMAs per output element.  PyTorch ran cast -> GEMM -> GEMM -> cast -> scale -> add per adapted linear (128 of them per
// UNet pass) and twice that in the backward pass; here:
//   gd_nn_lora_rowdot      h[m][r] = s * sum_k a[m][k] w(r, k)          one wave per row; w as [4][K] (down) or [K][4] (up, for dh)
//   gd_nn_lora_rank4_add   y[m][n] = base[m][n] + sum_r h[m][r] w(r, n) one thread per 8 outputs; w as [N][4] (up) or [4][N] (down, for dx)
//   gd_nn_lora_colreduce   g(r, j) = s * sum_m a[m][j] v[m][r]          weight gradients: per row-chunk partial sums in registers (four
//                                                                      waves per chunk, combined in wave order through LDS),
//                                                                      then a fixed-order sum over the chunks (no atomics:
//                                                                      the LoRA gradients are bitwise reproducible)
//   gd_nn_lora_row_fused  rowdot and rank4_add of one row by the SAME wave in one launch: h never makes the round trip through
//                          HBM before its use, and the pair costs one kernel floor (~4.7 us inside a hipGraph) instead of two --
//                          the branch is 128 adapted projections per UNet pass.  Forward: w1 = down, w2 = up, base = the frozen
//                          projection; backward: a = dy, w1 = up, w2 = down, base = the frozen projection's own dx (so autograd's
//                          add of the two input gradients is gone as well).  Same operations in the same order as the two
//                          kernels: bit-identical results.
// All HBM streams of [M][K] / [M][N] bf16 tensors; 16-byte accesses.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/gd_nn.h"

namespace {

thread_local char g_err[256] = "";
int fail(int code, const char* msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

struct alignas(16) u32x4 { uint32_t w[4]; };
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi)
{
    f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float lo16(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi16(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// rows per partial sum of the weight-gradient reduction: 32, or M / 64 for long token sequences (at most 64 chunks: the
// fixed-order second stage is a serial loop over them)
__host__ __device__ inline int chunk_rows(int64_t M)
{
    const int64_t r = ((M + 63) / 64 + 31) / 32 * 32;
    return (int)(r < 32 ? 32 : r);
}

// h[m] = scale * (a[m][:] . w(r, :)), r = 0..3.  WT = false: w is [4][K]; WT = true: w is [K][4].
template <bool WT>
__global__ __launch_bounds__(256) void lora_rowdot_kernel(const u32x4* __restrict__ a, const float* __restrict__ w,
                                                          float4* __restrict__ h, int M, int K8, float scale)
{
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int K = K8 * 8;
    for (int k8 = lane; k8 < K8; k8 += 64) {
        const u32x4 q = a[(size_t)m * K8 + k8];
        float x[8];
#pragma unroll
        for (int e = 0; e < 4; e++) { x[2 * e] = lo16(q.w[e]); x[2 * e + 1] = hi16(q.w[e]); }
        if (WT) {
            const float4* wp = (const float4*)w + (size_t)k8 * 8;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float4 t = wp[e];
                acc[0] = fmaf(x[e], t.x, acc[0]); acc[1] = fmaf(x[e], t.y, acc[1]);
                acc[2] = fmaf(x[e], t.z, acc[2]); acc[3] = fmaf(x[e], t.w, acc[3]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float4 t0 = *(const float4*)(w + (size_t)r * K + k8 * 8), t1 = *(const float4*)(w + (size_t)r * K + k8 * 8 + 4);
                acc[r] = fmaf(x[0], t0.x, fmaf(x[1], t0.y, fmaf(x[2], t0.z, fmaf(x[3], t0.w, acc[r]))));
                acc[r] = fmaf(x[4], t1.x, fmaf(x[5], t1.y, fmaf(x[6], t1.z, fmaf(x[7], t1.w, acc[r]))));
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[r] += __shfl_xor(acc[r], off, 64);
    if (lane == 0) h[m] = make_float4(scale * acc[0], scale * acc[1], scale * acc[2], scale * acc[3]);
}

// y[m][n..n+8) = (base ? base : 0) + sum_r h[m][r] * w(r, n).  WT = true: w is [N][4]; WT = false: w is [4][N].
template <bool WT>
__global__ __launch_bounds__(256) void lora_rank4_add_kernel(const float4* __restrict__ h, const float* __restrict__ w,
                                                             const u32x4* __restrict__ base, u32x4* __restrict__ y,
                                                             int64_t nvec, int N8)
{
    const int N = N8 * 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / N8;
        const int n8 = (int)(i - m * N8);
        const float4 hv = h[m];
        float o[8];
        if (WT) {
            const float4* wp = (const float4*)w + (size_t)n8 * 8;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float4 t = wp[e];
                o[e] = fmaf(hv.x, t.x, fmaf(hv.y, t.y, fmaf(hv.z, t.z, hv.w * t.w)));
            }
        } else {
            const float hr[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = 0.f;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float4 t0 = *(const float4*)(w + (size_t)r * N + n8 * 8), t1 = *(const float4*)(w + (size_t)r * N + n8 * 8 + 4);
                o[0] = fmaf(hr[r], t0.x, o[0]); o[1] = fmaf(hr[r], t0.y, o[1]); o[2] = fmaf(hr[r], t0.z, o[2]); o[3] = fmaf(hr[r], t0.w, o[3]);
                o[4] = fmaf(hr[r], t1.x, o[4]); o[5] = fmaf(hr[r], t1.y, o[5]); o[6] = fmaf(hr[r], t1.z, o[6]); o[7] = fmaf(hr[r], t1.w, o[7]);
            }
        }
        if (base) {
            const u32x4 b = base[i];
#pragma unroll
            for (int e = 0; e < 4; e++) { o[2 * e] += lo16(b.w[e]); o[2 * e + 1] += hi16(b.w[e]); }
        }
        u32x4 out;
#pragma unroll
        for (int e = 0; e < 4; e++) out.w[e] = pack_bf16(o[2 * e], o[2 * e + 1]);
        y[i] = out;
    }
}

// One wave per row m: h = scale * (a[m][:] . w1(r, :)) exactly as lora_rowdot_kernel<BWD> sums it (every lane ends the xor
// butterfly with the same bits), stored for the weight-gradient reductions / the backward pass, then
// y[m][:] = base[m][:] + sum_r h_r w2(r, :) exactly as lora_rank4_add_kernel<!BWD> evaluates it.
// BWD = false: w1 = down [4][K], w2 = up [N][4].  BWD = true: w1 = up [K][4], w2 = down [4][N].
template <bool BWD>
__global__ __launch_bounds__(256) void lora_row_fused_kernel(const u32x4* __restrict__ a, const float* __restrict__ w1,
                                                             const float* __restrict__ w2, const u32x4* __restrict__ base,
                                                             float4* __restrict__ h, u32x4* __restrict__ y, int M, int K8, int N8,
                                                             float scale)
{
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int K = K8 * 8, N = N8 * 8;
    // the eight weight vectors of output chunk n8: !BWD: w2[n8 * 8 + e][0..3]; BWD: w2[r][n8 * 8 .. + 7] as (2 r, 2 r + 1)
    auto load_w2 = [&](int n8, float4 (&wv)[8]) {
        if (!BWD) {
            const float4* wp = (const float4*)w2 + (size_t)n8 * 8;
#pragma unroll
            for (int e = 0; e < 8; e++) wv[e] = wp[e];
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                wv[2 * r] = *(const float4*)(w2 + (size_t)r * N + n8 * 8);
                wv[2 * r + 1] = *(const float4*)(w2 + (size_t)r * N + n8 * 8 + 4);
            }
        }
    };
    // the launch is a latency chain (a row is a few hundred bytes): the first output chunk's base vector and weights are
    // requested BEFORE the dot products, so the two memory round trips of a row overlap
    u32x4 b0 = {{0u, 0u, 0u, 0u}};
    float4 w0[8];
    if (lane < N8) {
        if (base) b0 = base[(size_t)m * N8 + lane];
        load_w2(lane, w0);
    }
    for (int k8 = lane; k8 < K8; k8 += 64) {
        const u32x4 q = a[(size_t)m * K8 + k8];
        float x[8];
#pragma unroll
        for (int e = 0; e < 4; e++) { x[2 * e] = lo16(q.w[e]); x[2 * e + 1] = hi16(q.w[e]); }
        if (BWD) {
            const float4* wp = (const float4*)w1 + (size_t)k8 * 8;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float4 t = wp[e];
                acc[0] = fmaf(x[e], t.x, acc[0]); acc[1] = fmaf(x[e], t.y, acc[1]);
                acc[2] = fmaf(x[e], t.z, acc[2]); acc[3] = fmaf(x[e], t.w, acc[3]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float4 t0 = *(const float4*)(w1 + (size_t)r * K + k8 * 8), t1 = *(const float4*)(w1 + (size_t)r * K + k8 * 8 + 4);
                acc[r] = fmaf(x[0], t0.x, fmaf(x[1], t0.y, fmaf(x[2], t0.z, fmaf(x[3], t0.w, acc[r]))));
                acc[r] = fmaf(x[4], t1.x, fmaf(x[5], t1.y, fmaf(x[6], t1.z, fmaf(x[7], t1.w, acc[r]))));
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[r] += __shfl_xor(acc[r], off, 64);
    const float hr[4] = {scale * acc[0], scale * acc[1], scale * acc[2], scale * acc[3]};
    if (h && lane == 0) h[m] = make_float4(hr[0], hr[1], hr[2], hr[3]);
    for (int n8 = lane; n8 < N8; n8 += 64) {
        const size_t i = (size_t)m * N8 + n8;
        float4 wv[8];
        u32x4 b = b0;
        if (n8 == lane) {
#pragma unroll
            for (int e = 0; e < 8; e++) wv[e] = w0[e];
        } else {
            load_w2(n8, wv);
            if (base) b = base[i];
        }
        float o[8];
        if (!BWD) {
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = fmaf(hr[0], wv[e].x, fmaf(hr[1], wv[e].y, fmaf(hr[2], wv[e].z, hr[3] * wv[e].w)));
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = 0.f;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float4 t0 = wv[2 * r], t1 = wv[2 * r + 1];
                o[0] = fmaf(hr[r], t0.x, o[0]); o[1] = fmaf(hr[r], t0.y, o[1]); o[2] = fmaf(hr[r], t0.z, o[2]); o[3] = fmaf(hr[r], t0.w, o[3]);
                o[4] = fmaf(hr[r], t1.x, o[4]); o[5] = fmaf(hr[r], t1.y, o[5]); o[6] = fmaf(hr[r], t1.z, o[6]); o[7] = fmaf(hr[r], t1.w, o[7]);
            }
        }
        if (base) {
#pragma unroll
            for (int e = 0; e < 4; e++) { o[2 * e] += lo16(b.w[e]); o[2 * e + 1] += hi16(b.w[e]); }
        }
        u32x4 out;
#pragma unroll
        for (int e = 0; e < 4; e++) out.w[e] = pack_bf16(o[2 * e], o[2 * e + 1]);
        y[i] = out;
    }
}

// part[chunk][r][j] = sum over the chunk's rows of a[m][j] * v[m][r].  A workgroup = 512 columns x one row chunk; thread = 8
// columns; its four waves take the rows m0 + w, m0 + w + 4, ... (a quarter of the serial row loop each -- these launches are
// latency chains, not bandwidth) and their 32 partial sums per thread are added in wave order 0, 1, 2, 3 through LDS: a fixed
// order, so the gradients stay bitwise reproducible.
__device__ __forceinline__ void colreduce_body(const u32x4* __restrict__ a, const float4* __restrict__ v, float* __restrict__ part,
                                               int M, int J8, int rows, int chunk, int colblock, float (*sred)[32][64])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j8 = colblock * 64 + lane;
    const bool live = j8 < J8;
    const int m0 = chunk * rows, m1 = min(M, m0 + rows);
    float acc[4][8];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int e = 0; e < 8; e++) acc[r][e] = 0.f;
    if (live) {
#pragma unroll 4
        for (int m = m0 + wave; m < m1; m += 4) {
            const u32x4 q = a[(size_t)m * J8 + j8];
            const float4 t = v[m];
            float x[8];
#pragma unroll
            for (int e = 0; e < 4; e++) { x[2 * e] = lo16(q.w[e]); x[2 * e + 1] = hi16(q.w[e]); }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                acc[0][e] = fmaf(x[e], t.x, acc[0][e]); acc[1][e] = fmaf(x[e], t.y, acc[1][e]);
                acc[2][e] = fmaf(x[e], t.z, acc[2][e]); acc[3][e] = fmaf(x[e], t.w, acc[3][e]);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int e = 0; e < 8; e++) sred[wave - 1][r * 8 + e][lane] = acc[r][e];
    }
    __syncthreads();
    if (wave == 0 && live) {
#pragma unroll
        for (int w = 0; w < 3; w++)
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int e = 0; e < 8; e++) acc[r][e] += sred[w][r * 8 + e][lane];
        const int J = J8 * 8;
        float* p = part + ((size_t)chunk * 4) * J + (size_t)j8 * 8;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            *(float4*)(p + (size_t)r * J) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
            *(float4*)(p + (size_t)r * J + 4) = make_float4(acc[r][4], acc[r][5], acc[r][6], acc[r][7]);
        }
    }
}

// sum over the chunks, IN CHUNK ORDER, of part[c * stride + i]: eight independent loads in flight, added one after the other
__device__ __forceinline__ float chunk_sum(const float* __restrict__ part, size_t stride, int i, int chunks)
{
    float s = 0.f;
    int c = 0;
    for (; c + 8 <= chunks; c += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) t[u] = part[(size_t)(c + u) * stride + i];
#pragma unroll
        for (int u = 0; u < 8; u++) s += t[u];
    }
    for (; c < chunks; c++) s += part[(size_t)c * stride + i];
    return s;
}

__global__ __launch_bounds__(256) void lora_colreduce_kernel(const u32x4* __restrict__ a, const float4* __restrict__ v,
                                                              float* __restrict__ part, int M, int J8, int rows)
{
    __shared__ float sred[3][32][64];
    colreduce_body(a, v, part, M, J8, rows, blockIdx.y, blockIdx.x, sred);
}

// g = scale * sum over chunks (in chunk order) of part[chunk][r][j]; out as [4][J] (transposed = 0) or [J][4] (transposed = 1)
__global__ __launch_bounds__(256) void lora_colreduce_finish_kernel(const float* __restrict__ part, float* __restrict__ g,
                                                                    int chunks, int J, float scale, int transposed)
{
    const int i = blockIdx.x * 256 + threadIdx.x;      // r * J + j
    if (i >= 4 * J) return;
    const float s = chunk_sum(part, (size_t)4 * J, i, chunks);
    const int r = i / J, j = i - r * J;
    g[transposed ? (size_t)j * 4 + r : (size_t)i] = scale * s;
}

// The two weight gradients of one adapter in ONE launch each stage (blockIdx.z / the output index selects the problem):
// problem 0 = d up [N][4] from (dy, scale * h), problem 1 = d down [4][K] from (x, dh).
struct ColPair {
    const u32x4* a[2];
    const float4* v[2];
    float* part[2];
    float* g[2];
    int J8[2];
    int acc;      // 1: g += the sum (a gradient buffer that accumulates over backward passes, torch's .grad convention)
};

__global__ __launch_bounds__(256) void lora_colreduce_pair_kernel(ColPair p, int M, int rows)
{
    __shared__ float sred[3][32][64];
    const int z = blockIdx.z;
    if ((int)blockIdx.x * 64 >= p.J8[z]) return;       // whole workgroup: no barrier is skipped by part of it
    colreduce_body(p.a[z], p.v[z], p.part[z], M, p.J8[z], rows, blockIdx.y, blockIdx.x, sred);
}

__global__ __launch_bounds__(256) void lora_colreduce_pair_finish_kernel(ColPair p, int chunks)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    const int n0 = 4 * 8 * p.J8[0], n1 = 4 * 8 * p.J8[1];
    if (i >= n0 + n1) return;
    const int z = i >= n0;
    if (z) i -= n0;
    const int J = 8 * p.J8[z];
    const float s = chunk_sum(p.part[z], (size_t)4 * J, i, chunks);
    const int r = i / J, j = i - r * J;
    float* dst = p.g[z] + (z == 0 ? (size_t)j * 4 + r : (size_t)i);          // d up as [N][4], d down as [4][K]
    *dst = p.acc ? *dst + s : s;
}

// The weight gradients of MANY adapters in one launch per stage (round 6): the 128-260 adapted projections of a LoRA UNet backward
// each ended with a pair-reduction launch and its finish launch of ~5 us -- 2.6 ms of launch-bound kernels on the training chain of
// the NeTF iteration, none of whose results anybody reads before the optimizer step.  The backward nodes now only RECORD their
// problem (gd_nn_lora_colreduce_group_desc), and the last one launches the table, 32 adapters per launch: blockIdx.z = 2 * adapter + (0: d up, 1: d down),
// grids sized for the largest adapter, workgroups beyond an adapter's own extent leave at once.  Same bodies, same summation
// orders: bit-identical to the per-adapter launches.
struct ColGroupEntry {
    ColPair p;
    int M, rows, chunks, pad;
};

constexpr int kGroupChunk = 32;        // adapters per launch: the table travels BY VALUE in the kernel arguments (3 KiB of the 4 KiB
struct ColGroupArgs {                  // limit) -- no device table, no host-to-device copy, nothing that a hipGraph capture forbids
    ColGroupEntry e[kGroupChunk];
};

__global__ __launch_bounds__(256) void lora_colreduce_group_kernel(const ColGroupArgs args)
{
    __shared__ float sred[3][32][64];
    const ColGroupEntry& e = args.e[blockIdx.z >> 1];
    const int z = blockIdx.z & 1;
    if ((int)blockIdx.x * 64 >= e.p.J8[z] || (int)blockIdx.y >= e.chunks) return;   // whole workgroup
    colreduce_body(e.p.a[z], e.p.v[z], e.p.part[z], e.M, e.p.J8[z], e.rows, blockIdx.y, blockIdx.x, sred);
}

__global__ __launch_bounds__(256) void lora_colreduce_group_finish_kernel(const ColGroupArgs args)
{
    const ColGroupEntry& e = args.e[blockIdx.y];
    int i = blockIdx.x * 256 + threadIdx.x;
    const int n0 = 4 * 8 * e.p.J8[0], n1 = 4 * 8 * e.p.J8[1];
    if (i >= n0 + n1) return;
    const int z = i >= n0;
    if (z) i -= n0;
    const int J = 8 * e.p.J8[z];
    const float s = chunk_sum(e.p.part[z], (size_t)4 * J, i, e.chunks);
    const int r = i / J, j = i - r * J;
    float* dst = e.p.g[z] + (z == 0 ? (size_t)j * 4 + r : (size_t)i);          // d up as [N][4], d down as [4][K]
    *dst = e.p.acc ? *dst + s : s;
}

}  // namespace

extern "C" {

size_t gd_nn_lora_colreduce_group_entry_bytes(void) { return sizeof(ColGroupEntry); }

int gd_nn_lora_colreduce_group_desc(void* entry, const void* dy, const float* hs, const void* x, const float* dh, float* scratch,
                                    float* d_up, float* d_down, int64_t M, int N, int K, int accumulate, int* grid_xyz)
{
    if (!entry || !dy || !hs || !x || !dh || !scratch || !d_up || !d_down || !grid_xyz)
        return fail(GD_NN_ERR_INVALID_ARG, "lora_colreduce_group_desc: null pointer");
    if (M <= 0 || M > 0x7fffffff / 4 || N <= 0 || K <= 0 || (N & 7) || (K & 7))
        return fail(GD_NN_ERR_INVALID_ARG, "lora_colreduce_group_desc: need M > 0, N % 8 == 0, K % 8 == 0");
    ColGroupEntry e;
    e.rows = chunk_rows(M);
    e.chunks = (int)((M + e.rows - 1) / e.rows);
    e.M = (int)M;
    e.pad = 0;
    e.p.a[0] = (const u32x4*)dy; e.p.v[0] = (const float4*)hs; e.p.part[0] = scratch; e.p.g[0] = d_up; e.p.J8[0] = N / 8;
    e.p.a[1] = (const u32x4*)x; e.p.v[1] = (const float4*)dh; e.p.part[1] = scratch + (size_t)e.chunks * 4 * N; e.p.g[1] = d_down;
    e.p.J8[1] = K / 8;
    e.p.acc = accumulate ? 1 : 0;
    memcpy(entry, &e, sizeof(e));
    const int jmax = N > K ? N / 8 : K / 8;
    grid_xyz[0] = (jmax + 63) / 64;             // stage 1: x
    grid_xyz[1] = e.chunks;                     // stage 1: y
    grid_xyz[2] = (4 * (N + K) + 255) / 256;    // stage 2: x
    return GD_NN_OK;
}

int gd_nn_lora_colreduce_group_launch(void* stream, const void* table_host, int n_entries, int grid1_x, int grid1_y, int grid2_x)
{
    if (!table_host || n_entries <= 0 || grid1_x <= 0 || grid1_y <= 0 || grid2_x <= 0)
        return fail(GD_NN_ERR_INVALID_ARG, "lora_colreduce_group_launch: bad arguments");
    const ColGroupEntry* t = (const ColGroupEntry*)table_host;
    // stage 1 of every chunk, then stage 2 of every chunk: the finish launches read what the reduction launches wrote
    for (int pass = 0; pass < 2; pass++)
        for (int i0 = 0; i0 < n_entries; i0 += kGroupChunk) {
            const int n = n_entries - i0 < kGroupChunk ? n_entries - i0 : kGroupChunk;
            ColGroupArgs a;
            memcpy(a.e, t + i0, sizeof(ColGroupEntry) * (size_t)n);
            if (pass == 0)
                hipLaunchKernelGGL(lora_colreduce_group_kernel, dim3(grid1_x, grid1_y, 2 * n), dim3(256), 0, (hipStream_t)stream, a);
            else
                hipLaunchKernelGGL(lora_colreduce_group_finish_kernel, dim3(grid2_x, n), dim3(256), 0, (hipStream_t)stream, a);
        }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

const char* gd_nn_lora_last_error(void) { return g_err; }

int gd_nn_lora_rowdot(void* stream, const void* a, const float* w, float* h, int64_t M, int K, float scale, int w_is_k_by_4)
{
    if (!a || !w || !h) return fail(GD_NN_ERR_INVALID_ARG, "lora_rowdot: null pointer");
    if (M <= 0 || M > 0x7fffffff / 4 || K <= 0 || (K & 7)) return fail(GD_NN_ERR_INVALID_ARG, "lora_rowdot: need M > 0 and K % 8 == 0");
    const dim3 grid((unsigned)((M + 3) / 4));
    if (w_is_k_by_4)
        hipLaunchKernelGGL(lora_rowdot_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const u32x4*)a, w, (float4*)h, (int)M, K / 8, scale);
    else
        hipLaunchKernelGGL(lora_rowdot_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const u32x4*)a, w, (float4*)h, (int)M, K / 8, scale);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

int gd_nn_lora_rank4_add(void* stream, const float* h, const float* w, const void* base, void* y, int64_t M, int N, int w_is_n_by_4)
{
    if (!h || !w || !y) return fail(GD_NN_ERR_INVALID_ARG, "lora_rank4_add: null pointer");
    if (M <= 0 || N <= 0 || (N & 7)) return fail(GD_NN_ERR_INVALID_ARG, "lora_rank4_add: need M > 0 and N % 8 == 0");
    const int64_t nvec = M * (N / 8);
    const int64_t blocks = (nvec + 255) / 256;
    const dim3 grid((unsigned)(blocks < 16384 ? blocks : 16384));
    if (w_is_n_by_4)
        hipLaunchKernelGGL(lora_rank4_add_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const float4*)h, w, (const u32x4*)base, (u32x4*)y, nvec, N / 8);
    else
        hipLaunchKernelGGL(lora_rank4_add_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const float4*)h, w, (const u32x4*)base, (u32x4*)y, nvec, N / 8);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

int gd_nn_lora_row_fused(void* stream, const void* a, const float* w1, const float* w2, const void* base, float* h, void* y,
                         int64_t M, int K, int N, float scale, int backward)
{
    if (!a || !w1 || !w2 || !y) return fail(GD_NN_ERR_INVALID_ARG, "lora_row_fused: null pointer");
    if (M <= 0 || M > 0x7fffffff / 4 || K <= 0 || N <= 0 || (K & 7) || (N & 7))
        return fail(GD_NN_ERR_INVALID_ARG, "lora_row_fused: need M > 0, K % 8 == 0, N % 8 == 0");
    const dim3 grid((unsigned)((M + 3) / 4));
    if (backward)
        hipLaunchKernelGGL(lora_row_fused_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const u32x4*)a, w1, w2,
                           (const u32x4*)base, (float4*)h, (u32x4*)y, (int)M, K / 8, N / 8, scale);
    else
        hipLaunchKernelGGL(lora_row_fused_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const u32x4*)a, w1, w2,
                           (const u32x4*)base, (float4*)h, (u32x4*)y, (int)M, K / 8, N / 8, scale);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

size_t gd_nn_lora_colreduce_scratch_floats(int64_t M, int J)
{
    if (M <= 0 || J <= 0) return 0;
    const int rows = chunk_rows(M);
    return (size_t)((M + rows - 1) / rows) * 4 * (size_t)J;
}

int gd_nn_lora_colreduce(void* stream, const void* a, const float* v, float* scratch, float* g, int64_t M, int J, float scale,
                         int g_is_j_by_4)
{
    if (!a || !v || !scratch || !g) return fail(GD_NN_ERR_INVALID_ARG, "lora_colreduce: null pointer");
    if (M <= 0 || M > 0x7fffffff / 4 || J <= 0 || (J & 7)) return fail(GD_NN_ERR_INVALID_ARG, "lora_colreduce: need M > 0 and J % 8 == 0");
    const int rows = chunk_rows(M);
    const int chunks = (int)((M + rows - 1) / rows);
    const int J8 = J / 8;
    hipLaunchKernelGGL(lora_colreduce_kernel, dim3((J8 + 63) / 64, chunks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)a,
                       (const float4*)v, scratch, (int)M, J8, rows);
    hipLaunchKernelGGL(lora_colreduce_finish_kernel, dim3((4 * J + 255) / 256), dim3(256), 0, (hipStream_t)stream, scratch, g, chunks, J,
                       scale, g_is_j_by_4);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

size_t gd_nn_lora_colreduce_pair_scratch_floats(int64_t M, int N, int K)
{
    return gd_nn_lora_colreduce_scratch_floats(M, N) + gd_nn_lora_colreduce_scratch_floats(M, K);
}

int gd_nn_lora_colreduce_pair(void* stream, const void* dy, const float* hs, const void* x, const float* dh, float* scratch,
                              float* d_up, float* d_down, int64_t M, int N, int K)
{
    return gd_nn_lora_colreduce_pair_into(stream, dy, hs, x, dh, scratch, d_up, d_down, M, N, K, 0);
}

int gd_nn_lora_colreduce_pair_into(void* stream, const void* dy, const float* hs, const void* x, const float* dh, float* scratch,
                                   float* d_up, float* d_down, int64_t M, int N, int K, int accumulate)
{
    if (!dy || !hs || !x || !dh || !scratch || !d_up || !d_down) return fail(GD_NN_ERR_INVALID_ARG, "lora_colreduce_pair: null pointer");
    if (M <= 0 || M > 0x7fffffff / 4 || N <= 0 || K <= 0 || (N & 7) || (K & 7))
        return fail(GD_NN_ERR_INVALID_ARG, "lora_colreduce_pair: need M > 0, N % 8 == 0, K % 8 == 0");
    const int rows = chunk_rows(M);
    const int chunks = (int)((M + rows - 1) / rows);
    ColPair p;
    p.a[0] = (const u32x4*)dy; p.v[0] = (const float4*)hs; p.part[0] = scratch; p.g[0] = d_up; p.J8[0] = N / 8;
    p.a[1] = (const u32x4*)x; p.v[1] = (const float4*)dh; p.part[1] = scratch + (size_t)chunks * 4 * N; p.g[1] = d_down; p.J8[1] = K / 8;
    p.acc = accumulate ? 1 : 0;
    const int jmax = N > K ? N / 8 : K / 8;
    hipLaunchKernelGGL(lora_colreduce_pair_kernel, dim3((jmax + 63) / 64, chunks, 2), dim3(256), 0, (hipStream_t)stream, p, (int)M, rows);
    hipLaunchKernelGGL(lora_colreduce_pair_finish_kernel, dim3((4 * (N + K) + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, chunks);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

}  // extern "C"
