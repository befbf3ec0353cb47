// nn_groupnorm.hip -- GroupNorm(32) (+ SiLU) on NHWC bf16 activations, forward and input-gradient.
//
// Why hand-written: PyTorch-ROCm's native GroupNorm kernels are NCHW-only, so with channels_last
// convolutions every GroupNorm costs two full-tensor layout copies plus separate moments / affine /
// SiLU kernels (9+ tensor passes).  Here: one statistics pass + one fused normalise*gamma+beta ->
// SiLU pass (3 tensor passes), directly on NHWC.  Pure HBM streams: 16-byte (8 x bf16) vector
// accesses, thread <-> fixed channel vector so gamma/beta/mean/rstd live in registers, fp32 math,
// fp64 global accumulation of the (few) per-workgroup partial sums.
//
// Statistics workspace (gd_nn_groupnorm_ws_bytes): [8 x N*G*2 fp64 accumulators][N u64 tickets].  It must be ZERO when
// a call starts and is zero again when the call's kernels have run: the last statistics workgroup to finish
// (ticket) turns the sums into fp32 results in a separate buffer (mean_rstd / group_sums) and clears what it
// read with atomic exchanges -- no memset and no finalize launch per GroupNorm (105 GroupNorms per SDS step).
// One zero-initialised workspace therefore serves any sequence of calls, of any shapes, on one stream.
//
// Layout: x[n][p][c], p = pixel (H*W), c fastest.  Workgroup = rows x vpp threads where
// vpp = C/8 vectors per pixel and rows = max(1, 256 / vpp); blockIdx.x walks pixel chunks,
// blockIdx.y = n.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gd_nn.h"

namespace {

thread_local char g_err[256] = "";

struct alignas(16) bf16x8 {
    uint16_t v[8];
};

__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                              // round to nearest even
    return (uint16_t)(u >> 16);
}

// ---- packed fp32 helpers.  Measured on MI355X (tools/probes/valu_rate_probe.hip): a wave64 VALU instruction holds
// its SIMD ~4.5 cycles, so these "streaming" passes were VALU-bound, not HBM-bound (SQ_INSTS_VALU x 4.5 cycles = the
// whole kernel time, profiles/r02_pmc.json) -- with an IEEE division (12 instructions) per sigmoid, a six-instruction
// software bf16 rounding per element and scalar fp32 arithmetic.  Here two channels share every arithmetic
// instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32), the sigmoid is exp2 + v_rcp_f32 and the rounding is
// v_cvt_pk_bf16_f32 (round to nearest even, the same bits as f2bf for every non-NaN input).
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
struct alignas(16) u32x4 { uint32_t w[4]; };
__device__ __forceinline__ f2 unpack2(uint32_t w) { return f2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; }
__device__ __forceinline__ uint32_t pack2(f2 v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2_t)); }
__device__ __forceinline__ f2 sigmoid2(f2 z)
{
    const f2 e = z * -1.44269504088896341f;
    const f2 den = 1.0f + f2{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
    return f2{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
}
// group of channel 8 tv + k for k = 0..7 with ONE integer division (cg = channels per group, any value >= 1)
__device__ __forceinline__ void groups_of_vector(int tv, int cg, int (&g)[8])
{
    const int base = tv * 8;
    int q = base / cg, r = base - q * cg;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        g[k] = q;
        if (++r == cg) { r = 0; q++; }
    }
}

constexpr int kMaxThreads = 320;  // C = 2560 -> 320 vectors per pixel
constexpr int kUnroll = 4;        // independent 16-byte loads per thread and loop iteration of the streaming kernels
constexpr int kSlots = 8;         // copies of the statistics accumulators (contention, see reduce_to_groups)

// Sum over the workgroup of per-thread per-channel partials, folded to per-group totals and added
// to ws[n][g][0..1] in fp64.  `a`, `b`: the thread's 8-channel partial sums of two quantities.
// MODE 0: results = {mean, rstd} (forward statistics);  MODE 1: results = {s1 / M, s2 / M} (backward)
template <int MODE>
__device__ __forceinline__ void reduce_to_groups(const float (&a)[8], const float (&b)[8], int vpp, int rows, int tv,
                                                 int tr, int C, int G, int N, int n, double M, float eps,
                                                 double* __restrict__ ws, float* __restrict__ result, float* lds)
{
    // kSlots copies of the accumulators, picked by chunk index: with up to 256 chunks per image all adding to the
    // same 2*G addresses, the (returning) atomics of one copy serialised for ~10 us per workgroup
    double* ws_n = ws + ((size_t)(blockIdx.x % kSlots) * N + n) * G * 2;
    // lds: [2][rows][C]
    float* la = lds;
    float* lb = lds + (size_t)rows * C;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        la[tr * C + tv * 8 + k] = a[k];
        lb[tr * C + tv * 8 + k] = b[k];
    }
    __syncthreads();
    // column sums over the workgroup's rows (parallel over channels), then channels -> groups
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float sa = 0.f, sb = 0.f;
        for (int r = 0; r < rows; r++) {
            sa += la[r * C + c];
            sb += lb[r * C + c];
        }
        la[c] = sa;   // row 0 is only read by this same thread for column c
        lb[c] = sb;
    }
    __syncthreads();
    const int cg = C / G;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        float sa = 0.f, sb = 0.f;
        for (int c = g * cg; c < (g + 1) * cg; c++) {
            sa += la[c];
            sb += lb[c];
        }
        // returning atomics, results consumed: the wave waits until both were performed at the device-coherent
        // point before it reaches the barrier below, which orders them before this workgroup's ticket.  (An
        // agent-scope __threadfence() here costs an L2 write-back + invalidate per workgroup on this multi-XCD
        // part: +9 ms per SDS step when tried.  Every cross-workgroup access below is itself an atomic.)
        const double o1 = atomicAdd(&ws_n[2 * g], (double)sa);
        const double o2 = atomicAdd(&ws_n[2 * g + 1], (double)sb);
        asm volatile("" ::"v"(o1), "v"(o2));
    }
    // last workgroup of this image: sums -> results, and leave accumulators + ticket zero for the next call
    __shared__ unsigned int s_last;
    __syncthreads();
    // one ticket per image: the last workgroup of image n finalises that image's G groups
    unsigned long long* ticket = reinterpret_cast<unsigned long long*>(ws + (size_t)kSlots * N * G * 2) + n;
    if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1ULL) == (unsigned long long)gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    for (int i = n * G + threadIdx.x; i < (n + 1) * G; i += blockDim.x) {
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int sl = 0; sl < kSlots; sl++) {
            unsigned long long* acc = reinterpret_cast<unsigned long long*>(ws + 2 * ((size_t)sl * N * G + i));
            sa += __longlong_as_double((long long)atomicExch(acc, 0ULL));
            sb += __longlong_as_double((long long)atomicExch(acc + 1, 0ULL));
        }
        if (MODE == 0) {
            const double mean = sa / M;
            double var = sb / M - mean * mean;
            var = var < 0 ? 0 : var;
            result[2 * i] = (float)mean;
            result[2 * i + 1] = rsqrtf((float)var + eps);
        } else {
            const float invM = 1.f / (float)M;
            result[2 * i] = (float)sa * invM;
            result[2 * i + 1] = (float)sb * invM;
        }
    }
    if (threadIdx.x == 0) atomicExch(ticket, 0ULL);
}

__global__ void gn_stats_kernel(const bf16x8* __restrict__ x, int HW, int C, int G, int vpp, int rows, int ppb,
                                float eps, double* __restrict__ ws, float* __restrict__ mean_rstd, int N, int n0)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int n = blockIdx.y + n0;
    const int tv = threadIdx.x % vpp, tr = threadIdx.x / vpp;
    const int p0 = blockIdx.x * ppb, p1 = min(HW, p0 + ppb);
    f2 s2[4], ss2[4];
#pragma unroll
    for (int k = 0; k < 4; k++) s2[k] = ss2[k] = f2{0.f, 0.f};
    const bf16x8* xn = x + (size_t)n * HW * vpp;
    auto body = [&](const bf16x8& v) {
        const u32x4 w = __builtin_bit_cast(u32x4, v);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const f2 f = unpack2(w.w[k]);
            s2[k] += f;
            ss2[k] += f * f;
        }
    };
    // kUnroll independent 16-byte loads per thread before any arithmetic: one load per iteration kept ~32 KB in
    // flight per CU, i.e. ~4 TB/s by Little's law; a plain copy reaches 5.5-6 TB/s on these tensors
    int p = p0 + tr;
    for (; p + (kUnroll - 1) * rows < p1; p += kUnroll * rows) {
        bf16x8 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) v[u] = xn[(size_t)(p + u * rows) * vpp + tv];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) body(v[u]);
    }
    for (; p < p1; p += rows) body(xn[(size_t)p * vpp + tv]);
    float s[8], ss[8];
#pragma unroll
    for (int k = 0; k < 4; k++) { s[2 * k] = s2[k].x; s[2 * k + 1] = s2[k].y; ss[2 * k] = ss2[k].x; ss[2 * k + 1] = ss2[k].y; }
    reduce_to_groups<0>(s, ss, vpp, rows, tv, tr, C, G, N, n, (double)HW * (C / G), eps, ws, mean_rstd, lds);
}

__device__ __forceinline__ float silu_f(float z) { return z / (1.f + __expf(-z)); }

__global__ void gn_apply_kernel(const bf16x8* __restrict__ x, bf16x8* __restrict__ y,
                                const uint16_t* __restrict__ gamma, const uint16_t* __restrict__ beta, int HW, int C,
                                int G, int vpp, int rows, int ppb, int apply_silu,
                                const float* __restrict__ mean_rstd)
{
    const int n = blockIdx.y;
    const int tv = threadIdx.x % vpp, tr = threadIdx.x / vpp;
    const int p0 = blockIdx.x * ppb, p1 = min(HW, p0 + ppb);
    const int cg = C / G;
    f2 a[4], b[4];
    {
        int g[8];
        groups_of_vector(tv, cg, g);
        float a1[8], b1[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int c = tv * 8 + k;
            const float mean = mean_rstd[((size_t)n * G + g[k]) * 2];
            const float rstd = mean_rstd[((size_t)n * G + g[k]) * 2 + 1];
            a1[k] = rstd * bf2f(gamma[c]);
            b1[k] = bf2f(beta[c]) - mean * a1[k];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) { a[k] = f2{a1[2 * k], a1[2 * k + 1]}; b[k] = f2{b1[2 * k], b1[2 * k + 1]}; }
    }
    const bf16x8* xn = x + (size_t)n * HW * vpp;
    bf16x8* yn = y + (size_t)n * HW * vpp;
    auto body = [&](const bf16x8& v, int p) {
        const u32x4 w = __builtin_bit_cast(u32x4, v);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            f2 z = unpack2(w.w[k]) * a[k] + b[k];
            if (apply_silu) z = z * sigmoid2(z);
            o.w[k] = pack2(z);
        }
        yn[(size_t)p * vpp + tv] = __builtin_bit_cast(bf16x8, o);
    };
    int p = p0 + tr;
    for (; p + (kUnroll - 1) * rows < p1; p += kUnroll * rows) {      // loads first: see gn_stats_kernel
        bf16x8 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) v[u] = xn[(size_t)(p + u * rows) * vpp + tv];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) body(v[u], p + u * rows);
    }
    for (; p < p1; p += rows) body(xn[(size_t)p * vpp + tv], p);
}

// ---- one-launch GroupNorm(+SiLU) for tensors whose (image, group) slice fits a workgroup's registers ------------
// The two-pass form above (statistics kernel, apply kernel) reads x twice and pays two launches; on the UNet's maps
// (<= 64x64 at <= 16 latents) most of its time is launch latency and the second read.  Here ONE workgroup owns one
// (image, group): its HW x C/G slice (<= 128 KB) is read ONCE into registers -- dword = 2 channels, lane-consecutive
// within a pixel's C/G-channel run, then on to the next pixel (stride C) -- summed (per-thread fp32, cross-thread fp64),
// normalised from the registers and written.  Read 1x + write 1x instead of read 2x + write 1x, one launch instead of
// two, no workspace / atomics (bit-reproducible).  Inference only: the backward pass wants mean / rstd (two-pass form).
template <int THREADS, int R>
__global__ __launch_bounds__(THREADS) void gn_group_fused_kernel(const uint32_t* __restrict__ x, uint32_t* __restrict__ y,
                                                                 const uint16_t* __restrict__ gamma,
                                                                 const uint16_t* __restrict__ beta, int HW, int C2,
                                                                 int cg2, int G, uint32_t magic, float eps,
                                                                 int apply_silu, float* __restrict__ mean_rstd)
{
    constexpr int WAVES = THREADS / 64;
    __shared__ double s_red[2][WAVES];
    __shared__ f2 s_a[64], s_b[64];
    // XCD-aware order: workgroup b runs on XCD b % 8 and every XCD has its own L2.  The C/G-channel runs of
    // neighbouring groups share cache lines (20 bytes of a 640-byte pixel row at C = 320), so the groups of an image
    // are kept on ONE XCD -- dealt round-robin, each line was fetched by up to six L2s
    int bid = blockIdx.x;
    const int nwg = gridDim.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int n = bid / G, g = bid - n * G, tid = threadIdx.x;
    const int total = HW * cg2;
    // image n as a buffer: 32-bit byte offsets (one VGPR per access instead of a 64-bit address pair), and an offset
    // past the end reads 0 / drops the store, so the ragged tail needs no branch
    const uint32_t img_bytes = (uint32_t)HW * (uint32_t)C2 * 4u;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)n * HW * C2), 0, (int)img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (size_t)n * HW * C2), 0, (int)img_bytes, 0x00020000);
    const uint32_t goff = (uint32_t)g * (uint32_t)cg2 * 4u;
    uint32_t v[R];
    f2 s = f2{0.f, 0.f}, ss = f2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int d = tid + i * THREADS;
        const uint32_t p = __umulhi((uint32_t)d, magic);          // d / cg2 (exact for d < 2^32 / cg2)
        const uint32_t off = d < total ? (p * (uint32_t)C2 + ((uint32_t)d - p * (uint32_t)cg2)) * 4u + goff : 0xfffffff0u;
        v[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_x, off, 0, 0);
        if ((i & (R >= 64 ? 3 : 7)) == (R >= 64 ? 3 : 7)) __builtin_amdgcn_sched_barrier(0);   // offsets are made a few at a time, not R at a time
    }
#pragma unroll
    for (int i = 0; i < R; i++) {
        const f2 f = unpack2(v[i]);
        s += f;
        ss += f * f;
        if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    double a = (double)s.x + (double)s.y, b = (double)ss.x + (double)ss.y;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off, 64);
        b += __shfl_xor(b, off, 64);
    }
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = a; s_red[1][tid >> 6] = b; }
    __syncthreads();
    if (tid < cg2) {
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int w = 0; w < WAVES; w++) { sa += s_red[0][w]; sb += s_red[1][w]; }
        const double M = (double)total * 2.0;
        const double mean = sa / M;
        double var = sb / M - mean * mean;
        var = var < 0 ? 0 : var;
        const float meanf = (float)mean, rstd = rsqrtf((float)var + eps);
        if (tid == 0 && mean_rstd) {       // training: the backward pass (gn_group_fused_bwd_kernel) normalises with these
            mean_rstd[((size_t)n * G + g) * 2] = meanf;
            mean_rstd[((size_t)n * G + g) * 2 + 1] = rstd;
        }
        const int c = (g * cg2 + tid) * 2;
        const float a0 = rstd * bf2f(gamma[c]), a1 = rstd * bf2f(gamma[c + 1]);
        s_a[tid] = f2{a0, a1};
        s_b[tid] = f2{bf2f(beta[c]) - meanf * a0, bf2f(beta[c + 1]) - meanf * a1};
    }
    __syncthreads();
    int tid2 = tid;
    asm volatile("" : "+v"(tid2));       // the offsets are recomputed here, not kept alive across the reduction
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int d = tid2 + i * THREADS;
        const uint32_t p = __umulhi((uint32_t)d, magic);
        const uint32_t j = (uint32_t)d - p * (uint32_t)cg2;
        const uint32_t off = d < total ? (p * (uint32_t)C2 + j) * 4u + goff : 0xfffffff0u;
        f2 z = unpack2(v[i]) * s_a[j & 63] + s_b[j & 63];
        if (apply_silu) z = z * sigmoid2(z);
        __builtin_amdgcn_raw_buffer_store_b32(pack2(z), rs_y, off, 0, 0);
        if ((i & (R >= 64 ? 3 : 7)) == (R >= 64 ? 3 : 7)) __builtin_amdgcn_sched_barrier(0);
    }
}


// The backward pass of the same layers in ONE launch (the LoRA UNet's training pass, NeTF stage): the workgroup of
// (image, group) holds its slices of x AND dy in registers, forms the group's two sums
//   m1 = mean(gamma dz),  m2 = mean(gamma dz xhat),   dz = dy silu'(z),  z = gamma xhat + beta,  xhat = (x - mean) rstd
// (per-thread fp32, cross-thread fp64, like the forward pass) and writes dx = rstd (gamma dz - m1 - xhat m2) from the
// registers: x and dy read once (the two-pass form reads them twice and takes two launches plus a memset-free workspace),
// no atomics, bit-reproducible.  mean / rstd come from the forward launch.
template <int THREADS, int R>
__global__ __launch_bounds__(THREADS) void gn_group_fused_bwd_kernel(const uint32_t* __restrict__ x, const uint32_t* __restrict__ dy,
                                                                     uint32_t* __restrict__ dx, const uint16_t* __restrict__ gamma,
                                                                     const uint16_t* __restrict__ beta,
                                                                     const float* __restrict__ mean_rstd, int HW, int C2, int cg2,
                                                                     int G, uint32_t magic, int apply_silu)
{
    constexpr int WAVES = THREADS / 64;
    __shared__ double s_red[2][WAVES];
    __shared__ f2 s_g[64], s_b[64];
    __shared__ float s_m[2];
    int bid = blockIdx.x;
    const int nwg = gridDim.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);      // an image's groups on one XCD (see the forward kernel)
    const int n = bid / G, g = bid - n * G, tid = threadIdx.x;
    const int total = HW * cg2;
    const uint32_t img_bytes = (uint32_t)HW * (uint32_t)C2 * 4u;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)n * HW * C2), 0, (int)img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + (size_t)n * HW * C2), 0, (int)img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)(dx + (size_t)n * HW * C2), 0, (int)img_bytes, 0x00020000);
    const uint32_t goff = (uint32_t)g * (uint32_t)cg2 * 4u;
    const float mean = mean_rstd[((size_t)n * G + g) * 2], rstd = mean_rstd[((size_t)n * G + g) * 2 + 1];
    if (tid < cg2) {
        const int c = (g * cg2 + tid) * 2;
        s_g[tid] = f2{bf2f(gamma[c]), bf2f(gamma[c + 1])};
        s_b[tid] = f2{bf2f(beta[c]), bf2f(beta[c + 1])};
    }
    uint32_t vx[R], vd[R];
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int d = tid + i * THREADS;
        const uint32_t p = __umulhi((uint32_t)d, magic);          // d / cg2
        const uint32_t off = d < total ? (p * (uint32_t)C2 + ((uint32_t)d - p * (uint32_t)cg2)) * 4u + goff : 0xfffffff0u;
        vx[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_x, off, 0, 0);
        vd[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_d, off, 0, 0);       // past the end: dy = 0 -> the cell adds nothing
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    // gamma dz and xhat of register i
    auto cell = [&](int t0, int i, f2& t, f2& xh) {
        const int d = t0 + i * THREADS;
        const uint32_t p = __umulhi((uint32_t)d, magic);
        const uint32_t j = ((uint32_t)d - p * (uint32_t)cg2) & 63u;
        const f2 gm = s_g[j];
        xh = (unpack2(vx[i]) - mean) * rstd;
        f2 dz = unpack2(vd[i]);
        if (apply_silu) {
            const f2 z = xh * gm + s_b[j];
            const f2 sg = sigmoid2(z);
            dz *= sg * (1.f + z * (1.f - sg));
        }
        t = dz * gm;
    };
    f2 q1 = f2{0.f, 0.f}, q2 = f2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < R; i++) {
        f2 t, xh;
        cell(tid, i, t, xh);
        q1 += t;
        q2 += t * xh;
        if (R < 16 ? (i & 3) == 3 : (i & 1) == 1) __builtin_amdgcn_sched_barrier(0);   // few cells in flight: 2 R data registers are live
    }
    double a = (double)q1.x + (double)q1.y, b = (double)q2.x + (double)q2.y;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off, 64);
        b += __shfl_xor(b, off, 64);
    }
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = a; s_red[1][tid >> 6] = b; }
    __syncthreads();
    if (tid == 0) {
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int w = 0; w < WAVES; w++) { sa += s_red[0][w]; sb += s_red[1][w]; }
        const double M = (double)total * 2.0;
        s_m[0] = (float)(sa / M);
        s_m[1] = (float)(sb / M);
    }
    __syncthreads();
    const float m1 = s_m[0], m2 = s_m[1];
    int tid2 = tid;
    asm volatile("" : "+v"(tid2));       // the offsets are recomputed here, not kept alive across the reduction
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int d = tid2 + i * THREADS;
        const uint32_t p = __umulhi((uint32_t)d, magic);
        const uint32_t off = d < total ? (p * (uint32_t)C2 + ((uint32_t)d - p * (uint32_t)cg2)) * 4u + goff : 0xfffffff0u;
        f2 t, xh;
        cell(tid2, i, t, xh);      // channel index and gamma / beta are looked up again, not carried over in registers
        const f2 r = (t - m1 - xh * m2) * rstd;
        __builtin_amdgcn_raw_buffer_store_b32(pack2(r), rs_o, off, 0, 0);
        if (R < 16 ? (i & 3) == 3 : (i & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
}


// Same pass with an e4m3 result (one fp32 scale per tensor, value = scale * byte): the input of the fp8 convolution of
// the no-grad UNet forward (csrc/nn_fp8.hip) -- the quantisation costs no extra pass and halves the bytes written.
__global__ void gn_apply_fp8_kernel(const bf16x8* __restrict__ x, uint2* __restrict__ y,
                                    const uint16_t* __restrict__ gamma, const uint16_t* __restrict__ beta, int HW, int C,
                                    int G, int vpp, int rows, int ppb, int apply_silu,
                                    const float* __restrict__ mean_rstd, float inv_scale)
{
    const int n = blockIdx.y;
    const int tv = threadIdx.x % vpp, tr = threadIdx.x / vpp;
    const int p0 = blockIdx.x * ppb, p1 = min(HW, p0 + ppb);
    const int cg = C / G;
    f2 a[4], b[4];
    {
        int g[8];
        groups_of_vector(tv, cg, g);
        float a1[8], b1[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int c = tv * 8 + k;
            const float mean = mean_rstd[((size_t)n * G + g[k]) * 2];
            const float rstd = mean_rstd[((size_t)n * G + g[k]) * 2 + 1];
            a1[k] = rstd * bf2f(gamma[c]);
            b1[k] = bf2f(beta[c]) - mean * a1[k];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) { a[k] = f2{a1[2 * k], a1[2 * k + 1]}; b[k] = f2{b1[2 * k], b1[2 * k + 1]}; }
    }
    const bf16x8* xn = x + (size_t)n * HW * vpp;
    uint2* yn = y + (size_t)n * HW * vpp;
    auto body = [&](const bf16x8& v, int p) {
        const u32x4 w = __builtin_bit_cast(u32x4, v);
        float z[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            f2 t = unpack2(w.w[k]) * a[k] + b[k];
            if (apply_silu) t = t * sigmoid2(t);
            // the bf16 rounding of the unfused path first (same values as gn_apply_kernel), then scale + saturate
            const f2 zz = unpack2(pack2(t)) * inv_scale;
            z[2 * k] = zz.x < -448.f ? -448.f : (zz.x > 448.f ? 448.f : zz.x);   // saturate; a NaN stays a NaN
            z[2 * k + 1] = zz.y < -448.f ? -448.f : (zz.y > 448.f ? 448.f : zz.y);
        }
        int o0 = 0, o1 = 0;
        o0 = __builtin_amdgcn_cvt_pk_fp8_f32(z[0], z[1], o0, false);
        o0 = __builtin_amdgcn_cvt_pk_fp8_f32(z[2], z[3], o0, true);
        o1 = __builtin_amdgcn_cvt_pk_fp8_f32(z[4], z[5], o1, false);
        o1 = __builtin_amdgcn_cvt_pk_fp8_f32(z[6], z[7], o1, true);
        yn[(size_t)p * vpp + tv] = make_uint2((uint32_t)o0, (uint32_t)o1);
    };
    int p = p0 + tr;
    for (; p + (kUnroll - 1) * rows < p1; p += kUnroll * rows) {      // loads first: see gn_stats_kernel
        bf16x8 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) v[u] = xn[(size_t)(p + u * rows) * vpp + tv];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) body(v[u], p + u * rows);
    }
    for (; p < p1; p += rows) body(xn[(size_t)p * vpp + tv], p);
}

// dz = dy * silu'(z) (or dy); per group: s1 = sum gamma*dz, s2 = sum gamma*dz*xhat
__global__ void gn_bwd_stats_kernel(const bf16x8* __restrict__ x, const bf16x8* __restrict__ dy,
                                    const uint16_t* __restrict__ gamma, const uint16_t* __restrict__ beta,
                                    const float* __restrict__ mean_rstd, int HW, int C, int G, int vpp, int rows,
                                    int ppb, int apply_silu, double* __restrict__ ws, float* __restrict__ m12, int N,
                                    int n0)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int n = blockIdx.y + n0;
    const int tv = threadIdx.x % vpp, tr = threadIdx.x / vpp;
    const int p0 = blockIdx.x * ppb, p1 = min(HW, p0 + ppb);
    const int cg = C / G;
    f2 mean[4], rstd[4], gm[4], bt[4], q1[4], q2[4];
    {
        int g[8];
        groups_of_vector(tv, cg, g);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int c = tv * 8 + 2 * k;
            mean[k] = f2{mean_rstd[((size_t)n * G + g[2 * k]) * 2], mean_rstd[((size_t)n * G + g[2 * k + 1]) * 2]};
            rstd[k] = f2{mean_rstd[((size_t)n * G + g[2 * k]) * 2 + 1], mean_rstd[((size_t)n * G + g[2 * k + 1]) * 2 + 1]};
            gm[k] = f2{bf2f(gamma[c]), bf2f(gamma[c + 1])};
            bt[k] = f2{bf2f(beta[c]), bf2f(beta[c + 1])};
            q1[k] = q2[k] = f2{0.f, 0.f};
        }
    }
    const bf16x8* xn = x + (size_t)n * HW * vpp;
    const bf16x8* dn = dy + (size_t)n * HW * vpp;
    auto body = [&](const bf16x8& v, const bf16x8& d) {
        const u32x4 wv = __builtin_bit_cast(u32x4, v), wd = __builtin_bit_cast(u32x4, d);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const f2 xh = (unpack2(wv.w[k]) - mean[k]) * rstd[k];
            f2 dz = unpack2(wd.w[k]);
            if (apply_silu) {
                const f2 z = xh * gm[k] + bt[k];
                const f2 sg = sigmoid2(z);
                dz *= sg * (1.f + z * (1.f - sg));
            }
            const f2 t = dz * gm[k];
            q1[k] += t;
            q2[k] += t * xh;
        }
    };
    int p = p0 + tr;
    for (; p + rows < p1; p += 2 * rows) {      // four independent loads in flight per thread (see gn_stats_kernel)
        const size_t i0 = (size_t)p * vpp + tv, i1 = (size_t)(p + rows) * vpp + tv;
        const bf16x8 v0 = xn[i0], d0 = dn[i0], v1 = xn[i1], d1 = dn[i1];
        body(v0, d0);
        body(v1, d1);
    }
    for (; p < p1; p += rows) body(xn[(size_t)p * vpp + tv], dn[(size_t)p * vpp + tv]);
    float s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 4; k++) { s1[2 * k] = q1[k].x; s1[2 * k + 1] = q1[k].y; s2[2 * k] = q2[k].x; s2[2 * k + 1] = q2[k].y; }
    reduce_to_groups<1>(s1, s2, vpp, rows, tv, tr, C, G, N, n, (double)HW * cg, 0.f, ws, m12, lds);
}

__global__ void gn_bwd_apply_kernel(const bf16x8* __restrict__ x, const bf16x8* __restrict__ dy,
                                    const uint16_t* __restrict__ gamma, const uint16_t* __restrict__ beta,
                                    const float* __restrict__ mean_rstd, bf16x8* __restrict__ dx, int HW, int C,
                                    int G, int vpp, int rows, int ppb, int apply_silu, const float* __restrict__ m12,
                                    const bf16x8* __restrict__ add, int n0)
{
    const int n = blockIdx.y + n0;
    const int tv = threadIdx.x % vpp, tr = threadIdx.x / vpp;
    const int p0 = blockIdx.x * ppb, p1 = min(HW, p0 + ppb);
    const int cg = C / G;
    f2 mean[4], rstd[4], gm[4], bt[4], m1[4], m2[4];
    {
        int g[8];
        groups_of_vector(tv, cg, g);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int c = tv * 8 + 2 * k;
            const size_t i0 = ((size_t)n * G + g[2 * k]) * 2, i1 = ((size_t)n * G + g[2 * k + 1]) * 2;
            mean[k] = f2{mean_rstd[i0], mean_rstd[i1]};
            rstd[k] = f2{mean_rstd[i0 + 1], mean_rstd[i1 + 1]};
            gm[k] = f2{bf2f(gamma[c]), bf2f(gamma[c + 1])};
            bt[k] = f2{bf2f(beta[c]), bf2f(beta[c + 1])};
            m1[k] = f2{m12[i0], m12[i1]};
            m2[k] = f2{m12[i0 + 1], m12[i1 + 1]};
        }
    }
    const bf16x8* xn = x + (size_t)n * HW * vpp;
    const bf16x8* dn = dy + (size_t)n * HW * vpp;
    bf16x8* on = dx + (size_t)n * HW * vpp;
    const bf16x8* an = add ? add + (size_t)n * HW * vpp : nullptr;
    auto body = [&](const bf16x8& v, const bf16x8& d, const bf16x8& a, int p) {
        const u32x4 wv = __builtin_bit_cast(u32x4, v), wd = __builtin_bit_cast(u32x4, d), wa = __builtin_bit_cast(u32x4, a);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const f2 xh = (unpack2(wv.w[k]) - mean[k]) * rstd[k];
            f2 dz = unpack2(wd.w[k]);
            if (apply_silu) {
                const f2 z = xh * gm[k] + bt[k];
                const f2 sg = sigmoid2(z);
                dz *= sg * (1.f + z * (1.f - sg));
            }
            f2 r = rstd[k] * (dz * gm[k] - m1[k] - xh * m2[k]);
            if (an) r += unpack2(wa.w[k]);   // the other gradient arriving at this tensor (a ResnetBlock's skip path)
            o.w[k] = pack2(r);
        }
        on[(size_t)p * vpp + tv] = __builtin_bit_cast(bf16x8, o);
    };
    int p = p0 + tr;
    for (; p + rows < p1; p += 2 * rows) {      // four to six independent loads in flight per thread (see gn_stats_kernel)
        const size_t i0 = (size_t)p * vpp + tv, i1 = (size_t)(p + rows) * vpp + tv;
        const bf16x8 v0 = xn[i0], d0 = dn[i0], v1 = xn[i1], d1 = dn[i1];
        bf16x8 a0 = v0, a1 = v1;
        if (an) { a0 = an[i0]; a1 = an[i1]; }
        body(v0, d0, a0, p);
        body(v1, d1, a1, p + rows);
    }
    for (; p < p1; p += rows) {
        const size_t i0 = (size_t)p * vpp + tv;
        const bf16x8 v0 = xn[i0], d0 = dn[i0];
        bf16x8 a0 = v0;
        if (an) a0 = an[i0];
        body(v0, d0, a0, p);
    }
}

struct Geo {
    int vpp, rows, threads, ppb, nchunks;
    int ppb_stats, nchunks_stats;   // statistics passes: fewer, larger chunks (see make_geo)
    size_t lds;
};

bool make_geo(int HW, int C, int G, int N, Geo* g)
{
    if (C <= 0 || G <= 0 || C % 8 || C % G || HW <= 0 || N <= 0) return false;
    g->vpp = C / 8;
    if (g->vpp > kMaxThreads) return false;
    g->rows = g->vpp >= 256 ? 1 : 256 / g->vpp;
    g->threads = g->rows * g->vpp;
    // enough workgroups to fill 256 CUs a few times over, but >= rows pixels each
    // 128 pixels per workgroup: workgroups are dispatched in order, so the set in flight sweeps a compact window of the
    // tensor (measured on 537 MB: 1024 pixels -> 128: apply 264 -> 234 us, backward + add 684 -> 637 us; below 64 the
    // per-workgroup prologue dominates)
    int ppb = 128;
    while (ppb > g->rows && (long)((HW + ppb - 1) / ppb) * N < 2048) ppb >>= 1;
    if (ppb < g->rows) ppb = g->rows;
    g->ppb = ppb;
    g->nchunks = (HW + ppb - 1) / ppb;
    // Every statistics workgroup ends with 2*G fp64 atomics onto the SAME 2*G addresses of its image;
    // with hundreds of chunks per image those serialise in L2 (35-75 us per call measured at batch 2).
    // Cap the chunks per image (64, or 256 for the big VAE tensors that need the parallelism).
    const int cap = HW >= 65536 ? 256 : 64;
    int ppbs = (HW + cap - 1) / cap;
    if (ppbs < ppb) ppbs = ppb;
    g->ppb_stats = ppbs;
    g->nchunks_stats = (HW + ppbs - 1) / ppbs;
    g->lds = (size_t)2 * g->rows * C * sizeof(float);
    return true;
}

// mean / rstd from the per-tile partial sums a convolution's epilogue left (nn_conv3x3.hip, stat_part):
// part[n][C/4][rows] float2 {sum, sum of squares} per 4-channel quad.  One workgroup per (image, group) adds its
// cg/4 quads x rows partials in fp64 -- the order is fixed, so the result does not vary from run to run.
__global__ __launch_bounds__(256) void gn_finish_partials_kernel(const float2* __restrict__ part, int rows, int C, int G,
                                                                  double M, float eps, float* __restrict__ result)
{
    const int n = blockIdx.y, g = blockIdx.x;
    const int qpg = (C / G) >> 2;
    const float2* p = part + ((size_t)n * (C >> 2) + (size_t)g * qpg) * rows;   // the group's quads are contiguous
    const int total = qpg * rows;
    double sa = 0.0, sb = 0.0;
    for (int i = threadIdx.x; i < total; i += 256) {
        const float2 v = p[i];
        sa += (double)v.x;
        sb += (double)v.y;
    }
    __shared__ double s_a[256], s_b[256];
    s_a[threadIdx.x] = sa;
    s_b[threadIdx.x] = sb;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s_a[threadIdx.x] += s_a[threadIdx.x + w];
            s_b[threadIdx.x] += s_b[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mean = s_a[0] / M;
        double var = s_b[0] / M - mean * mean;
        var = var < 0 ? 0 : var;
        result[2 * ((size_t)n * G + g)] = (float)mean;
        result[2 * ((size_t)n * G + g) + 1] = rsqrtf((float)var + eps);
    }
}

int fail(int code, const char* msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

}  // namespace

extern "C" {

size_t gd_nn_groupnorm_ws_bytes(int N, int G)
{
    return (size_t)kSlots * N * G * 2 * sizeof(double) + (size_t)N * sizeof(unsigned long long);
}

int gd_nn_groupnorm_silu_forward(void* stream, const void* x, void* y, const void* gamma, const void* beta, int N,
                                 int HW, int C, int G, float eps, int apply_silu, double* stats_ws, float* mean_rstd)
{
    Geo g;
    if (!x || !y || !gamma || !beta || !mean_rstd) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (!make_geo(HW, C, G, N, &g)) return fail(GD_NN_ERR_INVALID_ARG, "need C % 8 == 0, C % G == 0, C <= 2560");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(g.nchunks, N), block(g.threads), grid_s(g.nchunks_stats, N);
    if (stats_ws)   // NULL: mean_rstd is an INPUT (gd_nn_groupnorm_finish_partials, or a previous statistics pass)
        hipLaunchKernelGGL(gn_stats_kernel, grid_s, block, g.lds, s, (const bf16x8*)x, HW, C, G, g.vpp, g.rows,
                           g.ppb_stats, eps, stats_ws, mean_rstd, N, 0);
    hipLaunchKernelGGL(gn_apply_kernel, grid, block, 0, s, (const bf16x8*)x, (bf16x8*)y, (const uint16_t*)gamma,
                       (const uint16_t*)beta, HW, C, G, g.vpp, g.rows, g.ppb, apply_silu, mean_rstd);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

int gd_nn_groupnorm_silu_fused_supported(int N, int HW, int C, int G)
{
    if (N <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G || (C / G) % 2 || C / G < 4 || C / G > 128) return 0;
    // 10-channel runs (C = 320) on 64x64 maps: 20 four-byte loads per thread from 20-byte runs -- level with the two-pass
    // kernels at 16 latents, 0.84x at 2 (tools/gn_small_bench.py); every other shape of the UNet is 1.2-5x faster
    const long total = (long)HW * (C / G / 2);
    if (total > 1024L * 16 && C / G < 16) return 0;
    return total <= 1024L * 32 ? 1 : 0;
}

int gd_nn_groupnorm_silu_fused_forward(void* stream, const void* x, void* y, const void* gamma, const void* beta, int N,
                                       int HW, int C, int G, float eps, int apply_silu)
{
    return gd_nn_groupnorm_silu_fused_forward_stats(stream, x, y, gamma, beta, N, HW, C, G, eps, apply_silu, nullptr);
}

int gd_nn_groupnorm_silu_fused_backward_supported(int N, int HW, int C, int G)
{
    // x AND dy live in registers: half the forward pass's slice (32 dwords of each per thread spill 144 registers)
    return gd_nn_groupnorm_silu_fused_supported(N, HW, C, G) && (long)HW * (C / G / 2) <= 1024L * 16 ? 1 : 0;
}

int gd_nn_groupnorm_silu_fused_backward(void* stream, const void* x, const void* dy, const void* gamma, const void* beta,
                                        const float* mean_rstd, void* dx, int N, int HW, int C, int G, int apply_silu)
{
    if (!x || !dy || !gamma || !beta || !mean_rstd || !dx) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (!gd_nn_groupnorm_silu_fused_backward_supported(N, HW, C, G))
        return fail(GD_NN_ERR_INVALID_ARG, "fused GroupNorm backward: need C % G == 0, C / G even, 4 <= C / G <= 128, HW * C / G <= 32768");
    const int cg2 = C / G / 2, total = HW * cg2;
    const uint32_t magic = (uint32_t)((0x100000000ull + (uint64_t)cg2 - 1) / (uint64_t)cg2);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)G * (unsigned)N);
#define GD_GN_FUSED_BWD(T_, R_)                                                                                         \
    hipLaunchKernelGGL((gn_group_fused_bwd_kernel<T_, R_>), grid, dim3(T_), 0, s, (const uint32_t*)x, (const uint32_t*)dy, \
                       (uint32_t*)dx, (const uint16_t*)gamma, (const uint16_t*)beta, mean_rstd, HW, C / 2, cg2, G, magic, \
                       apply_silu)
    if (total <= 256 * 8) GD_GN_FUSED_BWD(256, 8);
    else if (total <= 1024 * 8) GD_GN_FUSED_BWD(1024, 8);
    else GD_GN_FUSED_BWD(1024, 16);
#undef GD_GN_FUSED_BWD
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

int gd_nn_groupnorm_silu_fused_forward_stats(void* stream, const void* x, void* y, const void* gamma, const void* beta, int N,
                                             int HW, int C, int G, float eps, int apply_silu, float* mean_rstd)
{
    if (!x || !y || !gamma || !beta) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (!gd_nn_groupnorm_silu_fused_supported(N, HW, C, G))
        return fail(GD_NN_ERR_INVALID_ARG, "fused GroupNorm: need C % G == 0, C / G even, 4 <= C / G <= 128, HW * C / G <= 65536");
    const int cg2 = C / G / 2, total = HW * cg2;
    const uint32_t magic = (uint32_t)((0x100000000ull + (uint64_t)cg2 - 1) / (uint64_t)cg2);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)G * (unsigned)N);
#define GD_GN_FUSED(T_, R_)                                                                                             \
    hipLaunchKernelGGL((gn_group_fused_kernel<T_, R_>), grid, dim3(T_), 0, s, (const uint32_t*)x, (uint32_t*)y,          \
                       (const uint16_t*)gamma, (const uint16_t*)beta, HW, C / 2, cg2, G, magic, eps, apply_silu, mean_rstd)
    if (total <= 256 * 8) GD_GN_FUSED(256, 8);
    else if (total <= 1024 * 8) GD_GN_FUSED(1024, 8);
    else if (total <= 1024 * 16) GD_GN_FUSED(1024, 16);
    else GD_GN_FUSED(1024, 32);      // (1024 x 64 = the 640 / 960-channel 64x64 maps would need 24 spilled registers)
#undef GD_GN_FUSED
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

int gd_nn_groupnorm_silu_forward_fp8(void* stream, const void* x, void* y_fp8, const void* gamma, const void* beta, int N,
                                     int HW, int C, int G, float eps, int apply_silu, double* stats_ws, float* mean_rstd,
                                     float inv_scale)
{
    Geo g;
    if (!x || !y_fp8 || !gamma || !beta || !stats_ws || !mean_rstd) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (!make_geo(HW, C, G, N, &g)) return fail(GD_NN_ERR_INVALID_ARG, "need C % 8 == 0, C % G == 0, C <= 2560");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(g.nchunks, N), block(g.threads), grid_s(g.nchunks_stats, N);
    hipLaunchKernelGGL(gn_stats_kernel, grid_s, block, g.lds, s, (const bf16x8*)x, HW, C, G, g.vpp, g.rows,
                       g.ppb_stats, eps, stats_ws, mean_rstd, N, 0);
    hipLaunchKernelGGL(gn_apply_fp8_kernel, grid, block, 0, s, (const bf16x8*)x, (uint2*)y_fp8, (const uint16_t*)gamma,
                       (const uint16_t*)beta, HW, C, G, g.vpp, g.rows, g.ppb, apply_silu, mean_rstd, inv_scale);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

int gd_nn_groupnorm_stats(void* stream, const void* x, int N, int HW, int C, int G, float eps, double* stats_ws,
                          float* mean_rstd)
{
    Geo g;
    if (!x || !stats_ws || !mean_rstd) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (!make_geo(HW, C, G, N, &g)) return fail(GD_NN_ERR_INVALID_ARG, "need C % 8 == 0, C % G == 0, C <= 2560");
    hipStream_t s = (hipStream_t)stream;
    dim3 block(g.threads), grid_s(g.nchunks_stats, N);
    hipLaunchKernelGGL(gn_stats_kernel, grid_s, block, g.lds, s, (const bf16x8*)x, HW, C, G, g.vpp, g.rows,
                       g.ppb_stats, eps, stats_ws, mean_rstd, N, 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

int gd_nn_groupnorm_finish_partials(void* stream, const float* stat_part, int N, size_t rows, int C, int G, int HW,
                                    float eps, float* mean_rstd)
{
    if (!stat_part || !mean_rstd) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (N <= 0 || rows == 0 || rows > (1u << 24) || G <= 0 || C % G || (C / G) % 4 || HW <= 0)
        return fail(GD_NN_ERR_INVALID_ARG, "finish_partials: need C % G == 0 and (C / G) % 4 == 0");
    hipLaunchKernelGGL(gn_finish_partials_kernel, dim3(G, N), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)stat_part, (int)rows, C, G, (double)HW * (double)(C / G), eps, mean_rstd);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

int gd_nn_groupnorm_silu_backward(void* stream, const void* x, const void* dy, const void* gamma, const void* beta,
                                  const float* mean_rstd, void* dx, int N, int HW, int C, int G, int apply_silu,
                                  double* stats_ws, float* group_sums, const void* add)
{
    Geo g;
    if (!x || !dy || !gamma || !beta || !stats_ws || !mean_rstd || !dx || !group_sums)
        return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (!make_geo(HW, C, G, N, &g)) return fail(GD_NN_ERR_INVALID_ARG, "need C % 8 == 0, C % G == 0, C <= 2560");
    hipStream_t s = (hipStream_t)stream;
    float* m12 = group_sums;
    // (Tried: image chunk by image chunk, statistics then apply, so that the apply pass re-reads x and dy from the
    // 256 MiB Infinity Cache.  +2.5 ... +3.5 ms per 8-view step at 96 / 160 MB chunks: the smaller launches lose more
    // to their tails than the on-die re-read gains.  The kernels keep their image-offset argument.)
    dim3 grid(g.nchunks, N), block(g.threads), grid_s(g.nchunks_stats, N);
    hipLaunchKernelGGL(gn_bwd_stats_kernel, grid_s, block, g.lds, s, (const bf16x8*)x, (const bf16x8*)dy,
                       (const uint16_t*)gamma, (const uint16_t*)beta, mean_rstd, HW, C, G, g.vpp, g.rows,
                       g.ppb_stats, apply_silu, stats_ws, m12, N, 0);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, grid, block, 0, s, (const bf16x8*)x, (const bf16x8*)dy,
                       (const uint16_t*)gamma, (const uint16_t*)beta, mean_rstd, (bf16x8*)dx, HW, C, G, g.vpp, g.rows,
                       g.ppb, apply_silu, m12, (const bf16x8*)add, 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

const char* gd_nn_last_error(void) { return g_err; }

}  // extern "C"
