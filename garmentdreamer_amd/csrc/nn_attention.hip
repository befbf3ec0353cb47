// nn_attention.hip -- fused self-attention forward for head_dim 64 (bf16, no mask, inference): the UNet's
// spatial self-attention (diffusers Attention -> F.scaled_dot_product_attention; reached through
// Garment_3DGS/threestudio/models/guidance/stable_diffusion_guidance.py:153-157).  S = 4096 / 1024 / 256 keys.
//
// CDNA4 mapping.  Workgroup = 4 wave64s x 32 query rows; K and V tiles of 64 keys stream through LDS by LDS-DMA
// (double buffered, source-side XOR swizzle -> conflict-free ds_read_b128 fragments).  Scores are computed
// TRANSPOSED, S^T = K Q^T (v_mfma_f32_32x32x16_bf16 with K as the A operand), so a lane holds 32 scores of ONE
// query: the row maximum / sum are in-lane loops plus a single cross-half exchange, and exp2 / rescaling never
// leave the lane.  The same registers, packed to bf16, ARE the B operand of O^T += V^T P^T.  The accumulator layout
// gives lane half fh the score rows {0-3, 8-11} + 4 fh of every 16: round 3 reads the K fragment's row
// pi(i) = {0-3, 8-11, 4-7, 12-15}[i] instead of i (a different LDS address, nothing else), so those registers hold keys
// 8 fh .. 8 fh + 7 in order and V^T is consumed in NATURAL key order -- it can come straight from a GEMM
// (W_v x^T, gd_nn_attention_d64_forward_vt) instead of a transposing pre-pass.  No shuffles, no LDS round trip for P.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/gd_nn.h"

namespace {

thread_local char g_err[256] = "";
#ifndef GD_ATTN_ABLATE
#define GD_ATTN_ABLATE 0   // timing builds only (tools/attn_ablate.sh): 1 no exp2, 2 no LDS-DMA after the prologue, 3 no MFMA, 4 no LDS fragment reads, 5 no running-maximum pass, 6 no per-tile barrier, 9 / 10 orders of the loop head
#endif
int g_attn_waves = 0;   // GD_NN_ATTN_WAVES = 4 / 8 forces the workgroup size (tuning)
int g_attn_xcd = 1;     // GD_NN_ATTN_XCD=0: query tiles dealt round-robin over the XCDs (A/B)
int fail(int code, const char* msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi)
{
    f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// byte offset of logical (row, 16-B chunk j) inside a swizzled [rows][64] bf16 tile image (256-byte lines)
__device__ __forceinline__ int swz(int row, int j)
{
    return (row >> 1) * 256 + (((((row & 1) << 3) | j) ^ ((row >> 1) & 15)) << 4);
}

__device__ __forceinline__ void bload_lds16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff, char* lds_wave_base)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff,
                                             0, 0);
}

constexpr int kTk = 64, kD = 64;
constexpr int kTile = kTk * kD * 2;   // 8 KB

// V [B][Skv][H*64] (row stride v_rs elements)  ->  Vt [B][H][64][Skv], keys in natural order, padded keys zero.
// 64 keys x 64 d per workgroup through LDS.  (Only for callers that hold V row-major: the cross-attention's K / V.)
__global__ __launch_bounds__(256) void attn_vt_kernel(const uint16_t* __restrict__ v, uint16_t* __restrict__ vt, int Skv,
                                                      int H, int64_t v_bs, int v_rs, int kv_len)
{
    __shared__ uint16_t tile[64][66];
    const int b = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * 64;
    const uint16_t* src = v + b * v_bs + (int64_t)k0 * v_rs + h * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int key = i >> 6, d = i & 63;
        tile[key][d] = k0 + key < kv_len ? src[(int64_t)key * v_rs + d] : (uint16_t)0;   // padded keys: V = 0
    }
    __syncthreads();
    uint16_t* dst = vt + (((int64_t)b * H + h) * 64) * Skv + k0;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int d = i >> 6, pos = i & 63;
        dst[(int64_t)d * Skv + pos] = tile[pos][d];
    }
}

// The backward pass's three transposes (K^T, Q^T, dO^T) in ONE launch: blockIdx.z = 3 b + which.
struct VtJob { const uint16_t* src; uint16_t* dst; int S; int64_t bs; int rs; int len; };
__global__ __launch_bounds__(256) void attn_vt3_kernel(VtJob j0, VtJob j1, VtJob j2, int H)
{
    __shared__ uint16_t tile[64][66];
    const int which = blockIdx.z % 3, b = blockIdx.z / 3;
    const VtJob j = which == 0 ? j0 : (which == 1 ? j1 : j2);
    const int h = blockIdx.y, k0 = blockIdx.x * 64;
    if (k0 >= j.S) return;                                  // (workgroup-uniform: the grid is sized for the longest tensor)
    const uint16_t* src = j.src + b * j.bs + (int64_t)k0 * j.rs + h * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int key = i >> 6, d = i & 63;
        tile[key][d] = k0 + key < j.len ? src[(int64_t)key * j.rs + d] : (uint16_t)0;
    }
    __syncthreads();
    uint16_t* dst = j.dst + (((int64_t)b * H + h) * 64) * j.S + k0;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int d = i >> 6, pos = i & 63;
        dst[(int64_t)d * j.S + pos] = tile[pos][d];
    }
}

// WAVES = 4 or 8 wave64s per workgroup (32 query rows each).  Eight waves share every K / V^T tile: half the LDS-DMA
// issue and LDS fill per query; four waves give twice the workgroups when the grid is small (one view per GPU).
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void attn_fwd_d64_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                           const uint16_t* __restrict__ vt, uint16_t* __restrict__ o, int S,
                                                           int Skv, int H, int64_t q_bs, int q_rs, int64_t k_bs, int k_rs,
                                                           int64_t o_bs, int o_rs, float c /* scale * log2(e) */, int kv_len,
                                                           float* __restrict__ lse /* [B][H][S] natural-log sum-exp of the scaled scores, or NULL */,
                                                           int64_t vt_bs /* elements between the V^T images of two batch entries */,
                                                           int xcd_remap)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;                 // 3 stages
    char* sV = smem + 3 * kTile;     // 3 stages
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware order (round 5): workgroups are dealt to the eight XCDs round-robin by their linear index, and every workgroup of a
    // (batch, head) streams that head's whole K / V^T (1 MB at 4096 keys): with the query tiles of a head spread over all XCDs each
    // private L2 fetched every head.  The linear index is re-dealt so that consecutive query tiles -- one head -- share an XCD.
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const unsigned gx = gridDim.x, total = gx * gridDim.y;
        if (xcd_remap && (total & 7u) == 0) {
            const unsigned id = blockIdx.x + gx * blockIdx.y;
            const unsigned id2 = (id & 7u) * (total >> 3) + (id >> 3);
            by = (int)(id2 / gx);
            bx = (int)(id2 - (unsigned)by * gx);
        }
    }
    const int b = by / H, h = by - b * H;
    const int fh = lane >> 5, fn = lane & 31;
    constexpr int THREADS = 64 * WAVES, NP = 512 / THREADS;   // 16-byte pieces of a 64 x 64 tile per thread
    const int qrow = bx * (32 * WAVES) + wave * 32 + fn;
    const int qld = qrow < S ? qrow : S - 1;

    // Q fragments (B operand of S^T = K Q^T): lane (query fn, half fh) holds d = 16 kk + 8 fh .. + 7
    bf16x8_t qf[4];
    {
        const uint16_t* qp = q + b * q_bs + (int64_t)qld * q_rs + h * kD + 8 * fh;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) qf[kk] = *(const bf16x8_t*)(qp + 16 * kk);
    }
    // K rows: [key][64 d], row stride k_rs elements; Vt rows: [d][Skv]
    const uint32_t k_row_bytes = (uint32_t)k_rs * 2u, v_row_bytes = (uint32_t)Skv * 2u;
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(k + b * k_bs + h * kD), 0, (int)((uint32_t)kv_len * k_row_bytes), 0x00020000);   // rows >= kv_len read as 0
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(vt + b * vt_bs + ((int64_t)h * kD) * Skv), 0, (int)((uint32_t)kD * v_row_bytes), 0x00020000);
    uint32_t k_off[NP], v_off[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const int cidx = tid + THREADS * i;
        const int line = cidx >> 4, cc = (cidx & 15) ^ (line & 15);
        const int r = 2 * line + (cc >> 3);
        k_off[i] = (uint32_t)r * k_row_bytes + (uint32_t)(cc & 7) * 16u;   // + tile * 64 rows (soffset)
        v_off[i] = (uint32_t)r * v_row_bytes + (uint32_t)(cc & 7) * 16u;   // + tile * 128 bytes (soffset)
    }
    auto issue = [&](int buf, int t) {
#pragma unroll
        for (int i = 0; i < NP; i++) {
            bload_lds16(rs_k, k_off[i], (uint32_t)t * kTk * k_row_bytes, sK + buf * kTile + (wave * 64 + THREADS * i) * 16);
            bload_lds16(rs_v, v_off[i], (uint32_t)t * (kTk * 2), sV + buf * kTile + (wave * 64 + THREADS * i) * 16);
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; r++) o0[r] = o1[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float grow_thr = 8.0f / c;     // 2^8 of head-room in exp2 units, expressed in raw score units
    const int ntiles = Skv / kTk;
    uint32_t krd[2], vrd[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        // K fragment of key block j, kk = 0 (kk: slot ^ 2kk); accumulator row fn <- key pi(fn) (see the header)
        const int p16 = fn & 15, key16 = p16 < 4 ? p16 : (p16 < 8 ? p16 + 4 : (p16 < 12 ? p16 - 4 : p16));
        krd[j] = (uint32_t)swz(32 * j + (fn & 16) + key16, fh);
        vrd[j] = (uint32_t)swz(32 * j + fn, fh);       // V^T fragment of d block j, 16-key group 0
    }
    // S^T tile of 64 keys x 32 queries (two 32x32 accumulators) from the K stage at `pk`
    auto qk = [&](const char* pk, f32x16& s0, f32x16& s1, int tile) {
#pragma unroll
        for (int r = 0; r < 16; r++) s0[r] = s1[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
#if GD_ATTN_ABLATE == 4
            const bf16x8_t k0 = qf[kk ^ 1], k1 = qf[kk ^ 2];
#else
            const bf16x8_t k0 = *(const bf16x8_t*)(pk + (krd[0] ^ (uint32_t)(kk << 5)));
            const bf16x8_t k1 = *(const bf16x8_t*)(pk + (krd[1] ^ (uint32_t)(kk << 5)));
#endif
#if GD_ATTN_ABLATE == 3
            s0[kk] += __builtin_bit_cast(float, (int)k0[0]); s1[kk] += __builtin_bit_cast(float, (int)k1[1]);
#else
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[kk], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[kk], s1, 0, 0, 0);
#endif
        }
        if ((tile + 1) * kTk > kv_len) {      // last tile of a key count that is not a multiple of 64 (cross-attention:
#pragma unroll                                   // 77 text tokens): padded keys get no weight
            for (int r = 0; r < 16; r++) {
                // accumulator row (r & 3) + 8 (r >> 2) + 4 fh holds key pi(row) = 16 (r >> 3) + 8 fh + 4 ((r >> 2) & 1) + (r & 3)
                const int key = tile * kTk + 16 * (r >> 3) + 8 * fh + 4 * ((r >> 2) & 1) + (r & 3);
                if (key >= kv_len) s0[r] = -INFINITY;
                if (key + 32 >= kv_len) s1[r] = -INFINITY;
            }
        }
    };
    // Part 1 of the online softmax of the lane's 64 scores: the (deferred) running maximum.
    auto update_max = [&](const f32x16& s0, const f32x16& s1) {
        // (a tree of v_max3_f32 instead of this chain measured the same, 457 us either way: tools/attn_ablate.sh)
        float mx = s0[0];
#pragma unroll
        for (int r = 1; r < 16; r++) mx = fmaxf(mx, s0[r]);
#pragma unroll
        for (int r = 0; r < 16; r++) mx = fmaxf(mx, s1[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // Deferred maximum: the running reference m_run only moves (and O, l are only rescaled) when some query's
        // maximum grew by more than 2^8 in exp2 units; otherwise P = exp2((s - m_run) c) <= 256 stays well inside
        // fp32 / bf16 range and the final division by l normalises it.  The maximum settles after the first
        // tiles, so the 32-register rescale of O leaves the steady-state loop.
        if (__builtin_amdgcn_ballot_w64(mx > m_run + grow_thr) != 0) {      // wave-uniform, rare
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                o0[r] *= alpha;
                o1[r] *= alpha;
            }
        }
    };
    // Part 2: P = exp2((s - m) c) in place, row sum, O^T += V^T P^T from the V stage at `pv`.
    auto exp_pv = [&](f32x16& s0, f32x16& s1, const char* pv) {
        // two scores per v_pk_fma_f32 / v_pk_add_f32: a packed fp32 instruction holds the SIMD ~5.5 cycles against
        // 2 x 4.5 for the scalar pair (tools/probes/valu_rate_probe.hip), and this loop is VALU-bound
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 c2 = {c, c}, nmc = {-m_run * c, -m_run * c};
        f2 psum = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const f2 a = f2{s0[r], s0[r + 1]} * c2 + nmc, b = f2{s1[r], s1[r + 1]} * c2 + nmc;
#if GD_ATTN_ABLATE == 1
            s0[r] = a.x; s0[r + 1] = a.y; s1[r] = b.x; s1[r + 1] = b.y;
#else
            s0[r] = __builtin_amdgcn_exp2f(a.x);
            s0[r + 1] = __builtin_amdgcn_exp2f(a.y);
            s1[r] = __builtin_amdgcn_exp2f(b.x);
            s1[r + 1] = __builtin_amdgcn_exp2f(b.y);
#endif
            psum += f2{s0[r], s0[r + 1]} + f2{s1[r], s1[r + 1]};
        }
        l_run += psum.x + psum.y;
        // P (bf16) as the B operand: regs 8u..8u+7 of score block j  <->  16-key group 2j+u
        bf16x8_t pb[4];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            uint32_t w0[4], w1[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                w0[i] = pack_bf16(s0[8 * u + 2 * i], s0[8 * u + 2 * i + 1]);
                w1[i] = pack_bf16(s1[8 * u + 2 * i], s1[8 * u + 2 * i + 1]);
            }
            pb[u] = __builtin_bit_cast(bf16x8_t, make_uint4(w0[0], w0[1], w0[2], w0[3]));
            pb[2 + u] = __builtin_bit_cast(bf16x8_t, make_uint4(w1[0], w1[1], w1[2], w1[3]));
        }
#pragma unroll
        for (int g16 = 0; g16 < 4; g16++) {      // 16-key group g16 = 2j + u -> chunks 2 g16 + fh
#if GD_ATTN_ABLATE == 4
            const bf16x8_t v0 = pb[g16 ^ 1], v1 = pb[g16 ^ 2];
#else
            const bf16x8_t v0 = *(const bf16x8_t*)(pv + (vrd[0] ^ (uint32_t)(g16 << 5)));
            const bf16x8_t v1 = *(const bf16x8_t*)(pv + (vrd[1] ^ (uint32_t)(g16 << 5)));
#endif
#if GD_ATTN_ABLATE == 3
            o0[g16] += __builtin_bit_cast(float, (int)v0[0]) + __builtin_bit_cast(float, (int)pb[g16][0]);
            o1[g16] += __builtin_bit_cast(float, (int)v1[1]) + __builtin_bit_cast(float, (int)pb[g16][1]);
#else
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pb[g16], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pb[g16], o1, 0, 0, 0);
#endif
        }
    };

    // Software pipeline over the KV tiles, three LDS stages: after the (rarely taken) rescale branch, ONE basic
    // block holds the score MFMAs of tile t+1, the exp2 / sum / pack VALU work of tile t and its P V MFMAs, so the
    // matrix pipe runs under the transcendental work; the LDS-DMA of tile t+2 is in flight meanwhile.  The loop is
    // unrolled by two so the score registers ping-pong without copies.
    issue(0, 0);
    if (ntiles > 1) issue(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 sa0, sa1, sb0, sb1;
    qk(sK, sa0, sa1, 0);
    auto iteration = [&](int t, f32x16& c0, f32x16& c1, f32x16& n0, f32x16& n1) {
        if (GD_ATTN_ABLATE != 6) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // tile t+1 (issued one iteration ago) has landed
        __syncthreads();                                      // ... for every wave; stage (t+2)%3 is free again
        }
#if GD_ATTN_ABLATE == 9      // the round-4 order (same-box A/B): DMA issue, running maximum, THEN the next tile's score MFMAs
        if (t + 2 < ntiles) issue((t + 2) % 3, t + 2);
        update_max(c0, c1);
        qk(sK + ((t + 1) % 3) * kTile, n0, n1, t + 1);
#elif GD_ATTN_ABLATE == 10   // experiment: the DMA issue behind the score MFMAs too
        qk(sK + ((t + 1) % 3) * kTile, n0, n1, t + 1);
        if (t + 2 < ntiles) issue((t + 2) % 3, t + 2);
        update_max(c0, c1);
#else
        // the next tile's score MFMAs go FIRST: they do not depend on the running maximum, and issued in front of it they are in
        // flight under its 31 dependent v_max, the cross-half shuffle and the (rarely taken) rescale branch -- the matrix pipe
        // idled through that block before (450 -> 432 us at 16 x 5 heads x 4096^2, profiles/r05_attn_ablation.txt)
        if (GD_ATTN_ABLATE != 2 && t + 2 < ntiles) issue((t + 2) % 3, t + 2);
        qk(sK + ((t + 1) % 3) * kTile, n0, n1, t + 1);
        if (GD_ATTN_ABLATE != 5) update_max(c0, c1);
#endif
        exp_pv(c0, c1, sV + (t % 3) * kTile);
    };
    int t = 0;
    for (; t + 2 < ntiles; t += 2) {
        iteration(t, sa0, sa1, sb0, sb1);
        iteration(t + 1, sb0, sb1, sa0, sa1);
    }
    if (t + 1 < ntiles) {            // one pipelined iteration left (even tile count)
        iteration(t, sa0, sa1, sb0, sb1);
        update_max(sb0, sb1);
        exp_pv(sb0, sb1, sV + ((ntiles - 1) % 3) * kTile);
    } else {
        update_max(sa0, sa1);
        exp_pv(sa0, sa1, sV + ((ntiles - 1) % 3) * kTile);
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    // what a flash-attention backward pass recomputes P from: ln sum_k exp(scale * s_k) = m * scale + ln l (m is the deferred
    // running reference, not necessarily the maximum: l is taken relative to it, so the sum is exact either way)
    if (lse && fh == 0 && qrow < S) lse[((size_t)b * H + h) * S + qrow] = m_run * c * 0.69314718055994531f + __logf(l_tot);
    if (qrow < S) {
        uint16_t* op = o + b * o_bs + (int64_t)qrow * o_rs + h * kD;
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {         // rows d = (r & 3) + 8 (r >> 2) + 4 fh of each 32-d block
            uint2 a, bq;
            a.x = pack_bf16(o0[4 * g4] * inv, o0[4 * g4 + 1] * inv);
            a.y = pack_bf16(o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv);
            bq.x = pack_bf16(o1[4 * g4] * inv, o1[4 * g4 + 1] * inv);
            bq.y = pack_bf16(o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv);
            *(uint2*)(op + 8 * g4 + 4 * fh) = a;
            *(uint2*)(op + 32 + 8 * g4 + 4 * fh) = bq;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward pass (training pass of the NeTF stage's LoRA UNet), built from the forward kernel's pieces: transposed score
// tiles with one "row-side" element per lane, P / dS packed to bf16 in place as the B operand of the next product.
//   MODE 0 (dQ):      lanes own QUERIES, 64-key tiles stream.   S^T = K Q^T,  dP^T = V dO^T,  P^T = exp2(S^T c - L[q]),
//                     dS^T = P^T (dP^T - D[q]),  dQ^T += K^T dS^T            (L, D: one value per lane)
//   MODE 1 (dK, dV):  lanes own KEYS, 64-query tiles stream.    S = Q K^T,  dP = dO V^T,  P = exp2(S c - L[q]),
//                     dS = P (dP - D[q]),  dV^T += dO^T P,  dK^T += Q^T dS   (L, D: per streamed row, 16 float4 per tile)
// L = lse log2(e) with the forward pass's log-sum-exp (no running maximum here), D = rowsum(dO o O) (attn_dsum_kernel).
// The transposed streamed operands (K^T; dO^T, Q^T) come from attn_vt_kernel, like V^T in the forward pass.  Two LDS
// stages, loads of tile t + 1 in flight during tile t.  P and dS are rounded to bf16 where they enter an MFMA, as in every
// flash backward.
__global__ __launch_bounds__(256) void attn_dsum_kernel(const uint16_t* __restrict__ o, const uint16_t* __restrict__ dout,
                                                        float* __restrict__ dsum, int B, int S, int H, int64_t o_bs, int o_rs,
                                                        int64_t d_bs, int d_rs)
{
    // one thread per 8 channels, 8 threads per (b, s, h) row of 64
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int part = (int)(i & 7);
    const int64_t row = i >> 3;                      // (b * S + s) * H + h
    const int h = (int)(row % H);
    const int64_t bs = row / H;
    const int sidx = (int)(bs % S), b = (int)(bs / S);
    float acc = 0.f;
    if (b < B) {
        const uint4 a = *(const uint4*)(o + b * o_bs + (int64_t)sidx * o_rs + h * 64 + part * 8);
        const uint4 g = *(const uint4*)(dout + b * d_bs + (int64_t)sidx * d_rs + h * 64 + part * 8);
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, gw[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int e = 0; e < 4; e++)
            acc += __uint_as_float(aw[e] << 16) * __uint_as_float(gw[e] << 16) +
                   __uint_as_float(aw[e] & 0xffff0000u) * __uint_as_float(gw[e] & 0xffff0000u);
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (part == 0 && b < B) dsum[((size_t)b * H + h) * S + sidx] = acc;
}

template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void attn_bwd_d64_kernel(
    const uint16_t* __restrict__ r1, int64_t r1_bs, int r1_rs, const uint16_t* __restrict__ r2, int64_t r2_bs, int r2_rs,
    const uint16_t* __restrict__ x1, int64_t x1_bs, int x1_rs, const uint16_t* __restrict__ x2, int64_t x2_bs, int x2_rs,
    const uint16_t* __restrict__ t1, const uint16_t* __restrict__ t2, const float* __restrict__ lse,
    const float* __restrict__ dsum, uint16_t* __restrict__ out1, int64_t o1_bs, int o1_rs, uint16_t* __restrict__ out2,
    int64_t o2_bs, int o2_rs, int Sr, int Sc, int H, float c, float scale, int r_len, int c_len, int Sq,
    int tiles_per_chunk /* 0: one workgroup walks every streamed tile */, float* __restrict__ part /* fp32 partials of the chunks */)
{
    // Chunked form (MODE 1 with a handful of keys -- the cross-attention over 77 text tokens has ONE key block per head): the
    // streamed (query) tiles are dealt to gridDim.z workgroups per key block, each leaves fp32 partial sums of dV^T / dK^T in
    // `part` [z][b h][gridDim.x * 32 WAVES rows][2][64], attn_bwd_reduce_kernel adds the chunks up in index order.
    constexpr int NT = MODE == 0 ? 3 : 4;            // tiles per stage: x1, x2, t1 (, t2)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y / H, h = blockIdx.y - b * H;
    const int fh = lane >> 5, fn = lane & 31;
    constexpr int THREADS = 64 * WAVES, NP = 512 / THREADS;     // 16-byte pieces of a 64 x 64 tile per thread
    const int rrow = blockIdx.x * (32 * WAVES) + wave * 32 + fn;
    const int rld = rrow < r_len ? rrow : r_len - 1;
    const float log2e = 1.4426950408889634f;

    bf16x8_t f1[4], f2[4];      // row-side fragments (B operands): lane (row fn, half fh) holds d = 16 kk + 8 fh .. + 7
    {
        const uint16_t* p1 = r1 + b * r1_bs + (int64_t)rld * r1_rs + h * kD + 8 * fh;
        const uint16_t* p2 = r2 + b * r2_bs + (int64_t)rld * r2_rs + h * kD + 8 * fh;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) { f1[kk] = *(const bf16x8_t*)(p1 + 16 * kk); f2[kk] = *(const bf16x8_t*)(p2 + 16 * kk); }
    }
    const size_t stat_base = ((size_t)b * H + h) * Sq;
    float Lr = 0.f, Dr = 0.f;
    if (MODE == 0) { Lr = lse[stat_base + rld] * log2e; Dr = dsum[stat_base + rld]; }

    const uint32_t x1_row = (uint32_t)x1_rs * 2u, x2_row = (uint32_t)x2_rs * 2u, t_row = (uint32_t)Sc * 2u;
    const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc((void*)(x1 + b * x1_bs + h * kD), 0, (int)((uint32_t)c_len * x1_row), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc((void*)(x2 + b * x2_bs + h * kD), 0, (int)((uint32_t)c_len * x2_row), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_t1 = __builtin_amdgcn_make_buffer_rsrc((void*)(t1 + (((int64_t)b * H + h) * kD) * Sc), 0, (int)((uint32_t)kD * t_row), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_t2 = __builtin_amdgcn_make_buffer_rsrc((void*)((MODE == 1 ? t2 : t1) + (((int64_t)b * H + h) * kD) * Sc), 0, (int)((uint32_t)kD * t_row), 0x00020000);
    uint32_t a_off[NP], b_off[NP], t_off[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const int cidx = tid + THREADS * i;
        const int line = cidx >> 4, cc = (cidx & 15) ^ (line & 15);
        const int r = 2 * line + (cc >> 3);
        a_off[i] = (uint32_t)r * x1_row + (uint32_t)(cc & 7) * 16u;
        b_off[i] = (uint32_t)r * x2_row + (uint32_t)(cc & 7) * 16u;
        t_off[i] = (uint32_t)r * t_row + (uint32_t)(cc & 7) * 16u;
    }
    auto issue = [&](int buf, int t) {
        char* st = smem + buf * (NT * kTile);
#pragma unroll
        for (int i = 0; i < NP; i++) {
            bload_lds16(rs_x1, a_off[i], (uint32_t)t * kTk * x1_row, st + (wave * 64 + THREADS * i) * 16);
            bload_lds16(rs_x2, b_off[i], (uint32_t)t * kTk * x2_row, st + kTile + (wave * 64 + THREADS * i) * 16);
            bload_lds16(rs_t1, t_off[i], (uint32_t)t * (kTk * 2), st + 2 * kTile + (wave * 64 + THREADS * i) * 16);
            if (MODE == 1) bload_lds16(rs_t2, t_off[i], (uint32_t)t * (kTk * 2), st + 3 * kTile + (wave * 64 + THREADS * i) * 16);
        }
    };
    uint32_t krd[2], vrd[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int p16 = fn & 15, key16 = p16 < 4 ? p16 : (p16 < 8 ? p16 + 4 : (p16 < 12 ? p16 - 4 : p16));
        krd[j] = (uint32_t)swz(32 * j + (fn & 16) + key16, fh);
        vrd[j] = (uint32_t)swz(32 * j + fn, fh);
    }
    f32x16 g0, g1, e0, e1;      // out1 (d blocks 0 / 1), out2 (MODE 1)
#pragma unroll
    for (int r = 0; r < 16; r++) g0[r] = g1[r] = e0[r] = e1[r] = 0.f;

    const int t_begin = tiles_per_chunk ? (int)blockIdx.z * tiles_per_chunk : 0;
    const int ntiles = tiles_per_chunk ? min(Sc / kTk, t_begin + tiles_per_chunk) : Sc / kTk;
    issue(t_begin & 1, t_begin);
    for (int t = t_begin; t < ntiles; t++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < ntiles) issue((t + 1) & 1, t + 1);
        const char* st = smem + (t & 1) * (NT * kTile);
        f32x16 s0, s1, p0, p1;
#pragma unroll
        for (int r = 0; r < 16; r++) s0[r] = s1[r] = p0[r] = p1[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const bf16x8_t a0 = *(const bf16x8_t*)(st + (krd[0] ^ (uint32_t)(kk << 5)));
            const bf16x8_t a1 = *(const bf16x8_t*)(st + (krd[1] ^ (uint32_t)(kk << 5)));
            const bf16x8_t b0 = *(const bf16x8_t*)(st + kTile + (krd[0] ^ (uint32_t)(kk << 5)));
            const bf16x8_t b1 = *(const bf16x8_t*)(st + kTile + (krd[1] ^ (uint32_t)(kk << 5)));
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, f1[kk], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, f1[kk], s1, 0, 0, 0);
            p0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, f2[kk], p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, f2[kk], p1, 0, 0, 0);
        }
        // accumulator row r of block j holds streamed element 32 j + 16 (r >> 3) + 8 fh + 4 ((r >> 2) & 1) + (r & 3)
        float Lq[2][16], Dq[2][16];
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int g = 0; g < 4; g++) {       // g = (r >> 2): u = g >> 1, w = g & 1
                    const size_t idx = stat_base + (size_t)t * kTk + 32 * j + 16 * (g >> 1) + 8 * fh + 4 * (g & 1);
                    const float4 l4 = *(const float4*)(lse + idx), d4 = *(const float4*)(dsum + idx);
                    Lq[j][4 * g] = l4.x * log2e; Lq[j][4 * g + 1] = l4.y * log2e; Lq[j][4 * g + 2] = l4.z * log2e; Lq[j][4 * g + 3] = l4.w * log2e;
                    Dq[j][4 * g] = d4.x; Dq[j][4 * g + 1] = d4.y; Dq[j][4 * g + 2] = d4.z; Dq[j][4 * g + 3] = d4.w;
                }
        }
        // P (into s*) and dS (into p*)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float m0 = 0.f, m1 = 0.f;      // masks of streamed padding (MODE 0: keys beyond c_len)
            if (MODE == 0) {
                const int key = t * kTk + 16 * (r >> 3) + 8 * fh + 4 * ((r >> 2) & 1) + (r & 3);
                m0 = key >= c_len ? -INFINITY : 0.f;
                m1 = key + 32 >= c_len ? -INFINITY : 0.f;
            }
            const float l0 = MODE == 0 ? Lr : Lq[0][r], l1 = MODE == 0 ? Lr : Lq[1][r];
            const float d0 = MODE == 0 ? Dr : Dq[0][r], d1 = MODE == 0 ? Dr : Dq[1][r];
            const float pa = __builtin_amdgcn_exp2f(s0[r] * c - l0 + m0), pb_ = __builtin_amdgcn_exp2f(s1[r] * c - l1 + m1);
            s0[r] = pa; s1[r] = pb_;
            p0[r] = pa * (p0[r] - d0);
            p1[r] = pb_ * (p1[r] - d1);
        }
        bf16x8_t pP[4], pS[4];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            uint32_t w0[4], w1[4], z0[4], z1[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                w0[i] = pack_bf16(s0[8 * u + 2 * i], s0[8 * u + 2 * i + 1]);
                w1[i] = pack_bf16(s1[8 * u + 2 * i], s1[8 * u + 2 * i + 1]);
                z0[i] = pack_bf16(p0[8 * u + 2 * i], p0[8 * u + 2 * i + 1]);
                z1[i] = pack_bf16(p1[8 * u + 2 * i], p1[8 * u + 2 * i + 1]);
            }
            pP[u] = __builtin_bit_cast(bf16x8_t, make_uint4(w0[0], w0[1], w0[2], w0[3]));
            pP[2 + u] = __builtin_bit_cast(bf16x8_t, make_uint4(w1[0], w1[1], w1[2], w1[3]));
            pS[u] = __builtin_bit_cast(bf16x8_t, make_uint4(z0[0], z0[1], z0[2], z0[3]));
            pS[2 + u] = __builtin_bit_cast(bf16x8_t, make_uint4(z1[0], z1[1], z1[2], z1[3]));
        }
#pragma unroll
        for (int g16 = 0; g16 < 4; g16++) {
            const bf16x8_t v0 = *(const bf16x8_t*)(st + 2 * kTile + (vrd[0] ^ (uint32_t)(g16 << 5)));
            const bf16x8_t v1 = *(const bf16x8_t*)(st + 2 * kTile + (vrd[1] ^ (uint32_t)(g16 << 5)));
            if (MODE == 0) {            // dQ^T += K^T dS^T
                g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pS[g16], g0, 0, 0, 0);
                g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pS[g16], g1, 0, 0, 0);
            } else {                    // dV^T += dO^T P,  dK^T += Q^T dS
                const bf16x8_t q0 = *(const bf16x8_t*)(st + 3 * kTile + (vrd[0] ^ (uint32_t)(g16 << 5)));
                const bf16x8_t q1 = *(const bf16x8_t*)(st + 3 * kTile + (vrd[1] ^ (uint32_t)(g16 << 5)));
                g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pP[g16], g0, 0, 0, 0);
                g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pP[g16], g1, 0, 0, 0);
                e0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q0, pS[g16], e0, 0, 0, 0);
                e1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q1, pS[g16], e1, 0, 0, 0);
            }
        }
    }
    if (MODE == 1 && part != nullptr) {
        if (rrow < r_len) {
            const size_t rows_pad = (size_t)gridDim.x * (32 * WAVES);
            float* pp = part + ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * rows_pad + rrow) * 2) * kD;
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {     // rows d = (r & 3) + 8 (r >> 2) + 4 fh of each 32-d block, as below
                *(float4*)(pp + 8 * g4 + 4 * fh) = make_float4(g0[4 * g4], g0[4 * g4 + 1], g0[4 * g4 + 2], g0[4 * g4 + 3]);
                *(float4*)(pp + 32 + 8 * g4 + 4 * fh) = make_float4(g1[4 * g4], g1[4 * g4 + 1], g1[4 * g4 + 2], g1[4 * g4 + 3]);
                *(float4*)(pp + kD + 8 * g4 + 4 * fh) = make_float4(e0[4 * g4], e0[4 * g4 + 1], e0[4 * g4 + 2], e0[4 * g4 + 3]);
                *(float4*)(pp + kD + 32 + 8 * g4 + 4 * fh) = make_float4(e1[4 * g4], e1[4 * g4 + 1], e1[4 * g4 + 2], e1[4 * g4 + 3]);
            }
        }
        return;
    }
    if (rrow < r_len) {
        const float s1f = MODE == 0 ? scale : 1.0f;
        uint16_t* op = out1 + b * o1_bs + (int64_t)rrow * o1_rs + h * kD;
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {         // rows d = (r & 3) + 8 (r >> 2) + 4 fh of each 32-d block
            uint2 a, bq;
            a.x = pack_bf16(g0[4 * g4] * s1f, g0[4 * g4 + 1] * s1f);
            a.y = pack_bf16(g0[4 * g4 + 2] * s1f, g0[4 * g4 + 3] * s1f);
            bq.x = pack_bf16(g1[4 * g4] * s1f, g1[4 * g4 + 1] * s1f);
            bq.y = pack_bf16(g1[4 * g4 + 2] * s1f, g1[4 * g4 + 3] * s1f);
            *(uint2*)(op + 8 * g4 + 4 * fh) = a;
            *(uint2*)(op + 32 + 8 * g4 + 4 * fh) = bq;
        }
        if (MODE == 1) {
            uint16_t* op2 = out2 + b * o2_bs + (int64_t)rrow * o2_rs + h * kD;
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                uint2 a, bq;
                a.x = pack_bf16(e0[4 * g4] * scale, e0[4 * g4 + 1] * scale);
                a.y = pack_bf16(e0[4 * g4 + 2] * scale, e0[4 * g4 + 3] * scale);
                bq.x = pack_bf16(e1[4 * g4] * scale, e1[4 * g4 + 1] * scale);
                bq.y = pack_bf16(e1[4 * g4 + 2] * scale, e1[4 * g4 + 3] * scale);
                *(uint2*)(op2 + 8 * g4 + 4 * fh) = a;
                *(uint2*)(op2 + 32 + 8 * g4 + 4 * fh) = bq;
            }
        }
    }
}

// dV = bf16(sum_z part[z][..][0][:]), dK = bf16(scale * sum_z part[z][..][1][:]): chunks added in index order (deterministic).
// One thread = 4 channels of one (b, h, key) row of one of the two outputs.
__global__ __launch_bounds__(256) void attn_bwd_reduce_kernel(const float* __restrict__ part, int chunks, int BH, int H,
                                                              int rows_pad, int r_len, uint16_t* __restrict__ out1, int64_t o1_bs,
                                                              int o1_rs, uint16_t* __restrict__ out2, int64_t o2_bs, int o2_rs,
                                                              float scale)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int d4 = (int)(i & 15), which = (int)((i >> 4) & 1);
    const int64_t row = i >> 5;                       // bh * r_len + key
    if (row >= (int64_t)BH * r_len) return;
    const int bh = (int)(row / r_len), key = (int)(row - (int64_t)bh * r_len);
    const int b = bh / H, h = bh - b * H;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < chunks; z++) {
        const float4 t = *(const float4*)(part + ((((size_t)z * BH + bh) * rows_pad + key) * 2 + which) * kD + 4 * d4);
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    const float f = which ? scale : 1.0f;
    uint16_t* op = which ? out2 + b * o2_bs + (int64_t)key * o2_rs + h * kD : out1 + b * o1_bs + (int64_t)key * o1_rs + h * kD;
    uint2 o;
    o.x = pack_bf16(acc.x * f, acc.y * f);
    o.y = pack_bf16(acc.z * f, acc.w * f);
    *(uint2*)(op + 4 * d4) = o;
}

// How many chunks of streamed (query) tiles the key-owning backward kernel is cut into: enough workgroups for the chip when
// a head has only one or two key blocks, at least two 64-query tiles per chunk.
static int bwd_key_chunks(int B, int S, int kv_len, int H)
{
    const int blocks = ((kv_len + 127) / 128) * B * H, tiles = S / kTk;
    if (blocks >= 128 || tiles < 4) return 1;
    int chunks = (256 + blocks - 1) / blocks;
    if (chunks > tiles / 2) chunks = tiles / 2;
    return chunks < 2 ? 1 : chunks;
}

}  // namespace

extern "C" {

const char* gd_nn_attention_last_error(void) { return g_err; }

size_t gd_nn_attention_ws_bytes(int B, int Skv, int H) { return (size_t)B * H * 64 * (size_t)Skv * 2; }

static int launch_attention(hipStream_t s, const void* q, const void* k, const void* vt, void* o, int B, int S, int Skv, int H,
                            int64_t q_bs, int q_rs, int64_t k_bs, int k_rs, int64_t o_bs, int o_rs, float scale, int kv_len,
                            float* lse = nullptr, int64_t vt_bs = -1)
{
    if (vt_bs < 0) vt_bs = (int64_t)H * kD * Skv;      // contiguous [B][H][64][Skv]
    if (const char* e = getenv("GD_NN_ATTN_WAVES")) g_attn_waves = atoi(e);
    if (const char* e = getenv("GD_NN_ATTN_XCD")) g_attn_xcd = atoi(e);      // 0: round-robin over the XCDs as before round 5 (A/B)
    const float c = scale * 1.4426950408889634f;
    int waves = 4;     // 8 waves per workgroup measured the same at batch 16 and worse on small grids (tools/attn_bench.py)
    if (g_attn_waves == 4 || g_attn_waves == 8) waves = g_attn_waves;
    if (waves == 8)
        hipLaunchKernelGGL(attn_fwd_d64_kernel<8>, dim3((S + 255) / 256, B * H), dim3(512), 6 * kTile, s, (const uint16_t*)q,
                           (const uint16_t*)k, (const uint16_t*)vt, (uint16_t*)o, S, Skv, H, q_bs, q_rs, k_bs, k_rs, o_bs,
                           o_rs, c, kv_len, lse, vt_bs, g_attn_xcd);
    else
        hipLaunchKernelGGL(attn_fwd_d64_kernel<4>, dim3((S + 127) / 128, B * H), dim3(256), 6 * kTile, s, (const uint16_t*)q,
                           (const uint16_t*)k, (const uint16_t*)vt, (uint16_t*)o, S, Skv, H, q_bs, q_rs, k_bs, k_rs, o_bs,
                           o_rs, c, kv_len, lse, vt_bs, g_attn_xcd);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

static int check_attention(const void* q, const void* k, const void* v, const void* o, int B, int S, int Skv, int H, int q_rs,
                           int k_rs, int o_rs, int kv_len)
{
    if (!q || !k || !v || !o) return fail(GD_NN_ERR_INVALID_ARG, "attention: null pointer");
    if (B <= 0 || S <= 0 || H <= 0 || Skv <= 0 || Skv % 64 || kv_len <= Skv - 64 || kv_len > Skv)
        return fail(GD_NN_ERR_INVALID_ARG, "attention: need Skv % 64 == 0 and Skv - 64 < kv_len <= Skv (head_dim is 64)");
    if (q_rs % 8 || k_rs % 8 || o_rs % 4 || (double)Skv * k_rs * 2.0 >= 2147483648.0)
        return fail(GD_NN_ERR_INVALID_ARG, "attention: row strides must keep 16-byte (q, k) / 8-byte (o) alignment");
    return GD_NN_OK;
}

int gd_nn_attention_d64_forward(void* stream, const void* q, const void* k, const void* v, void* o, void* vt_ws, int B, int S,
                                int Skv, int H, int64_t q_bs, int q_rs, int64_t k_bs, int k_rs, int64_t v_bs, int v_rs,
                                int64_t o_bs, int o_rs, float scale, int kv_len)
{
    if (!vt_ws) return fail(GD_NN_ERR_INVALID_ARG, "attention: null pointer");
    if (int e = check_attention(q, k, v, o, B, S, Skv, H, q_rs, k_rs, o_rs, kv_len)) return e;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(attn_vt_kernel, dim3(Skv / 64, H, B), dim3(256), 0, s, (const uint16_t*)v, (uint16_t*)vt_ws, Skv, H,
                       v_bs, v_rs, kv_len);
    return launch_attention(s, q, k, vt_ws, o, B, S, Skv, H, q_bs, q_rs, k_bs, k_rs, o_bs, o_rs, scale, kv_len);
}

int gd_nn_attention_d64_forward_lse(void* stream, const void* q, const void* k, const void* v, void* o, float* lse, void* vt_ws,
                                    int B, int S, int Skv, int H, int64_t q_bs, int q_rs, int64_t k_bs, int k_rs, int64_t v_bs,
                                    int v_rs, int64_t o_bs, int o_rs, float scale, int kv_len)
{
    if (!vt_ws || !lse) return fail(GD_NN_ERR_INVALID_ARG, "attention: null pointer");
    if (int e = check_attention(q, k, v, o, B, S, Skv, H, q_rs, k_rs, o_rs, kv_len)) return e;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(attn_vt_kernel, dim3(Skv / 64, H, B), dim3(256), 0, s, (const uint16_t*)v, (uint16_t*)vt_ws, Skv, H,
                       v_bs, v_rs, kv_len);
    return launch_attention(s, q, k, vt_ws, o, B, S, Skv, H, q_bs, q_rs, k_bs, k_rs, o_bs, o_rs, scale, kv_len, lse);
}

int gd_nn_attention_d64_forward_vt(void* stream, const void* q, const void* k, const void* vt, void* o, int B, int S, int Skv,
                                   int H, int64_t q_bs, int q_rs, int64_t k_bs, int k_rs, int64_t o_bs, int o_rs, float scale)
{
    if (int e = check_attention(q, k, vt, o, B, S, Skv, H, q_rs, k_rs, o_rs, Skv)) return e;
    return launch_attention((hipStream_t)stream, q, k, vt, o, B, S, Skv, H, q_bs, q_rs, k_bs, k_rs, o_bs, o_rs, scale, Skv);
}

int gd_nn_attention_d64_forward_vt_strided(void* stream, const void* q, const void* k, const void* vt, void* o, int B, int S,
                                           int Skv, int H, int64_t q_bs, int q_rs, int64_t k_bs, int k_rs, int64_t vt_bs,
                                           int64_t o_bs, int o_rs, float scale, int kv_len)
{
    if (int e = check_attention(q, k, vt, o, B, S, Skv, H, q_rs, k_rs, o_rs, kv_len)) return e;
    if (vt_bs < (int64_t)H * kD * Skv || vt_bs % 8)
        return fail(GD_NN_ERR_INVALID_ARG, "attention: vt batch stride must be >= H * 64 * Skv elements and a multiple of 8");
    return launch_attention((hipStream_t)stream, q, k, vt, o, B, S, Skv, H, q_bs, q_rs, k_bs, k_rs, o_bs, o_rs, scale, kv_len,
                            nullptr, vt_bs);
}

// Workspace of the backward pass: K^T [B][H][64][Skv], Q^T and dO^T [B][H][64][S] (bf16), D [B][H][S] (fp32).
size_t gd_nn_attention_bwd_ws_bytes(int B, int S, int Skv, int H)
{
    // ... + the fp32 partials of the chunked key-owning kernel (sized for the largest chunk count any kv_len <= Skv gives)
    const size_t base = (size_t)B * H * 64 * ((size_t)Skv + 2 * (size_t)S) * 2 + (size_t)B * H * S * 4;
    const int chunks = bwd_key_chunks(B, S, Skv, H);
    return base + (chunks > 1 ? (size_t)chunks * B * H * (size_t)((Skv + 127) / 128 * 128) * 2 * 64 * sizeof(float) : 0);
}

int gd_nn_attention_d64_backward(void* stream, const void* q, const void* k, const void* v, const void* o, const void* dout,
                                 const float* lse, void* dq, void* dk, void* dv, void* ws, int B, int S, int Skv, int H,
                                 int64_t q_bs, int q_rs, int64_t k_bs, int k_rs, int64_t v_bs, int v_rs, int64_t o_bs, int o_rs,
                                 int64_t do_bs, int do_rs, int64_t dq_bs, int dq_rs, int64_t dk_bs, int dk_rs, int64_t dv_bs,
                                 int dv_rs, float scale, int kv_len)
{
    if (!q || !k || !v || !o || !dout || !lse || !dq || !dk || !dv || !ws) return fail(GD_NN_ERR_INVALID_ARG, "attention_backward: null pointer");
    if (B <= 0 || S <= 0 || H <= 0 || Skv <= 0 || (S & 63) || (Skv & 63) || kv_len <= Skv - 64 || kv_len > Skv)
        return fail(GD_NN_ERR_INVALID_ARG, "attention_backward: need S % 64 == 0, Skv % 64 == 0, Skv - 64 < kv_len <= Skv");
    if (q_rs % 8 || k_rs % 8 || v_rs % 8 || o_rs % 8 || do_rs % 8 || dq_rs % 4 || dk_rs % 4 || dv_rs % 4 ||
        (double)Skv * k_rs * 2.0 >= 2147483648.0 || (double)S * q_rs * 2.0 >= 2147483648.0)
        return fail(GD_NN_ERR_INVALID_ARG, "attention_backward: row strides must keep 16-byte / 8-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    uint16_t* kt = (uint16_t*)ws;
    uint16_t* qt = kt + (size_t)B * H * 64 * Skv;
    uint16_t* dot = qt + (size_t)B * H * 64 * S;
    float* dsum = (float*)(dot + (size_t)B * H * 64 * S);
    {   // K^T, Q^T, dO^T: one launch (three 7 us launches per attention backward before round 5)
        const VtJob jk = {(const uint16_t*)k, kt, Skv, k_bs, k_rs, kv_len}, jq = {(const uint16_t*)q, qt, S, q_bs, q_rs, S},
                    jd = {(const uint16_t*)dout, dot, S, do_bs, do_rs, S};
        hipLaunchKernelGGL(attn_vt3_kernel, dim3((S > Skv ? S : Skv) / 64, H, 3 * B), dim3(256), 0, s, jk, jq, jd, H);
    }
    const int64_t rows8 = (int64_t)B * S * H * 8;
    hipLaunchKernelGGL(attn_dsum_kernel, dim3((unsigned)((rows8 + 255) / 256)), dim3(256), 0, s, (const uint16_t*)o,
                       (const uint16_t*)dout, dsum, B, S, H, o_bs, o_rs, do_bs, do_rs);
    const float c = scale * 1.4426950408889634f;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return fail(GD_NN_ERR_HIP, "hipGetDevice failed");
    static bool attr_set[16] = {false};      // per device, as every other launcher of the library keys it
    if (!attr_set[dev]) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_d64_kernel<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 3 * kTile);
        (void)hipFuncSetAttribute((const void*)attn_bwd_d64_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4 * kTile);
        attr_set[dev] = true;
    }
    // (64 row-side elements per workgroup -- the two-wave instantiation -- measured SLOWER on the one-latent grids it was meant
    // for: 309 against 260 us at S = 4096, 5 heads, tools/attn_bwd_bench.py; four waves always)
    // dQ: lanes = queries, keys stream
    hipLaunchKernelGGL((attn_bwd_d64_kernel<0, 4>), dim3((S + 127) / 128, B * H), dim3(256), 2 * 3 * kTile, s, (const uint16_t*)q,
                       q_bs, q_rs, (const uint16_t*)dout, do_bs, do_rs, (const uint16_t*)k, k_bs, k_rs, (const uint16_t*)v, v_bs,
                       v_rs, kt, (const uint16_t*)nullptr, lse, dsum, (uint16_t*)dq, dq_bs, dq_rs, (uint16_t*)nullptr, (int64_t)0,
                       0, S, Skv, H, c, scale, S, kv_len, S, 0, (float*)nullptr);
    // dK, dV: lanes = keys, queries stream -- in `chunks` workgroups per key block when a head has only a few keys
    const int chunks = bwd_key_chunks(B, S, kv_len, H) <= bwd_key_chunks(B, S, Skv, H) ? bwd_key_chunks(B, S, kv_len, H) : 1;
    const int gx = (kv_len + 127) / 128;
    if (chunks > 1) {
        float* part = (float*)((char*)(dsum + (size_t)B * H * S));
        const int tpc = (S / kTk + chunks - 1) / chunks, nz = (S / kTk + tpc - 1) / tpc;     // no empty chunk
        hipLaunchKernelGGL((attn_bwd_d64_kernel<1, 4>), dim3(gx, B * H, nz), dim3(256), 2 * 4 * kTile, s,
                           (const uint16_t*)k, k_bs, k_rs, (const uint16_t*)v, v_bs, v_rs, (const uint16_t*)q, q_bs, q_rs,
                           (const uint16_t*)dout, do_bs, do_rs, dot, qt, lse, dsum, (uint16_t*)dv, dv_bs, dv_rs, (uint16_t*)dk,
                           dk_bs, dk_rs, kv_len, S, H, c, scale, kv_len, S, S, tpc, part);
        const int64_t threads = (int64_t)B * H * kv_len * 32;
        hipLaunchKernelGGL(attn_bwd_reduce_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, part, nz, B * H, H,
                           gx * 128, kv_len, (uint16_t*)dv, dv_bs, dv_rs, (uint16_t*)dk, dk_bs, dk_rs, scale);
    } else {
        hipLaunchKernelGGL((attn_bwd_d64_kernel<1, 4>), dim3(gx, B * H), dim3(256), 2 * 4 * kTile, s,
                           (const uint16_t*)k, k_bs, k_rs, (const uint16_t*)v, v_bs, v_rs, (const uint16_t*)q, q_bs, q_rs,
                           (const uint16_t*)dout, do_bs, do_rs, dot, qt, lse, dsum, (uint16_t*)dv, dv_bs, dv_rs, (uint16_t*)dk,
                           dk_bs, dk_rs, kv_len, S, H, c, scale, kv_len, S, S, 0, (float*)nullptr);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

}  // extern "C"
