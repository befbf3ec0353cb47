// raster_densify.hip -- densify_and_prune of the Gaussian scene as two HIP passes over the flat parameter / Adam-moment
// buffers (include/gd_scene.h: gd_scene_densify_plan / gd_scene_densify_apply).
//
// Reference: GaussianModel.densify_and_prune -> densify_and_clone -> densify_and_split -> prune_points, each through
// densification_postfix / cat_tensors_to_optimizer / _prune_optimizer
// (Garment_3DGS/gaussiansplatting/scene/gaussian_model.py:283-413): three rounds of boolean-mask indexing, torch.cat
// and optimizer-state surgery over six parameter tensors and twelve moment tensors.  The composite is a pure function
// of the old point set: with  g = accum / denom (NaN -> 0),  s = max exp(scaling),  o = sigmoid(opacity)
//     clone   : sqrt(g g) >= thr  and  s <= dense          -> one copy (parameters copied, moments zero)
//     split   : g >= thr          and  s >  dense          -> original removed, two children:
//                                                             xyz = R(q) (z * exp(scaling)) + xyz,  z ~ N(0,1)
//                                                             scaling = log(exp(scaling) * (1 / 1.6))
//     prune   : o < min_opacity  or  (screen-size pruning on  and  s > 0.1 extent)     -- evaluated on the NEW set; the
//               reference's max_radii2D test never fires there because densification_postfix has just zeroed it
// and the surviving points keep the reference's order: originals, clones, first children, second children.
//
// Pass 1 (plan) classifies every point and leaves per-workgroup exclusive offsets of the four output streams + the
// four totals; the host reads the totals (the one sync: it has to size the new buffers), draws the 2 n_split x 3
// standard normals from the seeded generator (the same stream torch.normal(mean, std) consumes), and pass 2 (apply)
// writes the new flat parameter / exp_avg / exp_avg_sq buffers in one sweep.  Integer / byte work bounded by HBM.
// Every float expression follows the torch kernels of the reference's ops one rounding at a time (this file is built
// with -ffp-contract=off): true division, x * (1.0f / 1.6f) for the division by a Python scalar, 1 / (1 + exp(-x)).
#include <stdio.h>

#include "../../include/gd_scene.h"
#include "raster_common.h"

namespace gd {

namespace {

thread_local char g_densify_err[256] = "";
int dfail(int code, const char* msg)
{
    snprintf(g_densify_err, sizeof(g_densify_err), "%s", msg);
    return code;
}

constexpr uint32_t kKeepOrig = 1u, kKeepClone = 2u, kSplit = 4u, kKeepChild = 8u;

struct PlanArgs {
    float grad_threshold, dense_extent, min_opacity, max_world_scale;   // max_world_scale < 0: no world-size pruning
};

__device__ __forceinline__ uint32_t classify(int i, const float* __restrict__ accum, const float* __restrict__ denom,
                                             const float* __restrict__ opacity_raw, const float* __restrict__ scaling_raw,
                                             const PlanArgs a)
{
    float g = accum[i] / denom[i];
    if (g != g) g = 0.0f;                                               // grads[grads.isnan()] = 0.0
    const float s0 = expf(scaling_raw[3 * i]), s1 = expf(scaling_raw[3 * i + 1]), s2 = expf(scaling_raw[3 * i + 2]);
    const float smax = fmaxf(fmaxf(s0, s1), s2);
    const float opac = 1.0f / (1.0f + expf(-opacity_raw[i]));
    const bool clone = sqrtf(g * g) >= a.grad_threshold && smax <= a.dense_extent;     // torch.norm(grads, dim=-1)
    const bool split = g >= a.grad_threshold && smax > a.dense_extent;
    const bool low = opac < a.min_opacity;
    const bool prune_self = low || (a.max_world_scale >= 0.0f && smax > a.max_world_scale);
    // children: exp(log(s * (1 / 1.6)))
    const float inv = 1.0f / 1.6f;
    const float c0 = expf(logf(s0 * inv)), c1 = expf(logf(s1 * inv)), c2 = expf(logf(s2 * inv));
    const float cmax = fmaxf(fmaxf(c0, c1), c2);
    const bool prune_child = low || (a.max_world_scale >= 0.0f && cmax > a.max_world_scale);
    uint32_t c = 0;
    if (!split && !prune_self) c |= kKeepOrig;
    if (clone && !prune_self) c |= kKeepClone;
    if (split) c |= kSplit;
    if (split && !prune_child) c |= kKeepChild;
    return c;
}

// counts of the four class bits over the workgroup's 256 points -> block_counts[blk][4]
__global__ __launch_bounds__(256) void densify_classify_kernel(int P, const float* __restrict__ accum,
                                                               const float* __restrict__ denom,
                                                               const float* __restrict__ opacity_raw,
                                                               const float* __restrict__ scaling_raw, PlanArgs a,
                                                               uint8_t* __restrict__ cls, uint32_t* __restrict__ block_counts)
{
    __shared__ uint32_t s_cnt[4][4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t c = i < P ? classify(i, accum, denom, opacity_raw, scaling_raw, a) : 0u;
    if (i < P) cls[i] = (uint8_t)c;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t n = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(((c >> k) & 1u) != 0u));
        if (lane == 0) s_cnt[wave][k] = n;
    }
    __syncthreads();
    if (threadIdx.x < 4) block_counts[4 * blockIdx.x + threadIdx.x] =
        s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
}

// exclusive scan of block_counts[nblk][4] in place (one workgroup, thread k % 4 = stream); totals[4] receives the sums
__global__ __launch_bounds__(256) void densify_scan_kernel(uint32_t nblk, uint32_t* __restrict__ block_counts,
                                                           uint32_t* __restrict__ totals)
{
    __shared__ uint32_t s_part[64][4];
    const uint32_t k = threadIdx.x & 3u, t = threadIdx.x >> 2;      // 64 threads per stream
    const uint32_t per = (nblk + 63u) / 64u;
    const uint32_t lo = min(nblk, t * per), hi = min(nblk, lo + per);
    uint32_t sum = 0;
    for (uint32_t b = lo; b < hi; b++) sum += block_counts[4 * b + k];
    s_part[t][k] = sum;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t j = 0; j < t; j++) base += s_part[j][k];
    for (uint32_t b = lo; b < hi; b++) {
        const uint32_t v = block_counts[4 * b + k];
        block_counts[4 * b + k] = base;
        base += v;
    }
    if (t == 63u) totals[k] = base;
}

struct ApplyArgs {
    int P, newP, ngroups;
    int width[GD_SCENE_MAX_GROUPS];          // floats per point of each group
    int64_t old_begin[GD_SCENE_MAX_GROUPS], new_begin[GD_SCENE_MAX_GROUPS];
    int g_xyz, g_scaling, g_rotation;        // which groups get the split transform
    uint32_t n_orig, n_clone, n_split, n_child;
};

__global__ __launch_bounds__(256) void densify_apply_kernel(ApplyArgs a, const uint8_t* __restrict__ cls,
                                                            const uint32_t* __restrict__ block_offsets,
                                                            const float* __restrict__ z, const float* __restrict__ flat,
                                                            const float* __restrict__ exp_avg,
                                                            const float* __restrict__ exp_avg_sq, float* __restrict__ nflat,
                                                            float* __restrict__ nexp_avg, float* __restrict__ nexp_avg_sq)
{
    __shared__ uint32_t s_wave[4][4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t c = i < a.P ? cls[i] : 0u;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t rank[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint64_t m = __builtin_amdgcn_ballot_w64(((c >> k) & 1u) != 0u);
        rank[k] = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (lane == 0) s_wave[wave][k] = (uint32_t)__builtin_popcountll(m);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        for (uint32_t w = 0; w < wave; w++) rank[k] += s_wave[w][k];
        rank[k] += block_offsets[4 * blockIdx.x + k];
    }
    if (i >= a.P || c == 0u) return;
    // destinations in the new point order: originals | clones | first children | second children
    const int64_t d_orig = rank[0], d_clone = (int64_t)a.n_orig + rank[1];
    const int64_t d_ca = (int64_t)a.n_orig + a.n_clone + rank[3], d_cb = d_ca + a.n_child;
    float child_xyz[2][3] = {{0, 0, 0}, {0, 0, 0}}, child_scale[3] = {0, 0, 0};
    if (c & kKeepChild) {
        const float* sr = flat + a.old_begin[a.g_scaling] + 3 * (int64_t)i;
        const float* qr = flat + a.old_begin[a.g_rotation] + 4 * (int64_t)i;
        const float* xr = flat + a.old_begin[a.g_xyz] + 3 * (int64_t)i;
        const float s[3] = {expf(sr[0]), expf(sr[1]), expf(sr[2])};
        // build_rotation (utils/general_utils.py:78-99): q = r / |r|, then the nine entries
        const float norm = sqrtf(qr[0] * qr[0] + qr[1] * qr[1] + qr[2] * qr[2] + qr[3] * qr[3]);
        const float r = qr[0] / norm, x = qr[1] / norm, y = qr[2] / norm, zq = qr[3] / norm;
        const float R[3][3] = {{1 - 2 * (y * y + zq * zq), 2 * (x * y - r * zq), 2 * (x * zq + r * y)},
                               {2 * (x * y + r * zq), 1 - 2 * (x * x + zq * zq), 2 * (y * zq - r * x)},
                               {2 * (x * zq - r * y), 2 * (y * zq + r * x), 1 - 2 * (x * x + y * y)}};
        const float inv = 1.0f / 1.6f;
#pragma unroll
        for (int k = 0; k < 3; k++) child_scale[k] = logf(s[k] * inv);
#pragma unroll
        for (int n = 0; n < 2; n++) {
            // torch.normal(mean = 0, std): normal_(0, 1) * std + mean; the sample row of copy n is n * n_split + rank
            const float* zr = z + 3 * ((int64_t)n * a.n_split + rank[2]);
            const float smp[3] = {zr[0] * s[0] + 0.0f, zr[1] * s[1] + 0.0f, zr[2] * s[2] + 0.0f};
#pragma unroll
            for (int k = 0; k < 3; k++) child_xyz[n][k] = (R[k][0] * smp[0] + R[k][1] * smp[1] + R[k][2] * smp[2]) + xr[k];
        }
    }
    for (int g = 0; g < a.ngroups; g++) {
        const int w = a.width[g];
        const float* src = flat + a.old_begin[g] + (int64_t)w * i;
        const float* m1 = exp_avg + a.old_begin[g] + (int64_t)w * i;
        const float* m2 = exp_avg_sq + a.old_begin[g] + (int64_t)w * i;
        for (int k = 0; k < w; k++) {
            const float v = src[k];
            if (c & kKeepOrig) {
                const int64_t at = a.new_begin[g] + (int64_t)w * d_orig + k;
                nflat[at] = v; nexp_avg[at] = m1[k]; nexp_avg_sq[at] = m2[k];
            }
            if (c & kKeepClone) {
                const int64_t at = a.new_begin[g] + (int64_t)w * d_clone + k;
                nflat[at] = v; nexp_avg[at] = 0.0f; nexp_avg_sq[at] = 0.0f;
            }
            if (c & kKeepChild) {
                const int64_t ata = a.new_begin[g] + (int64_t)w * d_ca + k, atb = a.new_begin[g] + (int64_t)w * d_cb + k;
                float va = v, vb = v;
                if (g == a.g_xyz) { va = child_xyz[0][k]; vb = child_xyz[1][k]; }
                else if (g == a.g_scaling) { va = vb = child_scale[k]; }
                nflat[ata] = va; nflat[atb] = vb;
                nexp_avg[ata] = 0.0f; nexp_avg[atb] = 0.0f; nexp_avg_sq[ata] = 0.0f; nexp_avg_sq[atb] = 0.0f;
            }
        }
    }
}

}  // namespace
}  // namespace gd

extern "C" {

size_t gd_scene_densify_scratch_bytes(int P)
{
    const size_t nblk = ((size_t)(P < 0 ? 0 : P) + 255) / 256;
    return (size_t)(P < 0 ? 0 : P) + 256 + nblk * 16 + 64;   // cls[P] | block offsets [nblk][4] | totals[4]
}

int gd_scene_densify_plan(void* stream, int P, const float* xyz_gradient_accum, const float* denom,
                          const float* opacity_raw, const float* scaling_raw, float grad_threshold, float dense_extent,
                          float min_opacity, float max_world_scale, void* scratch, uint32_t* totals_host)
{
    using namespace gd;
    if (P <= 0) return dfail(-1, "densify_plan: P must be > 0");
    if (!xyz_gradient_accum || !denom || !opacity_raw || !scaling_raw || !scratch || !totals_host)
        return dfail(-1, "densify_plan: null pointer");
    const uint32_t nblk = (uint32_t)((P + 255) / 256);
    uint8_t* cls = static_cast<uint8_t*>(scratch);
    uint32_t* offs = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(cls) + (size_t)P + 255) & ~(uintptr_t)255);
    uint32_t* totals = offs + 4 * (size_t)nblk;
    PlanArgs a{grad_threshold, dense_extent, min_opacity, max_world_scale};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(densify_classify_kernel, dim3(nblk), dim3(256), 0, s, P, xyz_gradient_accum, denom, opacity_raw,
                       scaling_raw, a, cls, offs);
    hipLaunchKernelGGL(densify_scan_kernel, dim3(1), dim3(256), 0, s, nblk, offs, totals);
    // the one host sync of a densification event: the caller sizes the new buffers from the four totals
    hipError_t e = hipMemcpyAsync(totals_host, totals, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return dfail(-2, hipGetErrorString(e));
    return 0;
}

int gd_scene_densify_apply(void* stream, int P, int ngroups, const int* width, int g_xyz, int g_scaling, int g_rotation,
                           const uint32_t* totals_host, const void* scratch, const float* normals, const float* flat,
                           const float* exp_avg, const float* exp_avg_sq, float* new_flat, float* new_exp_avg,
                           float* new_exp_avg_sq)
{
    using namespace gd;
    if (P <= 0 || ngroups < 1 || ngroups > GD_SCENE_MAX_GROUPS || !width || !totals_host || !scratch)
        return dfail(-1, "densify_apply: bad arguments");
    if (g_xyz < 0 || g_xyz >= ngroups || g_scaling < 0 || g_scaling >= ngroups || g_rotation < 0 || g_rotation >= ngroups ||
        width[g_xyz] != 3 || width[g_scaling] != 3 || width[g_rotation] != 4)
        return dfail(-1, "densify_apply: xyz / scaling / rotation groups must have widths 3 / 3 / 4");
    ApplyArgs a;
    a.P = P; a.ngroups = ngroups; a.g_xyz = g_xyz; a.g_scaling = g_scaling; a.g_rotation = g_rotation;
    a.n_orig = totals_host[0]; a.n_clone = totals_host[1]; a.n_split = totals_host[2]; a.n_child = totals_host[3];
    const int64_t newP = (int64_t)a.n_orig + a.n_clone + 2 * (int64_t)a.n_child;
    if (newP > 0x7fffffff) return dfail(-1, "densify_apply: the new point count exceeds int32");
    a.newP = (int)newP;
    if (newP == 0) return 0;
    if (!flat || !exp_avg || !exp_avg_sq || !new_flat || !new_exp_avg || !new_exp_avg_sq || (a.n_child > 0 && !normals))
        return dfail(-1, "densify_apply: null pointer");
    int64_t ob = 0, nb = 0;
    for (int g = 0; g < ngroups; g++) {
        if (width[g] < 0) return dfail(-1, "densify_apply: negative group width");
        a.width[g] = width[g]; a.old_begin[g] = ob; a.new_begin[g] = nb;
        ob += (int64_t)width[g] * P; nb += (int64_t)width[g] * newP;
    }
    const uint32_t nblk = (uint32_t)((P + 255) / 256);
    const uint8_t* cls = static_cast<const uint8_t*>(scratch);
    const uint32_t* offs = reinterpret_cast<const uint32_t*>((reinterpret_cast<uintptr_t>(cls) + (size_t)P + 255) & ~(uintptr_t)255);
    hipLaunchKernelGGL(densify_apply_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a, cls, offs, normals, flat,
                       exp_avg, exp_avg_sq, new_flat, new_exp_avg, new_exp_avg_sq);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return dfail(-2, hipGetErrorString(e));
    return 0;
}

const char* gd_scene_densify_last_error(void) { return gd::g_densify_err; }

}  // extern "C"
