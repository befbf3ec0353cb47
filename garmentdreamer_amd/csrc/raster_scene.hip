// raster_scene.hip -- scene-side kernels either side of the rasterizer (C-ABI: include/gd_scene.h).
//
//   gd_scene_dist2         <- SimpleKNN::knn (simple-knn/simple_knn.cu:63-220): Morton order + boxed 3-NN
//   gd_scene_adam_step     <- torch.optim.Adam over the six attribute groups (scene/gaussian_model.py:156-167)
//   gd_scene_densify_stats <- add_densification_stats (scene/gaussian_model.py:415-419)
//
// Compiled with -ffp-contract=off: squared distances and Morton coordinates are compared bit for bit with the
// CPU oracle (oracle/gd_scene_oracle.c).
//
// dist2 on CDNA4.  The reference gives every thread its own walk over all boxes, each hit box re-read point by
// point through an index indirection (points[indices[i]]).  Here the points are gathered once into Morton order
// (contiguous float3), a workgroup = 256 consecutive Morton positions, and a box some lane needs is staged ONCE
// into LDS (12 KB) and scanned from there by the lanes that need it (same-address LDS reads broadcast).  The
// candidate order per query is the reference's (boxes ascending, points ascending), so the three best
// distances -- and their sum -- are identical.
#include <float.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gd_scene.h"
#include "raster_common.h"

namespace gd {

namespace {

thread_local char g_scene_err[256] = "";
int sfail(int code, const char* msg)
{
    snprintf(g_scene_err, sizeof(g_scene_err), "%s", msg);
    return code;
}

constexpr int kBox = GD_SCENE_KNN_BOX;

// order-preserving float <-> uint map so min / max can use integer atomics
__device__ __forceinline__ uint32_t f2ord(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// mm[0..2] = min xyz, mm[3..5] = max xyz (ordered-uint encoding), initialised to encode(0.0f):
// cub::DeviceReduce::Reduce(..., init = {0,0,0}) in the reference folds the origin into both boxes
// (simple_knn.cu:190-197).
__global__ void knn_minmax_init_kernel(uint32_t* mm)
{
    if (threadIdx.x < 6) mm[threadIdx.x] = 0x80000000u;   // f2ord(+0.0f)
}

__global__ __launch_bounds__(256) void knn_minmax_kernel(int P, const float* __restrict__ pts, uint32_t* mm)
{
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float v = pts[3 * (size_t)i + k];
            mn[k] = fminf(mn[k], v);
            mx[k] = fmaxf(mx[k], v);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], off, 64));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off, 64));
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            atomicMin(&mm[k], f2ord(mn[k]));
            atomicMax(&mm[3 + k], f2ord(mx[k]));
        }
    }
}

// simple_knn.cu:45-61
__device__ __forceinline__ uint32_t prep_morton(uint32_t x)
{
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}

__global__ __launch_bounds__(256) void knn_morton_kernel(int P, const float* __restrict__ pts,
                                                         const uint32_t* __restrict__ mm, uint64_t* __restrict__ keys,
                                                         uint32_t* __restrict__ vals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    uint32_t code = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float mn = ord2f(mm[k]), mx = ord2f(mm[3 + k]);
        const float t = ((pts[3 * (size_t)i + k] - mn) / (mx - mn)) * (float)((1 << 10) - 1);
        code |= prep_morton((uint32_t)t) << k;
    }
    keys[i] = code;
    vals[i] = (uint32_t)i;
}

// Gather into Morton order and reduce each run of kBox consecutive points to its bounding box
// (boxMinMax, simple_knn.cu:79-122).  One workgroup of 256 threads per box, 4 points per thread.
__global__ __launch_bounds__(256) void knn_gather_box_kernel(int P, const float* __restrict__ pts,
                                                             const uint32_t* __restrict__ idx,
                                                             float* __restrict__ spts, float* __restrict__ boxes)
{
    __shared__ float red[6][4];
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int j = threadIdx.x; j < kBox; j += 256) {
        const int i = blockIdx.x * kBox + j;
        if (i >= P) break;
        const uint32_t src = idx[i];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float v = pts[3 * (size_t)src + k];
            spts[3 * (size_t)i + k] = v;
            mn[k] = fminf(mn[k], v);
            mx[k] = fmaxf(mx[k], v);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], off, 64));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            red[k][threadIdx.x >> 6] = mn[k];
            red[3 + k][threadIdx.x >> 6] = mx[k];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float v = red[k][0];
        for (int w = 1; w < 4; w++) v = k < 3 ? fminf(v, red[k][w]) : fmaxf(v, red[k][w]);
        boxes[6 * (size_t)blockIdx.x + k] = v;
    }
}

// updateKBest<3> (simple_knn.cu:137-151)
__device__ __forceinline__ void update3(const float qx, const float qy, const float qz, const float px, const float py,
                                        const float pz, float (&best)[3])
{
    const float dx = px - qx, dy = py - qy, dz = pz - qz;
    float dist = dx * dx + dy * dy + dz * dz;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if (best[j] > dist) {
            const float t = best[j];
            best[j] = dist;
            dist = t;
        }
    }
}

// boxMeanDist (simple_knn.cu:153-186)
__global__ __launch_bounds__(256) void knn_mean_dist_kernel(int P, const float* __restrict__ spts,
                                                            const uint32_t* __restrict__ idx,
                                                            const float* __restrict__ boxes, int nboxes,
                                                            float* __restrict__ dists)
{
    __shared__ float s_box[kBox * 3];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < P;
    float qx = 0, qy = 0, qz = 0;
    float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    if (live) {
        qx = spts[3 * (size_t)i]; qy = spts[3 * (size_t)i + 1]; qz = spts[3 * (size_t)i + 2];
        const int lo = max(0, i - 3), hi = min(P - 1, i + 3);
        for (int j = lo; j <= hi; j++) {
            if (j == i) continue;
            update3(qx, qy, qz, spts[3 * (size_t)j], spts[3 * (size_t)j + 1], spts[3 * (size_t)j + 2], best);
        }
    }
    const float reject = best[2];
    best[0] = best[1] = best[2] = FLT_MAX;
    for (int b = 0; b < nboxes; b++) {
        bool need = false;
        if (live) {
            // distBoxPoint (simple_knn.cu:124-135)
            const float* bx = boxes + 6 * (size_t)b;
            float ddx = 0, ddy = 0, ddz = 0;
            if (qx < bx[0] || qx > bx[3]) ddx = fminf(fabsf(qx - bx[0]), fabsf(qx - bx[3]));
            if (qy < bx[1] || qy > bx[4]) ddy = fminf(fabsf(qy - bx[1]), fabsf(qy - bx[4]));
            if (qz < bx[2] || qz > bx[5]) ddz = fminf(fabsf(qz - bx[2]), fabsf(qz - bx[5]));
            const float dist = ddx * ddx + ddy * ddy + ddz * ddz;
            need = !(dist > reject || dist > best[2]);
        }
        // (also the barrier that separates the previous box's scan from the next staging)
        if (!__syncthreads_or(need ? 1 : 0)) continue;   // uniform: nobody in this workgroup needs box b
        const int b0 = b * kBox, cnt = min(kBox, P - b0);
        for (int j = threadIdx.x; j < cnt * 3; j += 256) s_box[j] = spts[3 * (size_t)b0 + j];
        __syncthreads();
        if (need) {
            for (int j = 0; j < cnt; j++) {
                if (b0 + j == i) continue;
                update3(qx, qy, qz, s_box[3 * j], s_box[3 * j + 1], s_box[3 * j + 2], best);
            }
        }
    }
    if (live) dists[idx[i]] = (best[0] + best[1] + best[2]) / 3.0f;
}

struct AdamGroups {
    int64_t end[GD_SCENE_MAX_GROUPS];
    float step_size[GD_SCENE_MAX_GROUPS];   // lr / (1 - beta1^t)
    int n;
};

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   AdamGroups grp, float beta2, float omb1, float omb2, float eps,
                                                   float bc2_sqrt)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int gi = 0;
        while (gi < grp.n - 1 && i >= grp.end[gi]) gi++;
        const float gr = g[i];
        const float mi = m[i] + omb1 * (gr - m[i]);          // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * beta2 + (omb2 * gr) * gr;    // mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] + (-grp.step_size[gi]) * (mi / denom);            // addcdiv_(exp_avg, denom, value=-step_size)
    }
}

__global__ __launch_bounds__(256) void densify_stats_kernel(int P, const int* __restrict__ radii,
                                                            const float* __restrict__ vg, float* __restrict__ max_r,
                                                            float* __restrict__ accum, float* __restrict__ denom)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int r = radii[i];
    if (r > 0) {
        max_r[i] = fmaxf(max_r[i], (float)r);
        const float gx = vg[3 * (size_t)i], gy = vg[3 * (size_t)i + 1];
        accum[i] += sqrtf(gx * gx + gy * gy);
        denom[i] += 1.0f;
    }
}

// Parameter activations of GaussianModel (scene/gaussian_model.py:95-115): one thread per Gaussian instead of
// exp / sigmoid / normalize (norm, clamp, div) / cat kernels and their autograd nodes.
__global__ __launch_bounds__(256) void activate_forward_kernel(int P, int M, const float* __restrict__ f_dc,
                                                               const float* __restrict__ f_rest,
                                                               const float* __restrict__ opacity_raw,
                                                               const float* __restrict__ scaling_raw,
                                                               const float* __restrict__ rotation_raw,
                                                               float* __restrict__ shs, float* __restrict__ opacity,
                                                               float* __restrict__ scales, float* __restrict__ rotations)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float* sh = shs + (size_t)i * M * 3;
    for (int k = 0; k < 3; k++) sh[k] = f_dc[(size_t)i * 3 + k];
    for (int k = 0; k < 3 * (M - 1); k++) sh[3 + k] = f_rest[(size_t)i * 3 * (M - 1) + k];
    opacity[i] = 1.0f / (1.0f + expf(-opacity_raw[i]));
    for (int k = 0; k < 3; k++) scales[(size_t)i * 3 + k] = expf(scaling_raw[(size_t)i * 3 + k]);
    const float4 q = *reinterpret_cast<const float4*>(rotation_raw + (size_t)i * 4);
    const float nrm = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);   // F.normalize eps
    *reinterpret_cast<float4*>(rotations + (size_t)i * 4) = make_float4(q.x / nrm, q.y / nrm, q.z / nrm, q.w / nrm);
}

// g_* += d(activated)/d(raw)^T . d_*  (accumulating, like AccumulateGrad on a zeroed .grad)
__global__ __launch_bounds__(256) void activate_backward_kernel(int P, int M, const float* __restrict__ opacity,
                                                                const float* __restrict__ scales,
                                                                const float* __restrict__ rotation_raw,
                                                                const float* __restrict__ d_shs,
                                                                const float* __restrict__ d_opacity,
                                                                const float* __restrict__ d_scales,
                                                                const float* __restrict__ d_rot, float* __restrict__ g_f_dc,
                                                                float* __restrict__ g_f_rest, float* __restrict__ g_opacity,
                                                                float* __restrict__ g_scaling, float* __restrict__ g_rotation)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    if (d_shs) {
        const float* ds = d_shs + (size_t)i * M * 3;
        for (int k = 0; k < 3; k++) g_f_dc[(size_t)i * 3 + k] += ds[k];
        for (int k = 0; k < 3 * (M - 1); k++) g_f_rest[(size_t)i * 3 * (M - 1) + k] += ds[3 + k];
    }
    if (d_opacity) {
        const float o = opacity[i];
        g_opacity[i] += d_opacity[i] * o * (1.0f - o);
    }
    if (d_scales)
        for (int k = 0; k < 3; k++) g_scaling[(size_t)i * 3 + k] += d_scales[(size_t)i * 3 + k] * scales[(size_t)i * 3 + k];
    if (d_rot) {
        const float4 q = *reinterpret_cast<const float4*>(rotation_raw + (size_t)i * 4);
        const float4 g = *reinterpret_cast<const float4*>(d_rot + (size_t)i * 4);
        const float len = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
        float4 r;
        if (len > 1e-12f) {   // d(q/|q|) = (g - n (n.g)) / |q|
            const float inv = 1.0f / len;
            const float4 n = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
            const float ng = n.x * g.x + n.y * g.y + n.z * g.z + n.w * g.w;
            r = make_float4((g.x - n.x * ng) * inv, (g.y - n.y * ng) * inv, (g.z - n.z * ng) * inv, (g.w - n.w * ng) * inv);
        } else {              // clamped denominator: q / 1e-12
            r = make_float4(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f, g.w * 1e12f);
        }
        float4* out = reinterpret_cast<float4*>(g_rotation + (size_t)i * 4);
        const float4 acc = *out;
        *out = make_float4(acc.x + r.x, acc.y + r.y, acc.z + r.z, acc.w + r.w);
    }
}

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct KnnScratch {
    uint32_t* mm;
    BinningState bin;
    float* spts;
    float* boxes;
    size_t total;
};

KnnScratch carve_knn(char* base, int P)
{
    KnnScratch k;
    size_t off = 0;
    k.mm = (uint32_t*)(base + off); off = align_up(off + 6 * sizeof(uint32_t));
    size_t used = 0;
    k.bin = carve_binning(base ? base + off : nullptr, (size_t)P, &used);
    off = align_up(off + used);
    k.spts = (float*)(base + off); off = align_up(off + (size_t)P * 3 * sizeof(float));
    const size_t nboxes = ((size_t)P + kBox - 1) / kBox;
    k.boxes = (float*)(base + off); off = align_up(off + nboxes * 6 * sizeof(float));
    k.total = off;
    return k;
}

}  // namespace
}  // namespace gd

extern "C" {

const char* gd_scene_last_error(void) { return gd::g_scene_err; }

size_t gd_scene_dist2_scratch_bytes(int P)
{
    if (P <= 0) return 256;
    return gd::carve_knn(nullptr, P).total;
}

int gd_scene_dist2(void* stream, int P, const float* points, float* mean_dists, void* scratch)
{
    using namespace gd;
    if (P < 0) return sfail(-1, "dist2: P must be >= 0");
    if (P == 0) return 0;
    if (!points || !mean_dists || !scratch) return sfail(-1, "dist2: null pointer");
    hipStream_t s = (hipStream_t)stream;
    KnnScratch k = carve_knn((char*)scratch, P);
    const int nblk = (P + 255) / 256;
    hipLaunchKernelGGL(knn_minmax_init_kernel, dim3(1), dim3(64), 0, s, k.mm);
    hipLaunchKernelGGL(knn_minmax_kernel, dim3(nblk < 1024 ? nblk : 1024), dim3(256), 0, s, P, points, k.mm);
    // keys / values are produced in the sorter's "alt" buffers: 3 passes of 10 bits end in the main ones
    hipLaunchKernelGGL(knn_morton_kernel, dim3(nblk), dim3(256), 0, s, P, points, k.mm, k.bin.keys_alt,
                       k.bin.point_list_alt);
    SortPlan plan;
    plan.total_bits = plan.live_bits = 30; plan.digit_bits = 10; plan.passes = 3;
    launch_radix_sort(s, k.bin, (uint32_t)P, plan, true);
    const int nboxes = (P + kBox - 1) / kBox;
    hipLaunchKernelGGL(knn_gather_box_kernel, dim3(nboxes), dim3(256), 0, s, P, points, k.bin.point_list, k.spts,
                       k.boxes);
    hipLaunchKernelGGL(knn_mean_dist_kernel, dim3(nblk), dim3(256), 0, s, P, k.spts, k.bin.point_list, k.boxes, nboxes,
                       mean_dists);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return sfail(-2, hipGetErrorString(e));
    return 0;
}

int gd_scene_adam_step(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                       int ngroups, const int64_t* group_end, const double* lr, double beta1, double beta2, double eps,
                       int step)
{
    using namespace gd;
    if (n < 0 || ngroups < 1 || ngroups > GD_SCENE_MAX_GROUPS || !group_end || !lr || step < 1)
        return sfail(-1, "adam: need 1 <= ngroups <= 8, group_end, lr, step >= 1");
    if (n == 0) return 0;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return sfail(-1, "adam: null pointer");
    AdamGroups g;
    g.n = ngroups;
    // scalars as torch computes them (python floats = double), rounded to fp32 only where torch's kernels do
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    int64_t prev = 0;
    for (int i = 0; i < GD_SCENE_MAX_GROUPS; i++) {
        g.end[i] = i < ngroups ? group_end[i] : n;
        g.step_size[i] = i < ngroups ? (float)(lr[i] / bc1) : 0.f;
        if (i < ngroups) {
            if (group_end[i] < prev || group_end[i] > n) return sfail(-1, "adam: group_end must be non-decreasing and <= n");
            prev = group_end[i];
        }
    }
    if (group_end[ngroups - 1] != n) return sfail(-1, "adam: the last group must end at n");
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream,
                       param, grad, exp_avg, exp_avg_sq, n, g, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps,
                       (float)sqrt(bc2));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return sfail(-2, hipGetErrorString(e));
    return 0;
}

int gd_scene_densify_stats(void* stream, int P, const int* radii, const float* viewspace_grad, float* max_radii2D,
                           float* xyz_gradient_accum, float* denom)
{
    using namespace gd;
    if (P < 0) return sfail(-1, "densify_stats: P must be >= 0");
    if (P == 0) return 0;
    if (!radii || !viewspace_grad || !max_radii2D || !xyz_gradient_accum || !denom)
        return sfail(-1, "densify_stats: null pointer");
    hipLaunchKernelGGL(densify_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, radii,
                       viewspace_grad, max_radii2D, xyz_gradient_accum, denom);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return sfail(-2, hipGetErrorString(e));
    return 0;
}

int gd_scene_activate_forward(void* stream, int P, int M, const float* f_dc, const float* f_rest, const float* opacity_raw,
                              const float* scaling_raw, const float* rotation_raw, float* shs, float* opacity,
                              float* scales, float* rotations)
{
    using namespace gd;
    if (P <= 0) return 0;
    if (M < 1 || !f_dc || (M > 1 && !f_rest) || !opacity_raw || !scaling_raw || !rotation_raw || !shs || !opacity ||
        !scales || !rotations)
        return sfail(-1, "activate_forward: null pointer or M < 1");
    hipLaunchKernelGGL(activate_forward_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, M, f_dc,
                       f_rest, opacity_raw, scaling_raw, rotation_raw, shs, opacity, scales, rotations);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return sfail(-2, hipGetErrorString(e));
    return 0;
}

int gd_scene_activate_backward(void* stream, int P, int M, const float* opacity, const float* scales,
                               const float* rotation_raw, const float* d_shs, const float* d_opacity,
                               const float* d_scales, const float* d_rotations, float* g_f_dc, float* g_f_rest,
                               float* g_opacity, float* g_scaling, float* g_rotation)
{
    using namespace gd;
    if (P <= 0) return 0;
    if (M < 1 || !opacity || !scales || !rotation_raw || !g_f_dc || (M > 1 && !g_f_rest) || !g_opacity || !g_scaling ||
        !g_rotation)
        return sfail(-1, "activate_backward: null pointer or M < 1");
    hipLaunchKernelGGL(activate_backward_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, M, opacity,
                       scales, rotation_raw, d_shs, d_opacity, d_scales, d_rotations, g_f_dc, g_f_rest, g_opacity,
                       g_scaling, g_rotation);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return sfail(-2, hipGetErrorString(e));
    return 0;
}

}  // extern "C"
