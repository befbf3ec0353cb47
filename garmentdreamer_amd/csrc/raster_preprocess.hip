// raster_preprocess.hip -- per-Gaussian kernels of the gfx950 rasterizer.
//
//   preprocess_kernel          <- preprocessCUDA (DGR/cuda_rasterizer/forward.cu:155-256)
//                                 + per-workgroup tile-count reduction (first half of the
//                                 InclusiveSum at rasterizer_impl.cu:278)
//   mark_visible_kernel        <- checkFrustum (rasterizer_impl.cu:54-66)
//   preprocess_backward_kernel <- computeCov2DCUDA + preprocessCUDA (backward.cu:144-274,
//                                 346-412), fused; the reference splits them only "due to
//                                 length" (backward.cu:141-143)
//
// This translation unit is compiled with -ffp-contract=off: radii, tile rectangles and the
// depth bits that become sort keys are derived from fp32 arithmetic here, and the parity
// contract is bit-exactness of those integers against oracle/gd_oracle.c (same expression
// order, no FMA contraction on either side).  HIP's default correctly-rounded fp32 divide
// and sqrt match the host's.  These kernels are HBM streams of <= ~150 B per Gaussian;
// arithmetic cost is irrelevant next to the render kernels.
//
// Matrix helper mirrors GLM's column-major mat3 (m[c][r]) including evaluation order.
#include "raster_common.h"
#include "raster_blend_math.h"

namespace gd {

namespace {

__device__ const float SH_C0 = 0.28209479177387814f;
__device__ const float SH_C1 = 0.4886025119029199f;
__device__ const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                   -1.0925484305920792f, 0.5462742152960396f};
__device__ const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                   0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                   -0.5900435899266435f};

struct Mat3 {
    float m[3][3];  // m[col][row]
};
__device__ __forceinline__ Mat3 mat3_cols(float a, float b, float c, float d, float e, float f, float g, float h,
                                          float i)
{
    Mat3 r;
    r.m[0][0] = a; r.m[0][1] = b; r.m[0][2] = c;
    r.m[1][0] = d; r.m[1][1] = e; r.m[1][2] = f;
    r.m[2][0] = g; r.m[2][1] = h; r.m[2][2] = i;
    return r;
}
__device__ __forceinline__ Mat3 mul(const Mat3& A, const Mat3& B)
{
    Mat3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int w = 0; w < 3; w++)
            r.m[c][w] = A.m[0][w] * B.m[c][0] + A.m[1][w] * B.m[c][1] + A.m[2][w] * B.m[c][2];
    return r;
}
__device__ __forceinline__ Mat3 transpose(const Mat3& A)
{
    Mat3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int w = 0; w < 3; w++) r.m[c][w] = A.m[w][c];
    return r;
}

__device__ __forceinline__ float ndc2pix(float v, int S)
{
    // double on purpose: the reference's literals are double (auxiliary.h:41-44)
    return (float)(((v + 1.0) * S - 1.0) * 0.5);
}

__device__ __forceinline__ void point4x3(const float p[3], const float* __restrict__ M, float o[3])
{
    o[0] = M[0] * p[0] + M[4] * p[1] + M[8] * p[2] + M[12];
    o[1] = M[1] * p[0] + M[5] * p[1] + M[9] * p[2] + M[13];
    o[2] = M[2] * p[0] + M[6] * p[1] + M[10] * p[2] + M[14];
}
__device__ __forceinline__ void point4x4(const float p[3], const float* __restrict__ M, float o[4])
{
    o[0] = M[0] * p[0] + M[4] * p[1] + M[8] * p[2] + M[12];
    o[1] = M[1] * p[0] + M[5] * p[1] + M[9] * p[2] + M[13];
    o[2] = M[2] * p[0] + M[6] * p[1] + M[10] * p[2] + M[14];
    o[3] = M[3] * p[0] + M[7] * p[1] + M[11] * p[2] + M[15];
}

// computeColorFromSH forward (forward.cu:20-71)
__device__ void sh_to_rgb(int g, size_t vp, int deg, int M, const float* __restrict__ means,
                          const float* __restrict__ campos, const float* __restrict__ shs,
                          uint8_t* __restrict__ clamped, float out[3])
{
    float dir[3] = {means[3 * g] - campos[0], means[3 * g + 1] - campos[1], means[3 * g + 2] - campos[2]};
    float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    float x = dir[0] / len, y = dir[1] / len, z = dir[2] / len;
    const float* sh = shs + (size_t)g * M * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
#define S(k) sh[3 * (k) + c]
        float r = SH_C0 * S(0);
        if (deg > 0) {
            r = r - SH_C1 * y * S(1) + SH_C1 * z * S(2) - SH_C1 * x * S(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2[0] * xy * S(4) + SH_C2[1] * yz * S(5) + SH_C2[2] * (2.0f * zz - xx - yy) * S(6) +
                    SH_C2[3] * xz * S(7) + SH_C2[4] * (xx - yy) * S(8);
                if (deg > 2) {
                    r = r + SH_C3[0] * y * (3.0f * xx - yy) * S(9) + SH_C3[1] * xy * z * S(10) +
                        SH_C3[2] * y * (4.0f * zz - xx - yy) * S(11) +
                        SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * S(12) +
                        SH_C3[4] * x * (4.0f * zz - xx - yy) * S(13) + SH_C3[5] * z * (xx - yy) * S(14) +
                        SH_C3[6] * x * (xx - 3.0f * yy) * S(15);
                }
            }
        }
#undef S
        r += 0.5f;
        clamped[3 * vp + c] = (r < 0);
        out[c] = fmaxf(r, 0.0f);
    }
}

// computeCov3D forward (forward.cu:118-152); quaternion deliberately NOT normalised (:127)
__device__ void cov3d_from_scale_rot(const float* __restrict__ s, float mod, const float* __restrict__ q,
                                     float* cov3D)
{
    Mat3 S = mat3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.m[0][0] = mod * s[0]; S.m[1][1] = mod * s[1]; S.m[2][2] = mod * s[2];
    float r = q[0], x = q[1], y = q[2], z = q[3];
    Mat3 R = mat3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                       2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                       2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    Mat3 Mm = mul(S, R);
    Mat3 Sg = mul(transpose(Mm), Mm);
    cov3D[0] = Sg.m[0][0]; cov3D[1] = Sg.m[0][1]; cov3D[2] = Sg.m[0][2];
    cov3D[3] = Sg.m[1][1]; cov3D[4] = Sg.m[1][2]; cov3D[5] = Sg.m[2][2];
}

struct Cov2DCtx {
    Mat3 T, Vrk, W;
    float t[3], txtz, tytz, limx, limy;
};

// computeCov2D forward (forward.cu:74-113), also the recompute of backward.cu:166-199
__device__ void cov2d(const float mean[3], float fx, float fy, float tanx, float tany, const float* cov3D,
                      const float* __restrict__ view, float out[3], Cov2DCtx* ctx)
{
    float t[3];
    point4x3(mean, view, t);
    const float limx = 1.3f * tanx, limy = 1.3f * tany;
    const float txtz = t[0] / t[2], tytz = t[1] / t[2];
    t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
    t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
    Mat3 J = mat3_cols(fx / t[2], 0.0f, -(fx * t[0]) / (t[2] * t[2]), 0.0f, fy / t[2],
                       -(fy * t[1]) / (t[2] * t[2]), 0, 0, 0);
    Mat3 Wm = mat3_cols(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    Mat3 T = mul(Wm, J);
    Mat3 Vrk = mat3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    Mat3 cov = mul(mul(transpose(T), transpose(Vrk)), T);
    cov.m[0][0] += 0.3f;
    cov.m[1][1] += 0.3f;
    out[0] = cov.m[0][0]; out[1] = cov.m[0][1]; out[2] = cov.m[1][1];
    if (ctx) {
        ctx->T = T; ctx->Vrk = Vrk; ctx->W = Wm;
        ctx->t[0] = t[0]; ctx->t[1] = t[1]; ctx->t[2] = t[2];
        ctx->txtz = txtz; ctx->tytz = tytz; ctx->limx = limx; ctx->limy = limy;
    }
}

__global__ __launch_bounds__(kGaussBlock) void preprocess_kernel(
    int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ scales, float scale_modifier,
    const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ shs,
    const float* __restrict__ cov3D_precomp, const float* __restrict__ colors_precomp,
    const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix, const float* __restrict__ cam_pos,
    int W, int H, ViewScalars vs, int* __restrict__ radii, GeomState gs, uint32_t gx, uint32_t gy, bool prefiltered)
{
    const size_t VP = (size_t)vs.V * P;
    const size_t vp = (size_t)blockIdx.x * kGaussBlock + threadIdx.x;
    uint32_t touched = 0;
    if (vp < VP) {
        const int v = (int)(vp / P);
        const int g = (int)(vp - (size_t)v * P);
        const float* view = viewmatrix + 16 * v;
        const float* proj = projmatrix + 16 * v;
        int my_r = 0;
        do {
            const float p_orig[3] = {means3D[3 * g], means3D[3 * g + 1], means3D[3 * g + 2]};
            float p_view[3];
            point4x3(p_orig, view, p_view);
            if (p_view[2] <= 0.2f) {  // in_frustum, auxiliary.h:153
                if (prefiltered) __builtin_trap();
                break;
            }
            float p_hom[4];
            point4x4(p_orig, proj, p_hom);
            float p_w = 1.0f / (p_hom[3] + 0.0000001f);
            float p_proj[3] = {p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w};
            float c3[6];
            if (cov3D_precomp != nullptr) {
#pragma unroll
                for (int i = 0; i < 6; i++) c3[i] = cov3D_precomp[6 * (size_t)g + i];
            } else {
                cov3d_from_scale_rot(scales + 3 * (size_t)g, scale_modifier, rotations + 4 * (size_t)g, c3);
#pragma unroll
                for (int i = 0; i < 6; i++) gs.cov3D[6 * vp + i] = c3[i];
            }
            float cov[3];
            cov2d(p_orig, vs.focal_x[v], vs.focal_y[v], vs.tan_fovx[v], vs.tan_fovy[v], c3, view, cov, nullptr);
            float det = (cov[0] * cov[2] - cov[1] * cov[1]);
            if (det == 0.0f) break;
            float det_inv = 1.f / det;
            float conic[3] = {cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv};
            float mid = 0.5f * (cov[0] + cov[2]);
            float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
            float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
            float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
            float px = ndc2pix(p_proj[0], W), py = ndc2pix(p_proj[1], H);
            uint32_t x0, y0, x1, y1;
            tile_rect(px, py, (int)my_radius, gx, gy, x0, y0, x1, y1);
            if ((x1 - x0) * (y1 - y0) == 0) break;
            float rgb[3];
            if (colors_precomp == nullptr) sh_to_rgb(g, vp, D, M, means3D, cam_pos + 3 * v, shs, gs.clamped, rgb);
            else {
                rgb[0] = colors_precomp[3 * (size_t)g]; rgb[1] = colors_precomp[3 * (size_t)g + 1];
                rgb[2] = colors_precomp[3 * (size_t)g + 2];
            }
            my_r = (int)my_radius;
            gs.means2D[vp] = make_float2(px, py);
            gs.conic_opacity[vp] = make_float4(conic[0], conic[1], conic[2], opacities[g]);
            gs.alpha_thr[vp] = alpha_threshold_exact(opacities[g]);     // the blend's contribution test, once per Gaussian
            gs.rgbd[vp] = make_float4(rgb[0], rgb[1], rgb[2], p_view[2]);
            touched = (y1 - y0) * (x1 - x0);
        } while (false);
        radii[vp] = my_r;
        gs.tiles_touched[vp] = touched;
    }
    // workgroup total of tiles_touched -> block_sums[blockIdx]
    __shared__ uint32_t wave_tot[kGaussBlock / 64];
    uint32_t t = touched;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) gs.block_sums[blockIdx.x] = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
}

__global__ void mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ view,
                                    uint8_t* __restrict__ present)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const float p[3] = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    float pv[3];
    point4x3(p, view, pv);
    present[idx] = !(pv[2] <= 0.2f);
}

// computeColorFromSH backward (backward.cu:20-139).  Returns the mean-gradient part; SH
// gradients are written (first == true) or accumulated into dsh.
__device__ void sh_backward(int g, size_t vp, int deg, int M, const float* __restrict__ means,
                            const float* __restrict__ campos, const float* __restrict__ shs,
                            const uint8_t* __restrict__ clamped, const float dL_dcolor[3], float dmean[3],
                            float* __restrict__ dsh, bool first)
{
    float dir_orig[3] = {means[3 * g] - campos[0], means[3 * g + 1] - campos[1], means[3 * g + 2] - campos[2]};
    float len = sqrtf(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
    float x = dir_orig[0] / len, y = dir_orig[1] / len, z = dir_orig[2] / len;
    const float* sh = shs + (size_t)g * M * 3;
    float dRGB[3];
#pragma unroll
    for (int c = 0; c < 3; c++) dRGB[c] = dL_dcolor[c] * (clamped[3 * vp + c] ? 0 : 1);
    float dx[3] = {0, 0, 0}, dy[3] = {0, 0, 0}, dz[3] = {0, 0, 0};
#pragma unroll
    for (int c = 0; c < 3; c++) {
#define S(k) sh[3 * (k) + c]
#define G(k, val)                         \
    do {                                  \
        float _v = (val);                 \
        if (first) dsh[3 * (k) + c] = _v; \
        else dsh[3 * (k) + c] += _v;      \
    } while (0)
        G(0, SH_C0 * dRGB[c]);
        if (deg > 0) {
            float d1 = -SH_C1 * y, d2 = SH_C1 * z, d3 = -SH_C1 * x;
            G(1, d1 * dRGB[c]); G(2, d2 * dRGB[c]); G(3, d3 * dRGB[c]);
            dx[c] = -SH_C1 * S(3); dy[c] = -SH_C1 * S(1); dz[c] = SH_C1 * S(2);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                G(4, (SH_C2[0] * xy) * dRGB[c]);
                G(5, (SH_C2[1] * yz) * dRGB[c]);
                G(6, (SH_C2[2] * (2.f * zz - xx - yy)) * dRGB[c]);
                G(7, (SH_C2[3] * xz) * dRGB[c]);
                G(8, (SH_C2[4] * (xx - yy)) * dRGB[c]);
                dx[c] += SH_C2[0] * y * S(4) + SH_C2[2] * 2.f * -x * S(6) + SH_C2[3] * z * S(7) + SH_C2[4] * 2.f * x * S(8);
                dy[c] += SH_C2[0] * x * S(4) + SH_C2[1] * z * S(5) + SH_C2[2] * 2.f * -y * S(6) + SH_C2[4] * 2.f * -y * S(8);
                dz[c] += SH_C2[1] * y * S(5) + SH_C2[2] * 2.f * 2.f * z * S(6) + SH_C2[3] * x * S(7);
                if (deg > 2) {
                    G(9, (SH_C3[0] * y * (3.f * xx - yy)) * dRGB[c]);
                    G(10, (SH_C3[1] * xy * z) * dRGB[c]);
                    G(11, (SH_C3[2] * y * (4.f * zz - xx - yy)) * dRGB[c]);
                    G(12, (SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * dRGB[c]);
                    G(13, (SH_C3[4] * x * (4.f * zz - xx - yy)) * dRGB[c]);
                    G(14, (SH_C3[5] * z * (xx - yy)) * dRGB[c]);
                    G(15, (SH_C3[6] * x * (xx - 3.f * yy)) * dRGB[c]);
                    dx[c] += (SH_C3[0] * S(9) * 3.f * 2.f * xy + SH_C3[1] * S(10) * yz + SH_C3[2] * S(11) * -2.f * xy +
                              SH_C3[3] * S(12) * -3.f * 2.f * xz + SH_C3[4] * S(13) * (-3.f * xx + 4.f * zz - yy) +
                              SH_C3[5] * S(14) * 2.f * xz + SH_C3[6] * S(15) * 3.f * (xx - yy));
                    dy[c] += (SH_C3[0] * S(9) * 3.f * (xx - yy) + SH_C3[1] * S(10) * xz +
                              SH_C3[2] * S(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * S(12) * -3.f * 2.f * yz +
                              SH_C3[4] * S(13) * -2.f * xy + SH_C3[5] * S(14) * -2.f * yz +
                              SH_C3[6] * S(15) * -3.f * 2.f * xy);
                    dz[c] += (SH_C3[1] * S(10) * xy + SH_C3[2] * S(11) * 4.f * 2.f * yz +
                              SH_C3[3] * S(12) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * S(13) * 4.f * 2.f * xz +
                              SH_C3[5] * S(14) * (xx - yy));
                }
            }
        }
#undef S
#undef G
    }
    float ddir[3];
    ddir[0] = dx[0] * dRGB[0] + dx[1] * dRGB[1] + dx[2] * dRGB[2];
    ddir[1] = dy[0] * dRGB[0] + dy[1] * dRGB[1] + dy[2] * dRGB[2];
    ddir[2] = dz[0] * dRGB[0] + dz[1] * dRGB[1] + dz[2] * dRGB[2];
    const float* v = dir_orig;  // dnormvdv, auxiliary.h:109-119
    float sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    dmean[0] = ((+sum2 - v[0] * v[0]) * ddir[0] - v[1] * v[0] * ddir[1] - v[2] * v[0] * ddir[2]) * invsum32;
    dmean[1] = (-v[0] * v[1] * ddir[0] + (sum2 - v[1] * v[1]) * ddir[1] - v[2] * v[1] * ddir[2]) * invsum32;
    dmean[2] = (-v[0] * v[2] * ddir[0] - v[1] * v[2] * ddir[1] + (sum2 - v[2] * v[2]) * ddir[2]) * invsum32;
}

// acc[vp][10] = sum of the rows the backward blend stored for the instance slots of (view, Gaussian) vp -- one slot per
// tile the splat touches, contiguous: point_offsets[vp] - tiles_touched[vp] ... ; rowpos[slot][strip] = 1 + the row's
// index (0: that strip blended nothing of the instance): deterministic, no atomics.  A quad of lanes per vp takes up to
// kOwnSlots slots itself; the few splats that cover more tiles are finished by the whole wave (64 slots per step +
// a butterfly sum), so that one large splat does not hold 63 idle lanes for hundreds of dependent loads.
constexpr uint32_t kOwnSlots = 32;   // per quad of lanes

__device__ __forceinline__ void add_slot_rows(const float* __restrict__ rows, const uint4* __restrict__ rowpos,
                                              uint32_t o, float a[10])
{
    const uint4 rp = rowpos[o];
    const uint32_t at[4] = {rp.x, rp.y, rp.z, rp.w};
#pragma unroll
    for (int w = 0; w < 4; w++) {
        if (at[w] == 0u) continue;
        const float2* row = reinterpret_cast<const float2*>(rows + 10 * (size_t)(at[w] - 1u));
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const float2 t = row[k];
            a[2 * k] += t.x; a[2 * k + 1] += t.y;
        }
    }
}

__global__ __launch_bounds__(256) void instance_sum_kernel(uint32_t VP, const int* __restrict__ radii,
                                                           const uint32_t* __restrict__ point_offsets,
                                                           const uint32_t* __restrict__ tiles_touched,
                                                           const float* __restrict__ rows,
                                                           const uint4* __restrict__ rowpos, float* __restrict__ acc)
{
    // a QUAD of lanes per (view, Gaussian): lane q takes the slots first + q, first + q + 4, ... -- four independent
    // chains of (rowpos -> row) gathers per splat instead of one -- and two DPP adds per value combine the quad
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t vp = gid >> 2, q = gid & 3u;
    const uint32_t lane = threadIdx.x & 63u;
    const bool vis = vp < VP && radii[vp] > 0;
    const uint32_t end = vis ? point_offsets[vp] : 0u;
    const uint32_t n = vis ? tiles_touched[vp] : 0u;
    const uint32_t first = end - n;
    float a[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t own_end = first + min(n, kOwnSlots);
    for (uint32_t o = first + q; o < own_end; o += 4u) add_slot_rows(rows, rowpos, o, a);
#pragma unroll
    for (int k = 0; k < 10; k++) {
        a[k] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[k]), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
        a[k] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[k]), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    }
    // the few splats that cover more tiles are finished by the whole wave (64 slots per step + a butterfly sum)
    for (uint64_t big = __builtin_amdgcn_ballot_w64(n > kOwnSlots && q == 0u); big; big &= big - 1) {
        const int src = (int)__builtin_ctzll(big);
        const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)own_end, src);
        const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)end, src);
        float part[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t o = f + lane; o < e; o += 64u) add_slot_rows(rows, rowpos, o, part);
#pragma unroll
        for (int k = 0; k < 10; k++) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) part[k] += __shfl_xor(part[k], off, 64);
            if ((int)lane == src) a[k] += part[k];
        }
    }
    if (vp < VP && q == 0u) {
        float2* dst = reinterpret_cast<float2*>(acc + 10 * (size_t)vp);
#pragma unroll
        for (int k = 0; k < 5; k++) dst[k] = make_float2(a[2 * k], a[2 * k + 1]);
    }
}

// One thread per Gaussian; loops over the V views so that every output element is written
// exactly once, in a fixed order (deterministic, no pre-zeroing, no atomics).
// acc[vp][10] = {dL_dcolor.rgb, dL_ddepth, dL_dmean2D.xy, dL_dconic.x/.y/.w, dL_dopacity} of (view, Gaussian), summed
// over the Gaussian's tiles by instance_sum_kernel (the reference accumulates the same terms with atomicAdd,
// backward.cu:555-598).
__global__ __launch_bounds__(kGaussBlock) void preprocess_backward_kernel(
    int P, int D, int M, const float* __restrict__ means3D, const int* __restrict__ radii,
    const float* __restrict__ shs, const uint8_t* __restrict__ clamped, const float* __restrict__ scales,
    const float* __restrict__ rotations, float scale_modifier, const float* __restrict__ cov3D,
    size_t cov3D_view_stride, const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix,
    const float* __restrict__ campos, ViewScalars vs, const float* __restrict__ acc, bool has_colors_precomp,
    float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity,
    float* __restrict__ dL_dcolor, float* __restrict__ dL_ddepth, float* __restrict__ dL_dmean3D,
    float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh, float* __restrict__ dL_dscale,
    float* __restrict__ dL_drot)
{
    const int g = blockIdx.x * kGaussBlock + threadIdx.x;
    if (g >= P) return;
    const float m[3] = {means3D[3 * g], means3D[3 * g + 1], means3D[3 * g + 2]};
    float s_mean[3] = {0, 0, 0}, s_scale[3] = {0, 0, 0}, s_rot[4] = {0, 0, 0, 0}, s_cov[6] = {0, 0, 0, 0, 0, 0};
    float s_opac = 0, s_col[3] = {0, 0, 0};
    bool sh_first = true;
    for (int v = 0; v < vs.V; v++) {
        const size_t vp = (size_t)v * P + g;
        const bool vis = radii[vp] > 0;
        const float* a = acc + 10 * vp;
        // per-view outputs (exist for parity with the reference's intermediates)
        if (dL_dmean2D) {
            dL_dmean2D[3 * vp] = vis ? a[4] : 0.f; dL_dmean2D[3 * vp + 1] = vis ? a[5] : 0.f;
            dL_dmean2D[3 * vp + 2] = 0.f;
        }
        if (dL_dconic) {
            dL_dconic[4 * vp] = vis ? a[6] : 0.f; dL_dconic[4 * vp + 1] = vis ? a[7] : 0.f;
            dL_dconic[4 * vp + 2] = 0.f; dL_dconic[4 * vp + 3] = vis ? a[8] : 0.f;
        }
        if (dL_ddepth) dL_ddepth[vp] = vis ? a[3] : 0.f;
        if (!vis) continue;
        const float* view = viewmatrix + 16 * v;
        const float* proj = projmatrix + 16 * v;
        const float dcol[3] = {a[0], a[1], a[2]};
        const float ddep = a[3], gx2 = a[4], gy2 = a[5];
        const float dc[3] = {a[6], a[7], a[8]};
        s_opac += a[9];
        s_col[0] += dcol[0]; s_col[1] += dcol[1]; s_col[2] += dcol[2];
        // ---- computeCov2DCUDA, backward.cu:144-274 ----
        float c3[6];
        {
            const float* cp = cov3D + (size_t)v * cov3D_view_stride + 6 * (size_t)g;
#pragma unroll
            for (int i = 0; i < 6; i++) c3[i] = cp[i];
        }
        float cov[3];
        Cov2DCtx cx;
        cov2d(m, vs.focal_x[v], vs.focal_y[v], vs.tan_fovx[v], vs.tan_fovy[v], c3, view, cov, &cx);
        const float x_grad_mul = (cx.txtz < -cx.limx || cx.txtz > cx.limx) ? 0 : 1;
        const float y_grad_mul = (cx.tytz < -cx.limy || cx.tytz > cx.limy) ? 0 : 1;
        const float ca = cov[0], cb = cov[1], cc = cov[2];
        const float denom = ca * cc - cb * cb;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float dcov[6] = {0, 0, 0, 0, 0, 0};
#define TT(i, j) cx.T.m[i][j]
#define VV(i, j) cx.Vrk.m[i][j]
#define WW(i, j) cx.W.m[i][j]
        if (denom2inv != 0) {
            dL_da = denom2inv * (-cc * cc * dc[0] + 2 * cb * cc * dc[1] + (denom - ca * cc) * dc[2]);
            dL_dc = denom2inv * (-ca * ca * dc[2] + 2 * ca * cb * dc[1] + (denom - ca * cc) * dc[0]);
            dL_db = denom2inv * 2 * (cb * cc * dc[0] - (denom + 2 * cb * cb) * dc[1] + ca * cb * dc[2]);
            dcov[0] = (TT(0, 0) * TT(0, 0) * dL_da + TT(0, 0) * TT(1, 0) * dL_db + TT(1, 0) * TT(1, 0) * dL_dc);
            dcov[3] = (TT(0, 1) * TT(0, 1) * dL_da + TT(0, 1) * TT(1, 1) * dL_db + TT(1, 1) * TT(1, 1) * dL_dc);
            dcov[5] = (TT(0, 2) * TT(0, 2) * dL_da + TT(0, 2) * TT(1, 2) * dL_db + TT(1, 2) * TT(1, 2) * dL_dc);
            dcov[1] = 2 * TT(0, 0) * TT(0, 1) * dL_da + (TT(0, 0) * TT(1, 1) + TT(0, 1) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 1) * dL_dc;
            dcov[2] = 2 * TT(0, 0) * TT(0, 2) * dL_da + (TT(0, 0) * TT(1, 2) + TT(0, 2) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 2) * dL_dc;
            dcov[4] = 2 * TT(0, 2) * TT(0, 1) * dL_da + (TT(0, 1) * TT(1, 2) + TT(0, 2) * TT(1, 1)) * dL_db + 2 * TT(1, 1) * TT(1, 2) * dL_dc;
        }
        float dL_dT00 = 2 * (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_da +
                        (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_db;
        float dL_dT01 = 2 * (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_da +
                        (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_db;
        float dL_dT02 = 2 * (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_da +
                        (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_db;
        float dL_dT10 = 2 * (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_dc +
                        (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_db;
        float dL_dT11 = 2 * (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_dc +
                        (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_db;
        float dL_dT12 = 2 * (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_dc +
                        (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_db;
        float dL_dJ00 = WW(0, 0) * dL_dT00 + WW(0, 1) * dL_dT01 + WW(0, 2) * dL_dT02;
        float dL_dJ02 = WW(2, 0) * dL_dT00 + WW(2, 1) * dL_dT01 + WW(2, 2) * dL_dT02;
        float dL_dJ11 = WW(1, 0) * dL_dT10 + WW(1, 1) * dL_dT11 + WW(1, 2) * dL_dT12;
        float dL_dJ12 = WW(2, 0) * dL_dT10 + WW(2, 1) * dL_dT11 + WW(2, 2) * dL_dT12;
#undef TT
#undef VV
#undef WW
        const float tz = 1.f / cx.t[2];
        const float tz2 = tz * tz;
        const float tz3 = tz2 * tz;
        const float h_x = vs.focal_x[v], h_y = vs.focal_y[v];
        float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
        float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
        float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * cx.t[0]) * tz3 * dL_dJ02 +
                       (2 * h_y * cx.t[1]) * tz3 * dL_dJ12;
        float dmean[3];
        dmean[0] = view[0] * dL_dtx + view[1] * dL_dty + view[2] * dL_dtz;
        dmean[1] = view[4] * dL_dtx + view[5] * dL_dty + view[6] * dL_dtz;
        dmean[2] = view[8] * dL_dtx + view[9] * dL_dty + view[10] * dL_dtz;
        // ---- preprocessCUDA backward, backward.cu:346-412 ----
        float m_hom[4];
        point4x4(m, proj, m_hom);
        float m_w = 1.0f / (m_hom[3] + 0.0000001f);
        float mul1 = (proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12]) * m_w * m_w;
        float mul2 = (proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13]) * m_w * m_w;
        float d0 = (proj[0] * m_w - proj[3] * mul1) * gx2 + (proj[1] * m_w - proj[3] * mul2) * gy2;
        float d1 = (proj[4] * m_w - proj[7] * mul1) * gx2 + (proj[5] * m_w - proj[7] * mul2) * gy2;
        float d2 = (proj[8] * m_w - proj[11] * mul1) * gx2 + (proj[9] * m_w - proj[11] * mul2) * gy2;
        dmean[0] += d0; dmean[1] += d1; dmean[2] += d2;
        float mul3 = view[2] * m[0] + view[6] * m[1] + view[10] * m[2] + view[14];
        float e0 = (view[2] - view[3] * mul3) * ddep;
        float e1 = (view[6] - view[7] * mul3) * ddep;
        float e2 = (view[10] - view[11] * mul3) * ddep;
        dmean[0] += e0; dmean[1] += e1; dmean[2] += e2;
        if (shs) {
            float sm[3];
            if (sh_first) {
                // bands above the active degree receive no gradient: exact zeros, as the reference's
                // torch::zeros output has (rasterize_points.cu:160); the output buffer is NOT pre-zeroed
                for (int i = 3 * (D + 1) * (D + 1); i < 3 * M; i++) dL_dsh[(size_t)g * M * 3 + i] = 0.f;
            }
            sh_backward(g, vp, D, M, means3D, campos + 3 * v, shs, clamped, dcol, sm, dL_dsh + (size_t)g * M * 3,
                        sh_first);
            sh_first = false;
            dmean[0] += sm[0]; dmean[1] += sm[1]; dmean[2] += sm[2];
        }
        s_mean[0] += dmean[0]; s_mean[1] += dmean[1]; s_mean[2] += dmean[2];
#pragma unroll
        for (int i = 0; i < 6; i++) s_cov[i] += dcov[i];
        if (scales) {
            // computeCov3D backward, backward.cu:278-341
            const float* sc = scales + 3 * (size_t)g;
            const float* q = rotations + 4 * (size_t)g;
            float r = q[0], x = q[1], y = q[2], z = q[3];
            Mat3 R = mat3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                               2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                               2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
            Mat3 S = mat3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
            const float sv[3] = {scale_modifier * sc[0], scale_modifier * sc[1], scale_modifier * sc[2]};
            S.m[0][0] = sv[0]; S.m[1][1] = sv[1]; S.m[2][2] = sv[2];
            Mat3 Mm = mul(S, R);
            Mat3 dSig = mat3_cols(dcov[0], 0.5f * dcov[1], 0.5f * dcov[2], 0.5f * dcov[1], dcov[3], 0.5f * dcov[4],
                                  0.5f * dcov[2], 0.5f * dcov[4], dcov[5]);
            Mat3 M2;
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int w = 0; w < 3; w++) M2.m[c][w] = Mm.m[c][w] * 2.0f;
            Mat3 dMt = transpose(mul(M2, dSig));
            Mat3 Rt = transpose(R);
#pragma unroll
            for (int k = 0; k < 3; k++)
                s_scale[k] += Rt.m[k][0] * dMt.m[k][0] + Rt.m[k][1] * dMt.m[k][1] + Rt.m[k][2] * dMt.m[k][2];
#pragma unroll
            for (int k = 0; k < 3; k++)
#pragma unroll
                for (int w = 0; w < 3; w++) dMt.m[k][w] *= sv[k];
#define Q(i, j) dMt.m[i][j]
            s_rot[0] += 2 * z * (Q(0, 1) - Q(1, 0)) + 2 * y * (Q(2, 0) - Q(0, 2)) + 2 * x * (Q(1, 2) - Q(2, 1));
            s_rot[1] += 2 * y * (Q(1, 0) + Q(0, 1)) + 2 * z * (Q(2, 0) + Q(0, 2)) + 2 * r * (Q(1, 2) - Q(2, 1)) - 4 * x * (Q(2, 2) + Q(1, 1));
            s_rot[2] += 2 * x * (Q(1, 0) + Q(0, 1)) + 2 * r * (Q(2, 0) - Q(0, 2)) + 2 * z * (Q(1, 2) + Q(2, 1)) - 4 * y * (Q(2, 2) + Q(0, 0));
            s_rot[3] += 2 * r * (Q(0, 1) - Q(1, 0)) + 2 * x * (Q(2, 0) + Q(0, 2)) + 2 * y * (Q(1, 2) + Q(2, 1)) - 4 * z * (Q(1, 1) + Q(0, 0));
#undef Q
        }
    }
    if (shs && sh_first)
        for (int i = 0; i < 3 * M; i++) dL_dsh[(size_t)g * M * 3 + i] = 0.f;
    dL_dmean3D[3 * g] = s_mean[0]; dL_dmean3D[3 * g + 1] = s_mean[1]; dL_dmean3D[3 * g + 2] = s_mean[2];
    dL_dopacity[g] = s_opac;
    if (dL_dcolor) { dL_dcolor[3 * g] = s_col[0]; dL_dcolor[3 * g + 1] = s_col[1]; dL_dcolor[3 * g + 2] = s_col[2]; }
    if (dL_dcov3D)
#pragma unroll
        for (int i = 0; i < 6; i++) dL_dcov3D[6 * (size_t)g + i] = s_cov[i];
    if (dL_dscale) { dL_dscale[3 * g] = s_scale[0]; dL_dscale[3 * g + 1] = s_scale[1]; dL_dscale[3 * g + 2] = s_scale[2]; }
    if (dL_drot) {
        dL_drot[4 * g] = s_rot[0]; dL_drot[4 * g + 1] = s_rot[1]; dL_drot[4 * g + 2] = s_rot[2];
        dL_drot[4 * g + 3] = s_rot[3];
    }
    (void)has_colors_precomp;
}

}  // namespace

void launch_preprocess(hipStream_t s, int P, int D, int M, const float* means3D, const float* scales,
                       float scale_modifier, const float* rotations, const float* opacities, const float* shs,
                       const float* cov3D_precomp, const float* colors_precomp, const float* viewmatrix,
                       const float* projmatrix, const float* cam_pos, int W, int H, const ViewScalars& vs,
                       int* radii, GeomState g, int tiles_x, int tiles_y, bool prefiltered)
{
    const size_t VP = (size_t)vs.V * P;
    const uint32_t nblk = (uint32_t)((VP + kGaussBlock - 1) / kGaussBlock);
    hipLaunchKernelGGL(preprocess_kernel, dim3(nblk), dim3(kGaussBlock), 0, s, P, D, M, means3D, scales,
                       scale_modifier, rotations, opacities, shs, cov3D_precomp, colors_precomp, viewmatrix,
                       projmatrix, cam_pos, W, H, vs, radii, g, (uint32_t)tiles_x, (uint32_t)tiles_y, prefiltered);
}

void launch_mark_visible(hipStream_t s, int P, const float* means3D, const float* viewmatrix, uint8_t* present)
{
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, viewmatrix, present);
}

void launch_preprocess_backward(hipStream_t s, int P, int D, int M, int V, const float* means3D, const int* radii,
                                const float* shs, const uint8_t* clamped, const float* scales,
                                const float* rotations, float scale_modifier, const float* cov3D,
                                size_t cov3D_view_stride, const float* viewmatrix, const float* projmatrix,
                                const float* campos, const ViewScalars& vs, const float* rows, const uint4* rowpos,
                                const uint32_t* point_offsets, const uint32_t* tiles_touched, float* acc,
                                bool colors_precomp,
                                float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                                float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                                float* dL_dscale, float* dL_drot, float* /*unused*/)
{
    const uint32_t VP = (uint32_t)V * (uint32_t)P;
    hipLaunchKernelGGL(instance_sum_kernel, dim3((4u * VP + 255u) / 256u), dim3(256), 0, s, VP, radii, point_offsets,
                       tiles_touched, rows, rowpos, acc);
    hipLaunchKernelGGL(preprocess_backward_kernel, dim3((P + kGaussBlock - 1) / kGaussBlock), dim3(kGaussBlock), 0,
                       s, P, D, M, means3D, radii, shs, clamped, scales, rotations, scale_modifier, cov3D,
                       cov3D_view_stride, viewmatrix, projmatrix, campos, vs, acc, colors_precomp, dL_dmean2D,
                       dL_dconic, dL_dopacity, dL_dcolor, dL_ddepth, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale,
                       dL_drot);
}

}  // namespace gd
