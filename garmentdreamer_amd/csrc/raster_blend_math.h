// raster_blend_math.h -- the operation-by-operation arithmetic of the alpha blend that the forward blend
// (raster_render.hip) and the per-Gaussian preprocessing (raster_preprocess.hip) must agree on bit for bit with
// oracle/gd_oracle.c: separately rounded fp32 operations, the defined exponential, and the exact contribution threshold.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace gd {
namespace {

// Single, separately rounded fp32 operations.  (HIP's __fmul_rn / __fadd_rn are ordinary inline functions compiled
// with the default fast contraction: after inlining the compiler still fuses them into FMAs -- measured: 7 % of the
// pixels of a forward pass differed from the oracle by one ulp.  Operators written under `fp contract(off)` carry no
// contraction licence.)
__device__ __forceinline__ float mul_rn(float a, float b)
{
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b)
{
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b)
{
#pragma clang fp contract(off)
    return a - b;
}

// exp of the blend, defined operation by operation (oracle/gd_oracle.c gd_expf: Cody-Waite reduction + Cephes degree-5
// polynomial, every step one correctly rounded fp32 operation): the forward pass then reproduces the oracle BIT FOR BIT
// -- the blended pairs, n_contrib, the pixels and the alpha image whose complement is the backward pass's T_final.
__device__ __forceinline__ float gd_expf(float x)
{
#pragma clang fp contract(off)
    if (x < -87.0f) return 0.0f;
    const float n = rintf(mul_rn(x, 1.44269504088896341f));
    float r = __fmaf_rn(n, -0.693359375f, x);
    r = __fmaf_rn(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __fmaf_rn(p, r, 1.3981999507e-3f);
    p = __fmaf_rn(p, r, 8.3334519073e-3f);
    p = __fmaf_rn(p, r, 4.1665795894e-2f);
    p = __fmaf_rn(p, r, 1.6666665459e-1f);
    p = __fmaf_rn(p, r, 5.0000001201e-1f);
    p = __fmaf_rn(p, mul_rn(r, r), r);
    p = add_rn(p, 1.0f);
    return ldexpf(p, (int)n);
}

// Smallest fp32 exponent p with  !(min(0.99, o * expf(p)) < 1/255): `power >= thr` is then the forward pass's
// contribution test itself (forward.cu:346-348), decided without evaluating the exponential per pair.
// It depends on the opacity alone, so it is evaluated once per (view, Gaussian) by preprocess_kernel (GeomState::alpha_thr)
// -- round 3 evaluated it per STAGED (tile, entry): 3.49 M times instead of 0.8 M at the benchmark size.
__device__ __forceinline__ float alpha_threshold_exact(const float o)
{
#pragma clang fp contract(off)
    const float k = 1.0f / 255.0f;
    float t = -logf(255.0f * o);
    if (!(o > 0.0f) || !isfinite(t)) return INFINITY;     // opacity 0 (or NaN): nothing ever contributes
#pragma unroll 1
    for (int it = 0; it < 8; it++) {      // walk down while the next lower exponent still passes
        const float d = nextafterf(t, -INFINITY);
        if (fminf(0.99f, mul_rn(o, gd_expf(d))) < k) break;
        t = d;
    }
#pragma unroll 1
    for (int it = 0; it < 16; it++) {     // walk up while this exponent fails
        if (!(fminf(0.99f, mul_rn(o, gd_expf(t))) < k)) break;
        t = nextafterf(t, INFINITY);
    }
    return t;
}


}  // namespace
}  // namespace gd
