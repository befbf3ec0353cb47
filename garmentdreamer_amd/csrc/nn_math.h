// Packed-fp32 helpers shared by the row passes (nn_elementwise.hip) and the GEMM epilogues that fuse them (nn_linear.hip):
// one definition, so that a fused epilogue rounds exactly like the separate kernel it replaces.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gdnn {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 unpack2(uint32_t w) { return f2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; }
__device__ __forceinline__ uint32_t pack2(f2 v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2_t)); }
__device__ __forceinline__ f2 round_bf16(f2 v) { return unpack2(pack2(v)); }

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, three orders below a bf16 ulp of the GELU): one rcp, one
// exp and a degree-5 Horner instead of libm erff's ~40 instructions -- the kernels are VALU-bound on this function.
// Two values per call: the polynomial runs on v_pk_fma_f32.
__device__ __forceinline__ f2 erf_as2(f2 x)
{
    const f2 ax = {fabsf(x.x), fabsf(x.y)};
    const f2 d = 0.3275911f * ax + 1.0f;
    const f2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    f2 p = 1.061405429f * t + -1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t + -0.284496736f;
    p = p * t + 0.254829592f;
    const f2 a = (ax * ax) * -1.44269504088896341f;
    const f2 e = 1.0f - (p * t) * f2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
    return f2{copysignf(e.x, x.x), copysignf(e.y, x.y)};
}

// diffusers GEGLU on two packed bf16 pairs: hidden * gelu(gate), F.gelu's result rounded to bf16 before the multiply
// (as the eager bf16 ops do), the product rounded once
__device__ __forceinline__ uint32_t geglu2(uint32_t hidden, uint32_t gate)
{
    const f2 gv = unpack2(gate);
    const f2 ge = round_bf16((0.5f * gv) * (1.0f + erf_as2(gv * 0.70710678118654752f)));
    return pack2(unpack2(hidden) * ge);
}

}  // namespace gdnn
