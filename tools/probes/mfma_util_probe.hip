// MFMA-utilisation normalisation probe for gfx950: a kernel that issues NOTHING but v_mfma_f32_32x32x16_bf16 (four
// independent accumulators per wave, `waves_per_simd` waves on every SIMD of the chip), run under
//   rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- /tmp/mfma_util_probe
// so that the normalisation bench.py / tools/pmc_all.sh use for `mfma_util` can be CHECKED on a stream whose utilisation
// is 1 by construction:   mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDS * GRBM_GUI_ACTIVE / XCDS_SUMMED).
// Prints the launched MFMA count (32 busy cycles each per the microarchitecture guide), the wall time and the implied
// clock, for the cross-check   SQ_VALU_MFMA_BUSY_CYCLES == 32 * n_mfma.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_util_probe tools/probes/mfma_util_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// FILL = 0: MFMA only.  FILL = 1: half of the MFMAs replaced by v_fma_f32 filler of roughly the same issue time (a stream whose
// utilisation is ~0.5 by construction -- checks that the metric is linear, not saturating).
template <int FILL>
__global__ __launch_bounds__(256) void mfma_only(float* out, int iters)
{
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x ^ i)); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    for (int it = 0; it < iters; it++) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        if (FILL == 0) {
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
        } else {
            // 2 x 32 cycles of VALU instead: 16 dependent-free v_fma_f32 at ~4.5 cycles each
            asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %0\n"
                         "v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %0\n"
                         "v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %0\n"
                         "v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %0\n"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
        }
    }
    float s = f0 + f1 + f2 + f3;
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char** argv)
{
    const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 2;
    const int iters = argc > 2 ? atoi(argv[2]) : 200000;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int blocks = cus * waves_per_simd;      // 256 threads = 4 waves = one per SIMD of a CU
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int fill = 0; fill < 2; fill++) {
        for (int rep = 0; rep < 2; rep++) {       // rep 0 = warm-up
            hipEventRecord(e0, 0);
            if (fill == 0) hipLaunchKernelGGL(mfma_only<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(mfma_only<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep == 0) continue;
            const double n_mfma = (double)blocks * 4 * iters * (fill ? 2 : 4);
            const double simd_busy = 32.0 * n_mfma;                   // cycles the matrix pipes are busy, all SIMDs
            const double simds = 4.0 * cus;
            printf("%s: CUs %d, %d wave(s)/SIMD, %.0f MFMAs (32x32x16 bf16) -> %.4g MFMA-busy SIMD-cycles expected; %.3f ms; "
                   "%.1f TFLOP/s; clock if the pipes never idle %.3f GHz; expected util %s\n",
                   fill ? "mfma_only<1> (half VALU filler)" : "mfma_only<0>", cus, waves_per_simd, n_mfma, simd_busy, ms,
                   n_mfma * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12, simd_busy / simds / (ms * 1e-3) / 1e9,
                   fill ? "~0.5" : "1.0");
        }
    }
    return 0;
}
