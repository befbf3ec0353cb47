// Does an fp32 MFMA issued between packed-fp32 VALU instructions cost the wave's SIMD a VALU issue slot on gfx950?
// (Question behind VERDICT r03 next-7: accumulate the backward blend's ten per-entry sums on the idle matrix pipe.)
// Each kind runs the same stream of 8 independent v_pk_fma_f32 per iteration, plus M MFMAs of one shape per iteration
// on independent accumulators, at 4 waves per SIMD (the backward kernel's occupancy) and at 1 wave per SIMD.
//   build: hipcc --offload-arch=gfx950 -O3 -o /tmp/coissue tools/probes/mfma_valu_coissue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// SHAPE 0: v_mfma_f32_4x4x1_16b_f32 (2 passes), SHAPE 1: v_mfma_f32_16x16x1_4b_f32 (8 passes); NV = 0 drops the VALU stream
template <int SHAPE, int M, int NV>
__global__ __launch_bounds__(256) void probe(float* out, int iters)
{
    const float t = (float)threadIdx.x;
    f2 p0 = {t, t + 1}, p1 = {t + 2, t + 3}, p2 = {t + 4, t + 5}, p3 = {t + 6, t + 7}, p4 = {t + 1, t + 2}, p5 = {t + 3, t + 4},
       p6 = {t + 5, t + 6}, p7 = {t + 7, t};
    const f2 pb = {1.0001f, 1.0001f}, pc = {0.5f, 0.5f};
    f4 c4[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f16v c16[2];
    for (int k = 0; k < 16; k++) { c16[0][k] = 0; c16[1][k] = 0; }
    float a = t * 0.001f, b = 1.f - a;
    for (int i = 0; i < iters; i++) {
        if (NV) {
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
        }
#pragma unroll
        for (int m = 0; m < M; m++) {
            if (SHAPE == 0) c4[m & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c4[m & 3], 0, 0, 0);
            else c16[m & 1] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, c16[m & 1], 0, 0, 0);
            if (NV && m == M / 2 - 1) {
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));
            }
        }
        if (NV && M < 2) {
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));
        }
        asm volatile("" : "+v"(a));
    }
    float s = p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x;
    for (int m = 0; m < 4; m++) s += c4[m][0] + c4[m][1] + c4[m][2] + c4[m][3];
    for (int k = 0; k < 16; k++) s += c16[0][k] + c16[1][k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, int M, int NV>
void run(const char* name, int waves_per_simd, float* out)
{
    const int iters = 20000;
    const int grid = 256 * waves_per_simd;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((probe<SHAPE, M, NV>), dim3(grid), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((probe<SHAPE, M, NV>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    // ns of SIMD time per iteration and wave
    printf("%-44s waves/SIMD %d: %8.3f ms -> %7.2f ns per (iteration, wave) on its SIMD\n", name, waves_per_simd, ms,
           ms * 1e6 / ((double)iters * waves_per_simd));
}

int main()
{
    float* out;
    hipMalloc(&out, 256 * 16 * 256 * sizeof(float));
    for (int w : {4, 1}) {
        run<0, 0, 1>("8 pk_fma", w, out);
        run<0, 1, 1>("8 pk_fma + 1 mfma 4x4x1 16B", w, out);
        run<0, 2, 1>("8 pk_fma + 2 mfma 4x4x1 16B", w, out);
        run<0, 4, 1>("8 pk_fma + 4 mfma 4x4x1 16B", w, out);
        run<0, 4, 0>("4 mfma 4x4x1 16B alone", w, out);
        run<1, 1, 1>("8 pk_fma + 1 mfma 16x16x1 4B", w, out);
        run<1, 2, 1>("8 pk_fma + 2 mfma 16x16x1 4B", w, out);
        run<1, 2, 0>("2 mfma 16x16x1 4B alone", w, out);
    }
    return 0;
}
