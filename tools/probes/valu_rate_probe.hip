// VALU issue-rate probe for gfx950: how many cycles does a wave64 v_fma_f32 / v_pk_fma_f32 / DPP / transcendental
// instruction occupy a SIMD?  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_probe tools/probes/valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, int iters, long long* cyc)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
    f2 pb = {b, b}, pc = {c, c};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 1) {
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));
        } else if (KIND == 2) {
            asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 3) {
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 4) {
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 5) {   // dependent chain of DPP adds on ONE register (latency + hazard)
            asm volatile(REP8("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n") : "+v"(a0));
        } else if (KIND == 6) {   // mixed: v_mov_dpp then v_fma (unfused broadcast)
            asm volatile("v_mov_b32_dpp %0, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fma_f32 %1, %1, %8, %9\n"
                         "v_mov_b32_dpp %2, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_mov_b32_dpp %4, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fma_f32 %5, %5, %8, %9\n"
                         "v_mov_b32_dpp %6, %8 row_newbcast:9 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 7) {   // v_cndmask with vcc
            asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");

        } else if (KIND == 9) {
            asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[10:11]\n v_cndmask_b32_e64 %1, %1, %8, s[10:11]\n v_cndmask_b32_e64 %2, %2, %8, s[10:11]\n v_cndmask_b32_e64 %3, %3, %8, s[10:11]\n"
                         "v_cndmask_b32_e64 %4, %4, %8, s[10:11]\n v_cndmask_b32_e64 %5, %5, %8, s[10:11]\n v_cndmask_b32_e64 %6, %6, %8, s[10:11]\n v_cndmask_b32_e64 %7, %7, %8, s[10:11]\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s10", "s11");
        } else if (KIND == 10) {
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));
        } else if (KIND == 11) {
            asm volatile("v_min_f32 %0, %0, %8\n v_bfe_i32 %1, %1, 3, 1\n v_and_b32 %2, %2, %8\n v_min_f32 %3, %3, %8\n"
                         "v_bfe_i32 %4, %4, 5, 1\n v_and_b32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_bfe_i32 %7, %7, 7, 1\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if (KIND == 12) {
            asm volatile("v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                         "v_mul_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %4, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_fmac_f32_dpp %6, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %7, %8, %9 row_newbcast:9 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 13) {   // compiler-generated select chain
            a0 = (a1 > c) ? a0 * b : a0; a1 = (a2 > c) ? a1 * b : a1; a2 = (a3 > c) ? a2 * b : a2; a3 = (a0 > c) ? a3 * b : a3;
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (KIND == 8) {   // LDS broadcast read b128 (all lanes of a 16-lane row read the same address)
            extern __shared__ float4 sm[];
            float4 r0 = sm[(threadIdx.x >> 4) + (i & 7)], r1 = sm[64 + (threadIdx.x >> 4) + (i & 7)];
            a0 += r0.x + r0.y + r0.z + r0.w; a1 += r1.x + r1.y + r1.z + r1.w;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
void run(const char* name, int waves_per_simd, float* out, long long* cyc)
{
    const int iters = 20000;
    // 256 CUs x 4 SIMDs x waves_per_simd waves = workgroups of 256 threads (4 waves, one per SIMD)
    const int grid = 256 * waves_per_simd;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(256), 4096, 0, out, 100, cyc);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(256), 4096, 0, out, iters, cyc);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    long long hc = 0;
    hipMemcpy(&hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
    const double instr_per_wave = (KIND == 8 ? 2.0 : KIND == 13 ? 12.0 : 8.0) * iters;
    // wall-clock ns per wave-instruction per SIMD
    printf("%-28s waves/SIMD %d: %8.3f ms  -> %.3f ns per wave-instr per SIMD (%.2f cycles @2.4GHz); readcyclecounter ticks/instr (one wave) %.2f\n",
           name, waves_per_simd, ms, ms * 1e6 / (instr_per_wave * waves_per_simd), ms * 1e6 / (instr_per_wave * waves_per_simd) * 2.4,
           (double)hc / instr_per_wave);
}

int main()
{
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 16 * 256 * sizeof(float));
    hipMalloc(&cyc, 8);
    for (int w : {4, 8}) {
        run<0>("v_fma_f32", w, out, cyc);
        run<1>("v_pk_fma_f32", w, out, cyc);
        run<2>("v_add_f32_dpp row_shr", w, out, cyc);
        run<3>("v_exp_f32", w, out, cyc);
        run<4>("v_rcp_f32", w, out, cyc);
        run<5>("dependent dpp chain+nop", w, out, cyc);
        run<6>("mov_dpp newbcast + fma", w, out, cyc);
        run<7>("v_cndmask_b32", w, out, cyc);
        run<8>("ds_read_b128 row-bcast x2", w, out, cyc);
        run<9>("v_cndmask_b32_e64 sgpr", w, out, cyc);
        run<10>("v_pk_mul/add_f32", w, out, cyc);
        run<11>("v_min/bfe_i32/and", w, out, cyc);
        run<12>("v_mul_dpp shr + v_fmac_dpp", w, out, cyc);
        run<13>("compiler select (4/iter)", w, out, cyc);
    }
    return 0;
}
