// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3): operand lane layout and scale semantics.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_scale_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <hip/hip_fp8.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void probe(const uint8_t* A, const uint8_t* B, float* D, int scale_a, int scale_b)
{
    // hypothesis: lane l holds row (l & 31), k = 32 * (l >> 5) ... + 31, 32 consecutive bytes
    const int lane = threadIdx.x;
    const int row = lane & 31, kb = lane >> 5;
    v8i a, b;
    const int* pa = (const int*)(A + row * 64 + kb * 32);
    const int* pb = (const int*)(B + row * 64 + kb * 32);   // B stored [n][k]
    for (int i = 0; i < 8; i++) { a[i] = pa[i]; b[i] = pb[i]; }
    f32x16 c;
    for (int i = 0; i < 16; i++) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
    // C/D: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    for (int r = 0; r < 16; r++) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), j = lane & 31;
        D[i * 32 + j] = c[r];
    }
}

static uint8_t f2e4m3(float f) { __hip_fp8_e4m3 v(f); return *(uint8_t*)&v; }
static float e4m32f(uint8_t b) { __hip_fp8_e4m3 v; *(uint8_t*)&v = b; return (float)v; }

int main()
{
    std::vector<uint8_t> A(32 * 64), B(32 * 64);
    std::vector<float> Af(32 * 64), Bf(32 * 64);
    srand(1);
    for (int i = 0; i < 32 * 64; i++) {
        float x = (rand() % 17 - 8) * 0.25f, y = (rand() % 13 - 6) * 0.5f;
        A[i] = f2e4m3(x); B[i] = f2e4m3(y); Af[i] = e4m32f(A[i]); Bf[i] = e4m32f(B[i]);
    }
    uint8_t *dA, *dB; float* dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 32 * 32 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    for (int t = 0; t < 4; t++) {
        int sa = t == 0 ? 0x7f7f7f7f : (t == 1 ? 0x7f : (t == 2 ? 0x80808080 : 0));
        int sb = t == 3 ? 0 : 0x7f7f7f7f;
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, sa, sb);
        std::vector<float> D(32 * 32);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        double maxerr = 0, ratio = 0; int cnt = 0;
        for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) {
            double ref = 0;
            for (int k = 0; k < 64; k++) ref += (double)Af[i * 64 + k] * Bf[j * 64 + k];
            maxerr = fmax(maxerr, fabs(ref - D[i * 32 + j]));
            if (fabs(ref) > 1.0) { ratio += D[i * 32 + j] / ref; cnt++; }
        }
        printf("scale_a=%08x scale_b=%08x: max |D - A B^T| = %g, mean D/ref = %g\n", sa, sb, maxerr, ratio / (cnt ? cnt : 1));
    }
    return 0;
}
