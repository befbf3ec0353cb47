// How fast can ONE workgroup per CU (512 threads) stream an L2-resident region into the CU -- the filter-slice traffic of
// the patch-staged / Winograd convolution kernels (every workgroup re-reads the same ~400 KB per tile)?
//   mode 0: buffer_load_dwordx4 ... lds (LDS-DMA), 1 KiB per wave-instruction
//   mode 1: buffer_load_dwordx4 into registers (consumed by a dummy add)
//   mode 2: as 0, but every workgroup reads its OWN region (no sharing between CUs; L2 / MALL resident)
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/l2probe tools/probes/l2_stream_probe.hip ; run: /tmp/l2probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ src, size_t region, int iters, uint32_t* out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6;
    const char* base = src + (MODE == 2 ? (size_t)blockIdx.x * region : 0);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)region, 0x00020000);
    const int nslice = (int)(region / 32768);
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        for (int sl = 0; sl < nslice; sl++) {
            // one 32 KiB slice per "step": 4 pieces per thread
            if (MODE == 1) {
                u32x4 v[4];
#pragma unroll
                for (int i = 0; i < 4; i++)
                    v[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16 + 8192 * i, sl * 32768, 0));
#pragma unroll
                for (int i = 0; i < 4; i++) acc += v[i];
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + ((sl % DEPTH) * 32768) + (wave * 64 + 512 * i) * 16),
                                                             16, tid * 16 + 8192 * i, sl * 32768, 0, 0);
                if (DEPTH == 1 || (sl % DEPTH) == DEPTH - 1) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
            }
        }
    }
    if (MODE != 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc[0] = *(uint32_t*)(smem + tid * 4);
    }
    out[blockIdx.x * 512 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int MODE, int DEPTH>
static void run(const char* name, const char* src, size_t region, int grid, uint32_t* out)
{
    const int iters = 40;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void*)probe<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    probe<MODE, DEPTH><<<grid, 512, 160 * 1024>>>(src, region, 2, out);
    hipEventRecord(a);
    probe<MODE, DEPTH><<<grid, 512, 160 * 1024>>>(src, region, iters, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)grid * iters * (double)(region / 32768 * 32768);
    printf("%-44s grid %3d region %4zu KB: %7.1f GB/s per CU, %6.2f TB/s chip (%s)\n", name, grid, region >> 10,
           bytes / grid / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
}

int main()
{
    const size_t region = 384 * 1024;
    char* src; uint32_t* out;
    hipMalloc(&src, region * 256);
    hipMemset(src, 1, region * 256);
    hipMalloc(&out, 256 * 512 * 4);
    for (int grid : {256, 64, 8}) {
        run<0, 1>("LDS-DMA, shared region, wait every slice", src, region, grid, out);
        run<0, 2>("LDS-DMA, shared region, wait every 2 slices", src, region, grid, out);
        run<0, 4>("LDS-DMA, shared region, wait every 4 slices", src, region, grid, out);
        run<1, 1>("global->VGPR, shared region", src, region, grid, out);
        run<2, 4>("LDS-DMA, private regions, wait every 4", src, region, grid, out);
    }
    return 0;
}
