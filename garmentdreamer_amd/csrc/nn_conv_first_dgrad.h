// nn_conv_first_dgrad.h -- input gradient of the VAE encoder's FIRST convolution (Cin <= 3 image channels, Cout = 128), included by
// nn_conv3x3.hip.  (autograd of `conv_in`: diffusers AutoencoderKL.encoder.conv_in, reached from
// Garment_3DGS/threestudio/models/guidance/stable_diffusion_guidance.py:165-166 encode_images with a gradient.)
//
// Until round 5 this ran on the implicit-GEMM kernel with the flipped filter zero-padded to 4 output channels on a 32-channel MFMA
// tile: 305 us per 8-view launch, bound by the NINE reads of every dy element (one per tap) through L2 / LDS -- 4.8 GB for a 537 MB
// tensor.  Here dy is read ONCE ("col2im" order):
//     T[q][(tap, c)] = sum_k dy[q][k] w[k][c][tap]        a 1x1 product, 128 -> 27 (9 taps x 3 channels: ONE 32-row MFMA block)
//     dx[p][c]       = sum_tap T[p - (ky - 1, kx - 1)][(tap, c)]
// A workgroup owns a 16x16 pixel tile: T of its 18x18 halo (eleven 32-pixel MFMA blocks, eight v_mfma_f32_32x32x16_bf16 each, the
// filter's A fragments resident in registers, the dy fragments straight from global memory -- each 256-byte pixel row is consumed
// whole by one wave within eight back-to-back loads) goes to LDS in fp32, then every thread sums the nine shifted entries of its
// pixel in fp32: one rounding to bf16 (the implicit-GEMM form rounded nothing either; same accuracy class, fewer terms per sum).
// HBM: 1.27x dy (halo) + 8 B per pixel out.

namespace {

constexpr int kFdPix = 16, kFdHalo = 18, kFdRow = 36;     // T row: 32 floats + 4 pad (144 B: conflict-free b128 writes)
constexpr int kFdBlocks = (kFdHalo * kFdHalo + 31) / 32;  // 11

// wp[m][k], m = 3 * tap + c (rows >= 9 * Cin are zero), k = output channel of the forward convolution:  w[k][tap][c] (KHWC)
__global__ void conv3x3_first_dgrad_weights_kernel(const uint16_t* __restrict__ w, uint16_t* __restrict__ wp, int Cin)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;     // 32 * 128
    if (i >= 32 * 128) return;
    const int m = i >> 7, k = i & 127;
    const int tap = m / 3, c = m - 3 * tap;
    wp[i] = (m < 27 && c < Cin) ? w[((size_t)k * 9 + tap) * Cin + c] : (uint16_t)0;
}

__global__ __launch_bounds__(256) void conv3x3_first_dgrad_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ wp,
                                                                  uint16_t* __restrict__ dx4, int H, int W, int tiles_x, int tiles_y)
{
    __shared__ __attribute__((aligned(16))) float sT[kFdBlocks * 32][kFdRow];      // 50.7 KB: three workgroups per CU
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fn = lane & 31, fh = lane >> 5;
    // XCD-aware order: workgroup b runs on XCD b % 8; neighbouring tiles (which share halo rows) stay on one XCD
    int bid = blockIdx.x;
    const int nwg = gridDim.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int tpi = tiles_x * tiles_y;
    const int n = bid / tpi, t = bid - n * tpi;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int y0 = ty * kFdPix - 1, x0 = tx * kFdPix - 1;

    bf16x8_t wf[8];                      // A fragments: row fn of wp, k = 16 kk + 8 fh .. + 7
#pragma unroll
    for (int kk = 0; kk < 8; kk++) wf[kk] = *(const bf16x8_t*)(wp + fn * 128 + 16 * kk + 8 * fh);

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + (size_t)n * H * W * 128), 0,
                                                                        (int)((uint32_t)H * (uint32_t)W * 256u), 0x00020000);
    for (int b = wave; b < kFdBlocks; b += 4) {
        const int hp = 32 * b + fn;
        const int hy = hp / kFdHalo, hx = hp - hy * kFdHalo;
        const int y = y0 + hy, x = x0 + hx;
        const bool in = hp < kFdHalo * kFdHalo && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        const uint32_t off = in ? ((uint32_t)y * (uint32_t)W + (uint32_t)x) * 256u + 16u * fh : kOOB;   // outside: reads 0
        bf16x8_t v[8];
#pragma unroll
        for (int kk = 0; kk < 8; kk++)
            v[kk] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 32 * kk, 0));
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 8; kk++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk], v[kk], acc, 0, 0, 0);
        // acc[r] = T[pixel fn][row (r & 3) + 8 (r >> 2) + 4 fh]: four consecutive rows per 16-byte store
#pragma unroll
        for (int g = 0; g < 4; g++)
            *(float4*)&sT[hp][8 * g + 4 * fh] = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    }
    __syncthreads();
    const int oy = tid >> 4, ox = tid & 15;
    const int y = ty * kFdPix + oy, x = tx * kFdPix + ox;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ky++)
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            // forward: y[q] += x[q + (ky - 1, kx - 1)] w[ky][kx]  =>  dx[p] += T[p - (ky - 1, kx - 1)][(ky, kx)]; halo origin = tile - 1
            const float* tr = &sT[(oy + 2 - ky) * kFdHalo + (ox + 2 - kx)][3 * (3 * ky + kx)];
            s0 += tr[0];
            s1 += tr[1];
            s2 += tr[2];
        }
    if (y < H && x < W) {
        uint2 o;
        o.x = pack_bf16(s0, s1);
        o.y = pack_bf16(s2, 0.f);
        *(uint2*)(dx4 + (((size_t)n * H + y) * W + x) * 4) = o;
    }
}

}  // namespace
