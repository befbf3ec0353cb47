// nn_conv_stream.h -- 3x3 / stride 1 / pad 1 convolution for maps of a FEW HUNDRED pixels (the UNet's 8^2 and 16^2 levels at one
// or two latents: M = N*H*W <= 512 GEMM rows against 640 ... 2560 input and 1280 output channels).  Included by nn_conv3x3.hip
// inside its anonymous namespace.
//
// WHY.  Such a layer is its filter bank: 1280 -> 1280 is 29.5 MB of weights against 0.3-1.3 MB of activations and 4-15 GFLOP,
// i.e. 4-7 us of HBM time and 2-6 us of matrix-pipe time.  The implicit-GEMM kernel's split-K form needs ~400 workgroups of
// 128 x 128 tiles to fill the chip, so at M = 128 it cuts K into 23 ranges: every range re-stages the same pixels through LDS,
// leaves a 64 KB fp32 partial tile, and a second launch adds them up (15.7 MB written and read back -- half the filter
// traffic again): 20 + 7 us per layer (profiles/r05_small_tile_sweep.txt), 15 + 13 such layers per UNet forward.
//
// HOW.  The filter bank is the only stream that matters, so it goes HBM -> REGISTERS, once, fully coalesced, with no LDS in
// between: conv3x3_stream_weights_kernel re-packs the frozen weights into MFMA A-fragment order -- for a 32-channel block cb
// and K step s (16 input channels of one tap) the 64 lanes' 16-byte fragments are 1 KB of consecutive memory -- and a wave
// owns FA channel blocks x ALL of its M-tile's pixels (FB blocks of 32) x a contiguous range of K steps.  The activations are
// tiny and L2-resident: their B fragments (lane = pixel, 8 channels) are read straight from global memory as well, 16 bytes
// per lane, the halo as hardware zero fill (out-of-range buffer offset), one step ahead of the MFMAs.  A workgroup = 8 waves =
// 8 consecutive K ranges of the SAME output tile: their accumulators are added pairwise through LDS (three rounds), so only
// every eighth K range leaves an fp32 partial tile, in fragment order; conv_stream_reduce_kernel adds the few that remain,
// with bias / residual, and rounds once.  No atomics; the summation order is fixed by the launch geometry (bit-reproducible).
#pragma once

constexpr int kStreamWaves = 8;

// w [Cout][3][3][Cin] bf16 -> wp [Cout/32][9 * Cin/16][64 lanes] x 16 bytes: lane l of (cb, s) holds
// w[cb * 32 + (l & 31)][tap = s / (Cin/16)][16 * (s % (Cin/16)) + 8 * (l >> 5) .. + 7].  Needs Cout % 32 == 0, Cin % 16 == 0.
__global__ __launch_bounds__(256) void conv3x3_stream_weights_kernel(const uint16_t* __restrict__ w, uint4* __restrict__ wp,
                                                                     int Cout, int Cin)
{
    const int kc = Cin / 16, steps = 9 * kc;
    const size_t total = (size_t)(Cout / 32) * steps * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(i & 63);
        const size_t cs = i >> 6;
        const int s = (int)(cs % steps), cb = (int)(cs / steps);
        const int tap = s / kc, c0 = (s - tap * kc) * 16 + 8 * (l >> 5);
        wp[i] = *(const uint4*)(w + ((size_t)(cb * 32 + (l & 31)) * 9 + tap) * Cin + c0);
    }
}

// One workgroup = 8 waves = 8 consecutive K ranges of output tile (channel tile blockIdx.x of 32 FA channels, pixel tile blockIdx.z
// of 32 FB pixels); blockIdx.y = which group of 8 ranges.  partial: [gridDim.y][gridDim.z][gridDim.x] tiles of FA * FB * 1024 floats
// in fragment order (((a * FB + b) * 4 + q) * 64 + lane) * 4.
template <int FA, int FB>
__global__ __launch_bounds__(64 * kStreamWaves) void conv3x3_stream_kernel(
    const uint16_t* __restrict__ in, const uint4* __restrict__ wp, float* __restrict__ partial, int Nimg, int H, int W, int Cin,
    int steps_total, int steps_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kTileFloats = FA * FB * 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fn = lane & 31, fk = lane >> 5;
    const int HW = H * W, M = Nimg * HW;
    const int kc = Cin / 16;
    const int range = (int)blockIdx.y * kStreamWaves + wave;
    const int s0 = range * steps_per_wave, s1 = min(steps_total, s0 + steps_per_wave);

    // activations as a buffer: 32-bit offsets, an offset beyond the tensor reads zeros (halo and ragged last pixel block)
    const uint32_t row_bytes = (uint32_t)Cin * 2u;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)((uint32_t)M * row_bytes), 0x00020000);
    int py[FB], px[FB];
    uint32_t pbase[FB];
#pragma unroll
    for (int b = 0; b < FB; b++) {
        const int m = ((int)blockIdx.z * FB + b) * 32 + fn;
        if (m < M) {
            const int n = m / HW, rem = m - n * HW;
            py[b] = rem / W;
            px[b] = rem - py[b] * W;
            pbase[b] = (uint32_t)m * row_bytes + (uint32_t)fk * 16u;
        } else {
            py[b] = -100000; px[b] = 0; pbase[b] = kOOB;
        }
    }
    const uint4* wq[FA];
#pragma unroll
    for (int a = 0; a < FA; a++) wq[a] = wp + ((size_t)((int)blockIdx.x * FA + a) * steps_total) * 64 + lane;

    f32x16 acc[FA][FB];
#pragma unroll
    for (int a = 0; a < FA; a++)
#pragma unroll
        for (int b = 0; b < FB; b++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc[a][b][k] = 0.f;

    // loader state: the tap of the step being fetched; per-pixel offsets are re-made when the tap changes (every Cin/16 steps)
    int ld_tap = -1;
    uint32_t voff[FB];
    auto fetch = [&](int s, bf16x8_t (&A)[FA], bf16x8_t (&B)[FB]) {
        const int tap = s / kc, cs = s - tap * kc;
        if (tap != ld_tap) {            // (wave-uniform)
            ld_tap = tap;
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const int shift = (dy * W + dx) * (int)row_bytes;
#pragma unroll
            for (int b = 0; b < FB; b++) {
                const bool ok = (unsigned)(py[b] + dy) < (unsigned)H && (unsigned)(px[b] + dx) < (unsigned)W;
                voff[b] = ok ? (uint32_t)((int)pbase[b] + shift) : kOOB;
            }
        }
#pragma unroll
        for (int a = 0; a < FA; a++) A[a] = __builtin_bit_cast(bf16x8_t, wq[a][(size_t)s * 64]);
#pragma unroll
        for (int b = 0; b < FB; b++)
            B[b] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)voff[b], cs * 32, 0));
    };
    auto mma = [&](const bf16x8_t (&A)[FA], const bf16x8_t (&B)[FB]) {
#pragma unroll
        for (int a = 0; a < FA; a++)
#pragma unroll
            for (int b = 0; b < FB; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a], B[b], acc[a][b], 0, 0, 0);
    };

    // two register sets, the next step's fragments in flight under the current step's MFMAs
    bf16x8_t A0[FA], B0[FB], A1[FA], B1[FB];
    if (s0 < s1) {
        fetch(s0, A0, B0);
        int s = s0;
        for (; s + 2 <= s1; s += 2) {
            fetch(s + 1, A1, B1);
            mma(A0, B0);
            if (s + 2 < s1) fetch(s + 2, A0, B0);
            mma(A1, B1);
        }
        if (s < s1) mma(A0, B0);
    }

    // ---- the 8 waves' accumulators added pairwise through LDS: waves [h, 2h) hand theirs to waves [0, h), h = 4, 2, 1
#pragma unroll
    for (int h = kStreamWaves / 2; h >= 1; h >>= 1) {
        if (wave >= h && wave < 2 * h) {
            float* dst = (float*)smem + (size_t)(wave - h) * kTileFloats;
#pragma unroll
            for (int a = 0; a < FA; a++)
#pragma unroll
                for (int b = 0; b < FB; b++)
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        *(float4*)(dst + (((a * FB + b) * 4 + q) * 64 + lane) * 4) =
                            make_float4(acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
        }
        __syncthreads();
        if (wave < h) {
            const float* src = (const float*)smem + (size_t)wave * kTileFloats;
#pragma unroll
            for (int a = 0; a < FA; a++)
#pragma unroll
                for (int b = 0; b < FB; b++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float4 t = *(const float4*)(src + (((a * FB + b) * 4 + q) * 64 + lane) * 4);
                        acc[a][b][4 * q] += t.x; acc[a][b][4 * q + 1] += t.y; acc[a][b][4 * q + 2] += t.z; acc[a][b][4 * q + 3] += t.w;
                    }
        }
        __syncthreads();
    }
    if (wave == 0) {
        float* dst = partial + (((size_t)blockIdx.y * gridDim.z + blockIdx.z) * gridDim.x + blockIdx.x) * kTileFloats;
#pragma unroll
        for (int a = 0; a < FA; a++)
#pragma unroll
            for (int b = 0; b < FB; b++)
#pragma unroll
                for (int q = 0; q < 4; q++)
                    *(float4*)(dst + (((a * FB + b) * 4 + q) * 64 + lane) * 4) =
                        make_float4(acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
    }
}

// out = bf16( sum_g partial[g] + bias + residual ): one thread = 4 consecutive channels of one pixel (one float4 of a tile image).
template <int FA, int FB>
__global__ __launch_bounds__(256) void conv_stream_reduce_kernel(const float* __restrict__ partial, int G, int tiles_n, int tiles_m,
                                                                 int M, int HW, int Cout, const uint16_t* __restrict__ bias,
                                                                 int bias_img_stride, const uint16_t* __restrict__ residual,
                                                                 uint16_t* __restrict__ out)
{
    constexpr int kQuads = FA * FB * 256;                   // float4 per tile
    const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int tile = i / kQuads, r = i - tile * kQuads;
    if (tile >= tiles_n * tiles_m) return;
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    const int lane = r & 63, q = (r >> 6) & 3, ab = r >> 8, b = ab % FB, a = ab / FB;
    const int m = (tm * FB + b) * 32 + (lane & 31);
    const int co = (tn * FA + a) * 32 + 8 * q + 4 * (lane >> 5);
    if (m >= M || co >= Cout) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* p = (const float4*)partial + (size_t)tile * kQuads + r;
    const size_t gstride = (size_t)tiles_n * tiles_m * kQuads;
    for (int g = 0; g < G; g++) {
        const float4 t = p[(size_t)g * gstride];
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const size_t o = (size_t)m * Cout + co;
    if (bias) {
        const uint2 bb = *(const uint2*)(bias + (size_t)(m / HW) * bias_img_stride + co);
        v.x += bf2f((uint16_t)(bb.x & 0xffff)); v.y += bf2f((uint16_t)(bb.x >> 16));
        v.z += bf2f((uint16_t)(bb.y & 0xffff)); v.w += bf2f((uint16_t)(bb.y >> 16));
    }
    if (residual) {
        const uint2 rr = *(const uint2*)(residual + o);
        v.x += bf2f((uint16_t)(rr.x & 0xffff)); v.y += bf2f((uint16_t)(rr.x >> 16));
        v.z += bf2f((uint16_t)(rr.y & 0xffff)); v.w += bf2f((uint16_t)(rr.y >> 16));
    }
    uint2 ov;
    ov.x = pack_bf16(v.x, v.y);
    ov.y = pack_bf16(v.z, v.w);
    *(uint2*)(out + o) = ov;
}
