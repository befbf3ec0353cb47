// C entry points of the filter-bank-in-registers experiment (nn_conv_regw.h), included by csrc/nn_conv3x3.hip when a tools/
// build defines GD_NN_EXPERIMENTAL_REGW (tools/regw_variants.sh).  Measured at parity with the wide tile (DESIGN.md 3.11,
// profiles/r04_regw_ablation.txt), so it is NOT part of libgd_nn.so, include/gd_nn.h or nn_ops.
extern "C" {
int gd_nn_conv3x3_regw_supported(int N, int H, int W, int Cin, int Cout);
size_t gd_nn_conv3x3_regw_weights_bytes(void);
int gd_nn_conv3x3_regw_weights(void* stream, const void* weight, void* u);
int gd_nn_conv3x3_regw_forward(void* stream, const void* x, const void* u, const void* bias, int bias_img_stride,
                               const void* residual, void* y, int N, int H, int W, int Cin, int Cout, float* stat_part);
}
// ---- 128 -> 128 channels with the filter bank resident in registers (nn_conv_regw.h)
size_t gd_nn_conv3x3_regw_weights_bytes(void) { return (size_t)4 * 72 * 64 * 16; }

int gd_nn_conv3x3_regw_weights(void* stream, const void* weight, void* u)
{
    if (!weight || !u) return fail(GD_NN_ERR_INVALID_ARG, "regw_weights: null pointer");
    hipLaunchKernelGGL(conv3x3_regw_weights_kernel, dim3(72), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)weight,
                       (uint16_t*)u);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

int gd_nn_conv3x3_regw_supported(int N, int H, int W, int Cin, int Cout)
{
    if (N <= 0 || Cin != 128 || Cout != 128 || H < 16 || W < 32 || (H & 15) || (W & 31)) return 0;
    if ((double)N * H * W * 256.0 >= 2147483648.0) return 0;
    return 1;
}

int gd_nn_conv3x3_regw_forward(void* stream, const void* x, const void* u, const void* bias, int bias_img_stride,
                               const void* residual, void* y, int N, int H, int W, int Cin, int Cout, float* stat_part)
{
    if (!x || !u || !y) return fail(GD_NN_ERR_INVALID_ARG, "null pointer");
    if (residual) return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_regw: no residual form");
    if (!gd_nn_conv3x3_regw_supported(N, H, W, Cin, Cout))
        return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_regw: need Cin = Cout = 128, H % 16 == 0, W % 32 == 0, tensors < 2 GiB");
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return fail(GD_NN_ERR_HIP, "hipGetDevice failed");
    hipStream_t s = (hipStream_t)stream;
    // column strips of 32 pixels, cut into vertical segments (multiples of 16 rows) until the grid fills the chip twice
    const int strips = W / 32;
    int segs = (512 + N * strips - 1) / (N * strips);
    if (segs > H / 16) segs = H / 16;
    if (segs < 1) segs = 1;
    int seg_rows = ((H + segs - 1) / segs + 15) & ~15;
    segs = (H + seg_rows - 1) / seg_rows;
    const int nwg = N * strips * segs;
    const int64_t M = (int64_t)N * H * W;
    hipEvent_t ea = nullptr, eb = nullptr;
    if (g_cprof.on) {
        std::lock_guard<std::mutex> lk(g_cprof.mu);
        ea = g_cprof.get(); eb = g_cprof.get();
        if (ea && eb) (void)hipEventRecord(ea, s);
    }
#define GD_LAUNCH_RW(STAT_)                                                                                        \
    do {                                                                                                           \
        auto kern = conv3x3_regw_kernel<STAT_>;                                                                    \
        static bool attr_set[16] = {false};                                                                        \
        if (!attr_set[dev]) {                                                                                      \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kRegwLds);    \
            attr_set[dev] = true;                                                                                  \
        }                                                                                                          \
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), kRegwLds, s, (const uint16_t*)x, (const uint16_t*)u,        \
                           (const uint16_t*)bias, bias_img_stride, (uint16_t*)y, N, H, W, strips, segs, seg_rows,  \
                           nwg, stat_part);                                                                        \
    } while (0)
    if (stat_part) GD_LAUNCH_RW(true); else GD_LAUNCH_RW(false);
#undef GD_LAUNCH_RW
    if (ea && eb) {
        (void)hipEventRecord(eb, s);
        std::lock_guard<std::mutex> lk(g_cprof.mu);
        g_cprof.pending.push_back({ea, eb});
        g_cprof.total_flops += 2.0 * (double)M * Cout * 9.0 * Cin;
        g_cprof.total_bytes += 2.0 * ((double)M * Cin + 9.0 * Cin * Cout + (double)M * Cout +
                                      (residual ? (double)M * Cout : 0.0));
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}
