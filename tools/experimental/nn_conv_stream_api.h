// C entry points of the weight-streaming small-map convolution experiment (nn_conv_stream.h), included by csrc/nn_conv3x3.hip
// when a tools/ build defines GD_NN_EXPERIMENTAL_STREAM (tools/stream_variants.sh).  Measured BEHIND the split-K implicit-GEMM
// route on every layer of the workload but the 64-pixel ones (profiles/r05_stream_conv.txt): not part of libgd_nn.so,
// include/gd_nn.h or nn_ops.
extern "C" {
int gd_nn_conv3x3_stream_supported(int N, int H, int W, int Cin, int Cout);
size_t gd_nn_conv3x3_stream_weights_bytes(int Cout, int Cin);
int gd_nn_conv3x3_stream_weights(void* stream, const void* weight, void* wp, int Cout, int Cin);
size_t gd_nn_conv3x3_stream_ws_bytes(int N, int H, int W, int Cin, int Cout);
int gd_nn_conv3x3_stream_forward(void* stream, const void* x, const void* wp, const void* bias, int bias_img_stride,
                                 const void* residual, void* y, int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes);
}
// ---- weight-streaming form for maps of a few hundred pixels (nn_conv_stream.h) ------------------------------------------
struct StreamPlan { int fa, fb, tiles_n, tiles_m, groups, steps_total, steps_per_wave; };
int g_stream_waves = 0;     // GD_NN_STREAM_WAVES: target number of waves (tuning); 0 = default
static bool stream_plan(int N, int H, int W, int Cin, int Cout, StreamPlan* p)
{
    static int env_read = 0;
    if (!env_read) { if (const char* e = getenv("GD_NN_STREAM_WAVES")) g_stream_waves = atoi(e); env_read = 1; }
    const int64_t M = (int64_t)N * H * W;
    if (N <= 0 || H < 1 || W < 1 || M > 512 || Cin < 64 || Cin % 16 || Cout < 64 || Cout % 32) return false;
    if ((double)M * Cin * 2.0 >= 2147483648.0) return false;
    if (M <= 64 && Cout % 64 == 0) { p->fa = 2; p->fb = 2; }
    else if (M <= 128 && Cout % 64 == 0) { p->fa = 2; p->fb = 4; }
    else { p->fa = 1; p->fb = 8; }
    p->tiles_n = Cout / (32 * p->fa);
    p->tiles_m = (int)((M + 32 * p->fb - 1) / (32 * p->fb));
    p->steps_total = 9 * (Cin / 16);
    const int target = g_stream_waves > 0 ? g_stream_waves : 1280;
    int groups = (target + p->tiles_n * p->tiles_m * kStreamWaves - 1) / (p->tiles_n * p->tiles_m * kStreamWaves);
    const int max_groups = p->steps_total / (kStreamWaves * 4) > 0 ? p->steps_total / (kStreamWaves * 4) : 1;   // >= 4 steps per wave
    if (groups > max_groups) groups = max_groups;
    if (groups < 1) groups = 1;
    p->steps_per_wave = (p->steps_total + groups * kStreamWaves - 1) / (groups * kStreamWaves);
    p->groups = (p->steps_total + p->steps_per_wave * kStreamWaves - 1) / (p->steps_per_wave * kStreamWaves);     // no empty group
    return true;
}

int gd_nn_conv3x3_stream_supported(int N, int H, int W, int Cin, int Cout)
{
    StreamPlan p;
    return stream_plan(N, H, W, Cin, Cout, &p) ? 1 : 0;
}

size_t gd_nn_conv3x3_stream_weights_bytes(int Cout, int Cin) { return (size_t)Cout * 9 * Cin * 2; }

int gd_nn_conv3x3_stream_weights(void* stream, const void* weight, void* wp, int Cout, int Cin)
{
    if (!weight || !wp) return fail(GD_NN_ERR_INVALID_ARG, "stream_weights: null pointer");
    if (Cout <= 0 || Cout % 32 || Cin <= 0 || Cin % 16) return fail(GD_NN_ERR_INVALID_ARG, "stream_weights: need Cout % 32 == 0, Cin % 16 == 0");
    const size_t total = (size_t)(Cout / 32) * 9 * (Cin / 16) * 64;
    hipLaunchKernelGGL(conv3x3_stream_weights_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256),
                       0, (hipStream_t)stream, (const uint16_t*)weight, (uint4*)wp, Cout, Cin);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

size_t gd_nn_conv3x3_stream_ws_bytes(int N, int H, int W, int Cin, int Cout)
{
    StreamPlan p;
    if (!stream_plan(N, H, W, Cin, Cout, &p)) return 0;
    return (size_t)p.groups * p.tiles_m * p.tiles_n * p.fa * p.fb * 1024 * sizeof(float);
}

int gd_nn_conv3x3_stream_forward(void* stream, const void* x, const void* wp, const void* bias, int bias_img_stride,
                                 const void* residual, void* y, int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes)
{
    if (!x || !wp || !y || !ws) return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_stream: null pointer");
    StreamPlan p;
    if (!stream_plan(N, H, W, Cin, Cout, &p))
        return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_stream: need N*H*W <= 512, Cin >= 64 and % 16 == 0, Cout >= 64 and % 32 == 0");
    if (ws_bytes < gd_nn_conv3x3_stream_ws_bytes(N, H, W, Cin, Cout)) return fail(GD_NN_ERR_INVALID_ARG, "conv3x3_stream: workspace too small");
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return fail(GD_NN_ERR_HIP, "hipGetDevice failed");
    hipStream_t s = (hipStream_t)stream;
    const int M = N * H * W;
    hipEvent_t ea = nullptr, eb = nullptr;
    if (g_cprof.on) {
        std::lock_guard<std::mutex> lk(g_cprof.mu);
        ea = g_cprof.get(); eb = g_cprof.get();
        if (ea && eb) (void)hipEventRecord(ea, s);
    }
#define GD_LAUNCH_ST(FA_, FB_)                                                                                        \
    do {                                                                                                              \
        auto kern = conv3x3_stream_kernel<FA_, FB_>;                                                                  \
        constexpr int lds = (kStreamWaves / 2) * FA_ * FB_ * 1024 * 4;                                                \
        static bool attr_set[16] = {false};                                                                           \
        if (!attr_set[dev]) {                                                                                         \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);            \
            attr_set[dev] = true;                                                                                     \
        }                                                                                                             \
        hipLaunchKernelGGL(kern, dim3(p.tiles_n, p.groups, p.tiles_m), dim3(64 * kStreamWaves), lds, s,               \
                           (const uint16_t*)x, (const uint4*)wp, (float*)ws, N, H, W, Cin, p.steps_total,             \
                           p.steps_per_wave);                                                                         \
        const int quads = p.tiles_n * p.tiles_m * FA_ * FB_ * 256;                                                    \
        hipLaunchKernelGGL((conv_stream_reduce_kernel<FA_, FB_>), dim3((quads + 255) / 256), dim3(256), 0, s,         \
                           (const float*)ws, p.groups, p.tiles_n, p.tiles_m, M, H * W, Cout, (const uint16_t*)bias,   \
                           bias_img_stride, (const uint16_t*)residual, (uint16_t*)y);                                 \
    } while (0)
    if (p.fa == 2 && p.fb == 2) GD_LAUNCH_ST(2, 2);
    else if (p.fa == 2 && p.fb == 4) GD_LAUNCH_ST(2, 4);
    else GD_LAUNCH_ST(1, 8);
#undef GD_LAUNCH_ST
    if (ea && eb) {
        (void)hipEventRecord(eb, s);
        std::lock_guard<std::mutex> lk(g_cprof.mu);
        g_cprof.pending.push_back({ea, eb});
        g_cprof.total_flops += 2.0 * (double)M * Cout * 9.0 * Cin;
        g_cprof.total_bytes += 2.0 * ((double)M * Cin + 9.0 * Cin * Cout + (double)M * Cout + (residual ? (double)M * Cout : 0.0));
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GD_NN_ERR_HIP, hipGetErrorString(e));
    return GD_NN_OK;
}

