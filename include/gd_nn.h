/*
 * gd_nn.h -- C-ABI of the hand-written gfx950 kernels used inside the SDS guidance step
 * (SD-2.1 VAE encoder + UNet).  In the reference these operations are dispatched by PyTorch from
 * diffusers==0.19.0 modules (un-vendored; call sites
 * Garment_3DGS/threestudio/models/guidance/stable_diffusion_guidance.py:153-157,165-166), e.g.
 * diffusers' ResnetBlock2D = GroupNorm(32) -> SiLU -> conv3x3 (skeleton visible in-tree at
 * Garment_Deformer_NeTF/netf/vsd/lora_unet.py:119-160).  Each entry point below replaces one such
 * PyTorch-dispatched op sequence; tensors are bf16, NHWC ("channels_last"), device pointers.
 *
 * Return >= 0 on success, negative GD_NN_ERR_* otherwise; gd_nn_last_error() has the message.
 */
#ifndef GD_NN_H_INCLUDED
#define GD_NN_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GD_NN_OK 0
#define GD_NN_ERR_INVALID_ARG (-1)
#define GD_NN_ERR_HIP (-2)

/* y = act(GroupNorm_G(x) * gamma + beta), act = SiLU if apply_silu else identity.
 * x, y: bf16 [N, HW, C] (NHWC); gamma, beta: bf16 [C]; C % 8 == 0, C % G == 0.
 * stats_ws: gd_nn_groupnorm_ws_bytes(N, G) bytes that are ZERO on entry (zero-initialise once; every call
 * leaves them zero again, so one workspace serves all calls on a stream -- the last statistics workgroup
 * finalises and clears, there is no memset / finalize launch); mean_rstd: N*G*2 floats out (saved for backward).  Replaces F.group_norm + F.silu (two kernels + two NCHW<->NHWC copies in
 * PyTorch-ROCm's native path).  stats_ws == NULL: mean_rstd is an INPUT (statistics already known -- from
 * gd_nn_groupnorm_finish_partials or gd_nn_groupnorm_stats) and only the apply pass runs. */
int gd_nn_groupnorm_silu_forward(void* stream, const void* x, void* y, const void* gamma, const void* beta, int N,
                                 int HW, int C, int G, float eps, int apply_silu, double* stats_ws,
                                 float* mean_rstd);

/* dx for the same op (weights frozen: no dgamma / dbeta, as in the guidance where
 * requires_grad_(False) is set on every VAE/UNet parameter,
 * stable_diffusion_guidance.py:99-102).  dy: gradient w.r.t. y.  add (optional, bf16, shape of dx): a second
 * gradient arriving at x -- a ResnetBlock's skip path -- summed in fp32 before the one bf16 rounding, which saves the
 * separate accumulation pass autograd would run (3 tensor passes -> 1 extra read). */
int gd_nn_groupnorm_silu_backward(void* stream, const void* x, const void* dy, const void* gamma,
                                  const void* beta, const float* mean_rstd, void* dx, int N, int HW, int C, int G,
                                  int apply_silu, double* stats_ws, float* group_sums /* N*G*2 floats scratch */,
                                  const void* add);

size_t gd_nn_groupnorm_ws_bytes(int N, int G);

/* y = conv3x3(x, w; stride 1, pad 1) [+ bias] [+ residual], NHWC bf16, as an MFMA implicit GEMM.
 * x: [N,H,W,Cin]; weight: [Cout][3][3][Cin] (PyTorch's channels_last weight storage); bias: [Cout]
 * (bias_img_stride = 0) or [N][Cout] (bias_img_stride = elements between rows, >= Cout and a multiple of 8:
 * per-image bias, e.g. conv bias + time-embedding projection of diffusers' ResnetBlock2D; a column slice of a
 * wider matrix is passed with that matrix's row stride); residual: [N,H,W,Cout] or NULL;
 * Cin % 64 == 0, Cout % 4 == 0.  Replaces F.conv2d (+ bias kernel + residual add kernel). */
int gd_nn_conv3x3_forward(void* stream, const void* x, const void* weight, const void* bias, int bias_img_stride,
                          const void* residual, void* y, int N, int H, int W, int Cin, int Cout);

/* flipped[ci][tap][co] = weight[co][8-tap][ci]: the weights with which the SAME kernel computes the
 * input gradient dx = conv3x3(dy, flipped) (weights are frozen in the guidance; no wgrad). */
int gd_nn_conv3x3_flip_weights(void* stream, const void* weight, void* flipped, int Cout, int Cin);

/* Input gradient of the FIRST convolution of the VAE encoder (1 <= Cin <= 3 image channels, Cout = 128; autograd of diffusers'
 * AutoencoderKL.encoder.conv_in under stable_diffusion_guidance.py:165-166 encode_images): dy [N,H,W,128] bf16 is read once --
 * a 128 -> 9 taps x 3 channels product on the matrix cores per 16x16 tile + halo, the nine shifted partial results of each pixel
 * summed in fp32 through LDS.  wpack: 32 x 128 bf16 from gd_nn_conv3x3_first_dgrad_weights (weight: [Cout][3][3][Cin]; once per
 * frozen weight); dx4: [N,H,W,4] bf16, channels >= Cin written as 0.  Replaces the Cout-padded-to-32 implicit-GEMM launch. */
int gd_nn_conv3x3_first_dgrad_supported(int N, int H, int W, int Cin, int Cout);
int gd_nn_conv3x3_first_dgrad_weights(void* stream, const void* weight, void* wpack, int Cout, int Cin);
int gd_nn_conv3x3_first_dgrad(void* stream, const void* dy, const void* wpack, void* dx4, int N, int H, int W, int Cin, int Cout);

/* Tuning hook: force the tile variant (0 = 128x128/4 waves, 1 = 128 ch x 256 px/8 waves,
 * 2 = 256x256/8 waves, -1 = built-in heuristic). */
int gd_nn_conv_force_variant(int v);

/* Event timing of the conv kernel for bench.py's roofline object (off by default). */
/* nn.Linear on the MFMA implicit-GEMM kernel (one tap): y[M][Nout] = x[M][K] . w[Nout][K]^T + bias + residual, bf16,
 * K % 64 == 0, Nout % 4 == 0 (diffusers Attention / FeedForward / proj_in / proj_out linears). */
int gd_nn_linear_forward(void* stream, const void* x, const void* weight, const void* bias, const void* residual, void* y,
                         int64_t M, int K, int Nout);
/* The transformer linears as a GEMM designed as a GEMM (csrc/nn_gemm.hip; replaces hipBLASLt behind diffusers' Attention /
 * FeedForward projections, call site stable_diffusion_guidance.py:153-157): persistent 256 x 256 x 64 tiles, 8 waves, both
 * operands through a ten-slot LDS-DMA ring with counted waits, XCD-aware tile order.  bf16 in / out, fp32 accumulation.
 *   gd_nn_gemm_forward        y[M][N] = bf16(residual[M][N] + bias[N] + x[M][K] . w[N][K]^T): bias and residual are the
 *                             accumulators' starting value, ONE rounding (torch.addmm's, not the eager `linear` then `add`
 *                             pair's two); bias / residual may be NULL
 *   gd_nn_gemm_geglu_forward  diffusers GEGLU(K, inner): w [2 inner][K] (hidden rows, then gate rows), bias [2 inner] or NULL,
 *                             y[M][inner] = (x w_h^T + b_h) * gelu(x w_g^T + b_g) with the rounding points of the projection
 *                             followed by gd_nn_geglu_forward (bit-identical to that pair when the projection is this kernel's)
 * The summation order of an output depends on K only (not on M / the tile / the grid): batch-invariant by construction.
 * gd_nn_gemm_supported: K % 64 == 0, N % 8 == 0, every tensor < 2 GiB.  Pointers 16-byte aligned (y / residual 8). */
int gd_nn_gemm_supported(int64_t M, int K, int N, int geglu);
int gd_nn_gemm_forward(void* stream, const void* x, const void* weight, const void* bias, const void* residual, void* y,
                       int64_t M, int K, int N);
int gd_nn_gemm_geglu_forward(void* stream, const void* x, const void* weight, const void* bias, void* y, int64_t M, int K,
                             int inner);
const char* gd_nn_gemm_last_error(void);
int gd_nn_conv_profile_enable(int on);
int gd_nn_conv_profile_reset(void);
int gd_nn_conv_profile_read(double* total_ms, int64_t* launches, double* total_flops);
int gd_nn_conv_profile_read_bytes(double* total_bytes);   /* algorithmic HBM bytes of the same launches */

/* gd_nn_conv3x3_forward with a caller-provided scratch buffer: layers whose 128x128 tile grid cannot fill the
 * chip (small feature maps, one view per GPU) are split over the nine taps (3 or 9 workgroups per tile, fp32
 * partials in `ws`, combined with bias / residual by a second small kernel).  gd_nn_conv3x3_ws_bytes() gives
 * the bytes needed (0: the layer is not split); with ws == NULL the call behaves like gd_nn_conv3x3_forward. */
size_t gd_nn_conv3x3_ws_bytes(int N, int H, int W, int Cin, int Cout);
int gd_nn_conv3x3_forward_ws(void* stream, const void* x, const void* weight, const void* bias, int bias_img_stride,
                             const void* residual, void* y, int N, int H, int W, int Cin, int Cout, void* ws,
                             size_t ws_bytes);
/* tuning hook: -1 heuristic (default), 1 never split, 3 / 9 force that split factor */
int gd_nn_conv_force_split(int s);
/* Batch-invariant kernel selection for sharded runs: every tile / split-K / patch-vs-GEMM rule that looks at the
 * batch sees N * k images, so a rank that holds 1/k of the views (threestudio's DDP split of the camera batch,
 * GaussianDreamer.py:189-191) picks the kernels - and therefore the bf16 summation orders - the single-rank run
 * of the whole batch picks.  k = 1 (default): each launch is tuned for the batch it gets. */
int gd_nn_conv_set_route_scale(int k);

/* gd_nn_groupnorm_silu_forward as ONE launch for inference on small feature maps (the UNet's GroupNorms at <= 16
 * latents: diffusers ResnetBlock2D.norm1 / norm2, Transformer2DModel.norm, conv_norm_out): one workgroup per (image,
 * group) holds its HW x C/G slice in registers -- x is read once, no workspace.  _supported: C % G == 0, C / G even, 4 <= C / G <= 128, HW * C / G <= 65536 elements
 * (<= 32768 when C / G < 16: short channel runs load poorly)
 * (the slice lives in the workgroup's registers). */
int gd_nn_groupnorm_silu_fused_supported(int N, int HW, int C, int G);
int gd_nn_groupnorm_silu_fused_forward(void* stream, const void* x, void* y, const void* gamma, const void* beta, int N,
                                       int HW, int C, int G, float eps, int apply_silu);
/* The same launch with the statistics kept (mean_rstd[N][G][2] = {mean, 1 / sqrt(var + eps)}, may be NULL), and the
 * input gradient of such a layer as ONE launch as well -- the training pass of the NeTF stage's LoRA UNet
 * (Garment_Deformer_NeTF/netf/vsd/lora_unet.py:119-160 skeleton, trained by netf/trainer.py:215-256), whose feature maps
 * are one or two latents: the (image, group) workgroup holds its slices of x and dy in registers, forms the two group
 * sums of gd_nn_groupnorm_silu_backward and writes dx (frozen gamma / beta; bit-reproducible); larger slices take
 * gd_nn_groupnorm_silu_backward on the kept statistics. */
int gd_nn_groupnorm_silu_fused_forward_stats(void* stream, const void* x, void* y, const void* gamma, const void* beta, int N,
                                             int HW, int C, int G, float eps, int apply_silu, float* mean_rstd);
int gd_nn_groupnorm_silu_fused_backward_supported(int N, int HW, int C, int G);   /* _supported and HW * C / G <= 32768 */
int gd_nn_groupnorm_silu_fused_backward(void* stream, const void* x, const void* dy, const void* gamma, const void* beta,
                                        const float* mean_rstd, void* dx, int N, int HW, int C, int G, int apply_silu);

/* GroupNorm statistics only: mean_rstd[N][G][2] = {mean, 1/sqrt(var + eps)} of x (bf16 [N,HW,C]); stats_ws as in
 * gd_nn_groupnorm_silu_forward.  Feeds gd_nn_conv3x3_gn_forward (and gd_nn_groupnorm_silu_backward). */
int gd_nn_groupnorm_stats(void* stream, const void* x, int N, int HW, int C, int G, float eps, double* stats_ws,
                          float* mean_rstd);

/* y = conv3x3_s1_p1( act( GroupNorm_G(x) * gamma + beta ) ) + bias (+ residual): diffusers ResnetBlock2D's
 * ``conv(nonlinearity(norm(x)))`` (skeleton: Garment_Deformer_NeTF/netf/vsd/lora_unet.py:119-160) as ONE kernel --
 * the normalised / activated tensor is produced in the convolution's activation loader (16x16 spatial patch
 * staged once per 64 channels, nine taps read it from LDS) and never written to HBM.  mean_rstd from
 * gd_nn_groupnorm_stats (NULL: plain convolution of x); act = SiLU if apply_silu.  Other arguments as
 * gd_nn_conv3x3_forward. */
int gd_nn_conv3x3_gn_forward(void* stream, const void* x, const float* mean_rstd, const void* gamma, const void* beta,
                             int groups, int apply_silu, const void* weight, const void* bias, int bias_img_stride,
                             const void* residual, void* y, int N, int H, int W, int Cin, int Cout);

/* GroupNorm statistics of a convolution's OUTPUT from that convolution's epilogue, for ResnetBlock2D chains
 * (conv -> GroupNorm -> conv ...: Garment_Deformer_NeTF/netf/vsd/lora_unet.py:119-160 pattern; the VAE encoder of
 * threestudio's StableDiffusionGuidance.encode_images is eleven such blocks): the statistics pass of the next
 * GroupNorm is a full re-read of the tensor just written (537 MB at the 512^2 level, 8 views).  The `_stats` entries
 * run the patch-staged kernels with one extra output: stat_part[N][Cout/4][rows] float2 = {sum, sum of squares} of the
 * bf16 values stored, per 4-channel quad and per 16-pixel row segment of a tile; rows =
 * gd_nn_conv3x3_stat_rows(N, H, W, Cout, gn_entry) (0: this shape does not run on a patch-staged kernel --
 * gn_entry = 1 for gd_nn_conv3x3_gn_forward_stats, which always does).  Every element of stat_part is written by
 * every call (no zeroing needed, no atomics: the statistics are bit-reproducible).
 * gd_nn_groupnorm_finish_partials turns them into mean_rstd[N][G][2] (fp64 sums; needs (C / G) % 4 == 0). */
size_t gd_nn_conv3x3_stat_rows(int N, int H, int W, int Cout, int gn_entry);
int gd_nn_conv3x3_gn_forward_stats(void* stream, const void* x, const float* mean_rstd, const void* gamma,
                                   const void* beta, int groups, int apply_silu, const void* weight, const void* bias,
                                   int bias_img_stride, const void* residual, void* y, int N, int H, int W, int Cin,
                                   int Cout, float* stat_part);
int gd_nn_conv3x3_forward_stats(void* stream, const void* x, const void* weight, const void* bias, int bias_img_stride,
                                const void* residual, void* y, int N, int H, int W, int Cin, int Cout, float* stat_part);
/* The same stride-1 convolution with the Winograd F(2,3) minimal-filtering transform along x
 * (csrc/nn_conv_wino.h): 2/3 of the matrix-core work of the direct form.  u = the transformed filter bank of a
 * weight tensor [Cout][3][3][Cin], stored as the sequence of 32 KB LDS images the kernel streams -- per (128-channel
 * block, ky, 32-channel chunk): [4 positions][128 rows][32 ch], gd_nn_conv3x3_wino_weights_bytes() bytes in all
 * (gd_nn_conv3x3_wino_weights; the caller
 * caches it per frozen weight like the flipped dgrad weights -- dgrad = the same kernel on the transform of the
 * flipped weights).  Needs Cin % 32 == 0 and Cout % 8 == 0 (gd_nn_conv3x3_wino_supported).  stat_part: NULL or the GroupNorm partial
 * sums of the output as in gd_nn_conv3x3_forward_stats (rows = ceil(H/16) * ceil(W/16) * 8).
 * Replaces the same reference call as gd_nn_conv3x3_forward (diffusers ResnetBlock2D conv1 / conv2, reached from
 * Garment_3DGS/threestudio/models/guidance/stable_diffusion_guidance.py:153-167). */
int gd_nn_conv3x3_wino_supported(int N, int H, int W, int Cin, int Cout);
size_t gd_nn_conv3x3_wino_weights_bytes(int Cout, int Cin);     /* bytes of u (Cout padded to a multiple of 128) */
int gd_nn_conv3x3_wino_weights(void* stream, const void* weight, void* u, int Cout, int Cin);
int gd_nn_conv3x3_wino_forward(void* stream, const void* x, const void* u, const void* bias, int bias_img_stride,
                               const void* residual, void* y, int N, int H, int W, int Cin, int Cout, float* stat_part);
/* ... with GroupNorm(+SiLU) of the input applied in the loader, as gd_nn_conv3x3_gn_forward(_stats). */
int gd_nn_conv3x3_wino_gn_forward(void* stream, const void* x, const float* mean_rstd, const void* gamma, const void* beta,
                                  int groups, int apply_silu, const void* u, const void* bias, int bias_img_stride,
                                  const void* residual, void* y, int N, int H, int W, int Cin, int Cout, float* stat_part);
/* The same stride-1 convolution on a 128-channel x (16 x 32)-pixel tile (csrc/nn_conv_wide.h), for layers whose FEW
 * output channels (Cout <= 128 per tile) leave the 256-channel tile of gd_nn_conv3x3_forward unavailable: 48 MFMAs per
 * wave and barrier, 0.77 KB of filter / patch LDS-DMA per MFMA.  u = the filter bank re-packed as the sequence of 24 KB
 * LDS images the kernel streams -- per (128-channel block, ky, 32-channel chunk): [3 kx][128 rows][32 ch],
 * gd_nn_conv3x3_wide_weights_bytes() bytes (gd_nn_conv3x3_wide_weights; cached per frozen weight by the caller, dgrad =
 * the same kernel on the packing of the flipped weights).  Needs Cin % 32 == 0 and Cout % 8 == 0.  stat_part and the
 * _gn_ form as for the Winograd entry points above (same partial-sum rows).  Replaces the same reference call. */
int gd_nn_conv3x3_wide_supported(int N, int H, int W, int Cin, int Cout);
size_t gd_nn_conv3x3_wide_weights_bytes(int Cout, int Cin);
int gd_nn_conv3x3_wide_weights(void* stream, const void* weight, void* u, int Cout, int Cin);
int gd_nn_conv3x3_wide_forward(void* stream, const void* x, const void* u, const void* bias, int bias_img_stride,
                               const void* residual, void* y, int N, int H, int W, int Cin, int Cout, float* stat_part);
int gd_nn_conv3x3_wide_gn_forward(void* stream, const void* x, const float* mean_rstd, const void* gamma, const void* beta,
                                  int groups, int apply_silu, const void* u, const void* bias, int bias_img_stride,
                                  const void* residual, void* y, int N, int H, int W, int Cin, int Cout, float* stat_part);
/* ... and of the first convolution (gd_nn_conv3x3_first_forward below; Cout == 128 only, the VAE encoder's conv_in):
 * rows = gd_nn_conv3x3_first_stat_rows(N, H, W, Cin, Cout), 0 when that shape has no statistics path. */
size_t gd_nn_conv3x3_first_stat_rows(int N, int H, int W, int Cin, int Cout);
int gd_nn_conv3x3_first_forward_stats(void* stream, const void* x, const void* weight, const void* bias, void* y, int N,
                                      int H, int W, int Cin, int Cout, float* stat_part);
int gd_nn_groupnorm_finish_partials(void* stream, const float* stat_part, int N, size_t rows, int C, int G, int HW,
                                    float eps, float* mean_rstd);

/* nn.Linear with K = 320 and N = 320, 640 or 2560 on long row sets (to_q / to_out / proj_in / proj_out, the fused q|k
 * projection and the GEGLU projection of the UNet's 64x64-token transformer blocks: y[M][N] = x[M][320] .
 * weight[N][320]^T + bias): an HBM stream, run with the weights held in registers by persistent ten-wave workgroups,
 * one block of 320 output channels each (csrc/nn_linear.hip).  gd_nn_linear_320_supported: 1 when (M, K, N) is such a
 * product. */
int gd_nn_linear_320_supported(int64_t M, int K, int N);
int gd_nn_linear_k320_forward(void* stream, const void* x, const void* weight, const void* bias, void* y, int64_t M, int N);
int gd_nn_linear_320_forward(void* stream, const void* x, const void* weight, const void* bias, void* y, int64_t M);
/* ... with diffusers' GEGLU as the epilogue: weight [2 * inner][320] (hidden rows, then gate rows), bias [2 * inner],
 * y[M][inner] = (x W_h^T + b_h) * gelu(x W_g^T + b_g) with the rounding points of gd_nn_linear_k320_forward followed by
 * gd_nn_geglu_forward (bit-identical to that pair); inner = 1280.  The [M][2 * inner] intermediate never exists. */
int gd_nn_linear_k320_geglu_forward(void* stream, const void* x, const void* weight, const void* bias, void* y, int64_t M,
                                    int inner);
const char* gd_nn_linear_320_last_error(void);

/* First convolution (image / latent -> features): 3x3 / s1 / p1 with Cin <= 4, + bias.  x: bf16 [N,H,W,Cin];
 * weight: bf16 [Cout][3][3][Cin]; y: bf16 [N,H,W,Cout]; Cout % 8 == 0, 36*Cin*Cout bytes of LDS <= 64 KiB.
 * Cout == 128 (diffusers `conv_in` of the VAE encoder): matrix-core kernel, im2col operand gathered from the image,
 * bias as an extra K column; other Cout (the UNet's 320): VALU kernel. */
int gd_nn_conv3x3_first_forward(void* stream, const void* x, const void* weight, const void* bias, void* y, int N, int H,
                                int W, int Cin, int Cout);

/* 3x3 convolution with stride 2 and padding (pad_lo, 1) per spatial dim -- pad_lo = 1: Conv2d(k3, s2, p1), the
 * UNet's Downsample2D; pad_lo = 0: the VAE encoder's F.pad(x, (0,1,0,1)) + Conv2d(k3, s2, p0) (diffusers
 * Downsample2D; un-vendored, reached through stable_diffusion_guidance.py:153-166) without materialising the
 * padded tensor.  x: bf16 [N,Hin,Win,Cin]; weight: bf16 [Cout][3][3][Cin]; bias: bf16 [Cout] or NULL;
 * y: bf16 [N,Ho,Wo,Cout], Ho = (Hin + pad_lo - 2) / 2 + 1.  Same MFMA kernel as gd_nn_conv3x3_forward. */
int gd_nn_conv3x3_s2_forward(void* stream, const void* x, const void* weight, const void* bias, void* y, int N, int Hin,
                             int Win, int Cin, int Cout, int pad_lo);

/* y = conv3x3_s1_p1( nearest_upsample_2x(x) ) + bias, diffusers Upsample2D (F.interpolate(scale_factor=2, "nearest")
 * followed by the 3x3 convolution; UNet up blocks) without the upsampled tensor: each output-pixel parity class
 * (py, px) is a 2x2-tap convolution of x with pre-summed filters -- 16 tap GEMMs instead of 36, and no 4x larger
 * intermediate.  x: [N,H,W,Cin], y: [N,2H,2W,Cout], bias [Cout] or NULL.  w_even_rows / w_odd_rows: [Cout][9][Cin]
 * bf16, slot px*4 + ty*2 + tx = sum of the original taps (ky, kx) with ky in R(py, ty), kx in R(px, tx),
 * R(0,0) = {0}, R(0,1) = {1,2}, R(1,0) = {0,1}, R(1,1) = {2}; slot 8 unused. */
int gd_nn_conv3x3_up2_forward(void* stream, const void* x, const void* w_even_rows, const void* w_odd_rows,
                              const void* bias, void* y, int N, int H, int W, int Cin, int Cout);

/* Input gradient of gd_nn_conv3x3_s2_forward: dx[N,Hin,Win,Cin] from dy[N,Ho,Wo,Cout] and
 * weight_flipped = gd_nn_conv3x3_flip_weights(weight) ([Cin][3][3][Cout]).  Four launches, one per parity
 * class of the input pixel, each walking only the taps that reach it (9 taps in total: no zero-insertion
 * waste); every dx element is written exactly once.  Needs Cout % 64 == 0. */
int gd_nn_conv3x3_s2_dgrad(void* stream, const void* dy, const void* weight_flipped, void* dx, int N, int Hin, int Win,
                           int Cin, int Cout, int pad_lo);

/* Split-K forms of the two stride-2 entry points for small maps (one or two views per GPU: a 16x16 -> 8x8 layer
 * has 10 tiles for 256 CUs): the K-step sequence is dealt to several workgroups per tile, fp32 partials go to `ws`.
 * gd_nn_conv3x3_s2_ws_bytes() gives the size the heuristic wants (0 = runs unsplit; dgrad != 0: for the input
 * gradient); with ws == NULL or too small these behave exactly like the entry points above. */
size_t gd_nn_conv3x3_s2_ws_bytes(int N, int Hin, int Win, int Cin, int Cout, int pad_lo, int dgrad);
int gd_nn_conv3x3_s2_forward_ws(void* stream, const void* x, const void* weight, const void* bias, void* y, int N,
                                int Hin, int Win, int Cin, int Cout, int pad_lo, void* ws, size_t ws_bytes);
int gd_nn_conv3x3_s2_dgrad_ws(void* stream, const void* dy, const void* weight_flipped, void* dx, int N, int Hin, int Win,
                              int Cin, int Cout, int pad_lo, void* ws, size_t ws_bytes);

/* y[rows, inner] = x[rows, :inner] * gelu(x[rows, inner:])  (erf GELU, bf16, inner % 8 == 0): diffusers'
 * GEGLU activation of the transformer blocks' feed-forward (``hidden, gate = proj(x).chunk(2, -1);
 * hidden * F.gelu(gate)``; the UNet skeleton is un-vendored, call site stable_diffusion_guidance.py:153-157).
 * Replaces chunk + gelu + mul (5 row passes) by one 3-pass kernel.  Inference only. */
int gd_nn_geglu_forward(void* stream, const void* x, void* y, int64_t rows, int inner);
/* Input gradients of the two row passes, for the training pass of the NeTF stage's LoRA UNet (frozen base: no parameter
 * gradients).  geglu_backward: dx [rows][2 * inner] = [dy * gelu(g) | dy * h * gelu'(g)] from x = [h | g] and dy [rows][inner]
 * (autograd of diffusers GEGLU, netf/vsd/lora_unet.py's FeedForward).  layernorm_backward: dx = d LayerNorm(s) / d s applied to
 * dy, + ds (the gradient reaching s directly through the residual stream; may be NULL); mean / rstd recomputed from s. */
int gd_nn_geglu_backward(void* stream, const void* x, const void* dy, void* dx, int64_t rows, int inner);
int gd_nn_layernorm_backward(void* stream, const void* s, const void* dy, const void* weight, const void* ds, void* dx,
                             int64_t rows, int C, float eps);

/* s = x + residual (residual may be NULL -> s = x); if sum_out != NULL store s (bf16) there;
 * y = LayerNorm_C(s) * weight + bias.  x, residual, sum_out, y: bf16 [rows, C]; weight, bias: bf16 [C];
 * C % 8 == 0, C <= 2048.  Fuses BasicTransformerBlock's ``x = x + attn(...)`` with the following
 * ``norm(x)`` (one wave per row, fp32 statistics).  Inference only. */
int gd_nn_add_layernorm_forward(void* stream, const void* x, const void* residual, const void* weight, const void* bias,
                                float eps, void* sum_out, void* y, int64_t rows, int C);

/* Fused self-attention forward, head_dim 64, bf16, no mask (diffusers Attention -> scaled_dot_product_attention in
 * the UNet's spatial self-attention).  q, o: [B][S][H*64] with row stride q_rs / o_rs and batch stride q_bs / o_bs
 * (elements); k, v: [B][kv_len][H*64] likewise; Skv = kv_len rounded up to 64 (the padded keys get no weight: the
 * UNet's cross-attention over 77 text tokens runs as Skv = 128, kv_len = 77).  vt_ws: gd_nn_attention_ws_bytes(B, Skv, H) bytes of
 * scratch (V transposed per head with the key order the MFMA accumulator layout wants).  o = softmax(q k^T scale) v. */
size_t gd_nn_attention_ws_bytes(int B, int Skv, int H);
int gd_nn_attention_d64_forward(void* stream, const void* q, const void* k, const void* v, void* o, void* vt_ws, int B, int S,
                                int Skv, int H, int64_t q_bs, int q_rs, int64_t k_bs, int k_rs, int64_t v_bs, int v_rs,
                                int64_t o_bs, int o_rs, float scale, int kv_len);
/* The same kernel for a caller that already holds V TRANSPOSED: vt = [B][H][64][Skv] contiguous bf16, keys in natural
 * order, all Skv keys valid (Skv % 64 == 0) -- e.g. the output of the GEMM  W_v x^T, which replaces the V projection
 * AND the transposing pre-pass of the entry above. */
/* The same forward pass, also returning lse [B][H][S] fp32 = ln sum_k exp(scale * q.k) per query -- what a flash-attention
 * BACKWARD pass recomputes the probabilities from (training pass of the NeTF stage's LoRA UNet: the forward runs here, the
 * backward on the library's flash kernels, which take this tensor). */
int gd_nn_attention_d64_forward_lse(void* stream, const void* q, const void* k, const void* v, void* o, float* lse, void* vt_ws,
                                    int B, int S, int Skv, int H, int64_t q_bs, int q_rs, int64_t k_bs, int k_rs, int64_t v_bs,
                                    int v_rs, int64_t o_bs, int o_rs, float scale, int kv_len);
/* Backward of the same attention (head_dim 64) from q, k, v, the forward output o, its gradient dout and lse: dq, dk, dv in the
 * layouts of q, k, v ([B][S | Skv][H * 64] rows with their own strides).  S % 64 == 0; Skv = the key count rounded up to 64,
 * kv_len the true count.  ws = gd_nn_attention_bwd_ws_bytes(B, S, Skv, H) bytes (K^T, Q^T, dO^T, rowsum(dO o O)).  Replaces
 * autograd's backward of F.scaled_dot_product_attention in the training pass of the NeTF stage's LoRA UNet
 * (netf/vsd/lora_unet.py's attention processors; netf/trainer.py:215-256). */
size_t gd_nn_attention_bwd_ws_bytes(int B, int S, int Skv, int H);
int gd_nn_attention_d64_backward(void* stream, const void* q, const void* k, const void* v, const void* o, const void* dout,
                                 const float* lse, void* dq, void* dk, void* dv, void* ws, int B, int S, int Skv, int H,
                                 int64_t q_bs, int q_rs, int64_t k_bs, int k_rs, int64_t v_bs, int v_rs, int64_t o_bs, int o_rs,
                                 int64_t do_bs, int do_rs, int64_t dq_bs, int dq_rs, int64_t dk_bs, int dk_rs, int64_t dv_bs,
                                 int dv_rs, float scale, int kv_len);
int gd_nn_attention_d64_forward_vt(void* stream, const void* q, const void* k, const void* vt, void* o, int B, int S, int Skv,
                                   int H, int64_t q_bs, int q_rs, int64_t k_bs, int k_rs, int64_t o_bs, int o_rs, float scale);
/* ... with the V^T images of the batch entries `vt_bs` elements apart (each image [H*64][Skv] contiguous) and kv_len <= Skv
 * live keys (columns >= kv_len of V^T must be zero): the cross-attention of a frozen UNet, whose V^T of the 77 text tokens
 * for ALL its layers comes out of ONE GEMM (W_v_cat . context^T; sd21.UNet2DConditionModel._project_context) -- the
 * per-layer slice is read in place, no transposing pre-pass runs. */
int gd_nn_attention_d64_forward_vt_strided(void* stream, const void* q, const void* k, const void* vt, void* o, int B, int S,
                                           int Skv, int H, int64_t q_bs, int q_rs, int64_t k_bs, int k_rs, int64_t vt_bs,
                                           int64_t o_bs, int o_rs, float scale, int kv_len);
const char* gd_nn_attention_last_error(void);

/* The guidance's image prologue as ONE kernel each way (threestudio stable_diffusion_guidance.py:394-396 + :164):
 *   y = bf16( 2 * bilinear_resize(x, (OH, OW), align_corners = False) - 1 ),  x: planar fp32 [N,3,H,W] (what the
 * rasterizer writes), y: NHWC bf16 [N,OH,OW,3] (what the VAE's first convolution reads).  Backward: dy NHWC bf16 with
 * CG >= 3 channels per pixel (only the first 3 are read) -> dx planar fp32 [N,3,H,W]; a gather per source pixel, no
 * atomics, every dx element written once. */
int gd_nn_vae_prologue_forward(void* stream, const float* x, void* y, int N, int H, int W, int OH, int OW);
int gd_nn_vae_prologue_backward(void* stream, const void* dy, float* dx, int N, int H, int W, int OH, int OW, int CG);

/* Depth-sparsity head (threestudio systems/GaussianDreamer.py:215,253): with x = depth / (dmax + 1e-5),
 *   sums2[0] = sum_i sqrt(x_i^2 + 0.01),  sums2[1] = sum_i x_i^2 / sqrt(x_i^2 + 0.01)      (fp64, zeroed inside)
 * so that loss = sums2[0] / n and d loss / d dmax = -sums2[1] / (n (dmax + 1e-5)).  dmax is a DEVICE scalar (the batch /
 * all-ranks maximum, produced by the caller so that its own gradient path is kept).  Backward:
 *   d_depth[i] = grad_out / n * x_i / sqrt(x_i^2 + 0.01) / (dmax + 1e-5). */
int gd_nn_sparsity_forward(void* stream, const float* depth, const float* dmax, int64_t n, double* sums2);
int gd_nn_sparsity_backward(void* stream, const float* depth, const float* dmax, const float* grad_out, int64_t n,
                            float* d_depth);
const char* gd_nn_prologue_last_error(void);

/* ---- fp8 (OCP e4m3) path of the no-grad UNet forward (csrc/nn_fp8.hip) --------------------------------------------
 * One fp32 scale per tensor: value = scale * e4m3 byte.  dq = scale_x * scale_w is applied to the fp32 accumulator,
 * then bias (bf16) and residual (bf16) are added and the result is stored as bf16.
 *   gd_nn_fp8_quantize       y8[i] = e4m3(clamp(x[i] * inv_scale, +-448)), x bf16, n % 8 == 0
 *   gd_nn_fp8_pack_weights   w8[r][0..Kp) = e4m3(w[r][0..K) * inv_scale) zero-padded to Kp (% 128 == 0); rows = Cout
 *                            (linear) or Cout * 9 (3x3 weights in [Cout][3][3][Cin] order)
 *   gd_nn_fp8_linear_forward y[M][Nout] = dq * x8[M][K] . w8[Nout][Kp]^T + bias[Nout] + residual[M][Nout]
 *                            (nn.Linear of the transformer blocks: diffusers Attention / FeedForward / proj_in / proj_out)
 *   gd_nn_fp8_conv3x3_forward  3x3 / stride 1 / pad 1 on NHWC e4m3 activations, per-image bias stride as
 *                            gd_nn_conv3x3_forward (diffusers ResnetBlock2D convolutions). */
int gd_nn_fp8_quantize(void* stream, const void* x_bf16, void* y_fp8, int64_t n, float inv_scale);
/* gd_nn_groupnorm_silu_forward with an e4m3 result: y8 = e4m3(clamp(bf16(act(GN(x))) * inv_scale, +-448)), NHWC bytes. */
int gd_nn_groupnorm_silu_forward_fp8(void* stream, const void* x, void* y_fp8, const void* gamma, const void* beta, int N,
                                     int HW, int C, int G, float eps, int apply_silu, double* stats_ws, float* mean_rstd,
                                     float inv_scale);
int gd_nn_fp8_pack_weights(void* stream, const void* w_bf16, void* w_fp8, int64_t rows, int K, int Kp, float inv_scale);
int gd_nn_fp8_linear_forward(void* stream, const void* x_fp8, const void* w_fp8, const void* bias, const void* residual,
                             void* y, int64_t M, int K, int Kp, int Nout, float dq);
int gd_nn_fp8_conv3x3_forward(void* stream, const void* x_fp8, const void* w_fp8, const void* bias, int bias_img_stride,
                              const void* residual, void* y, int N, int H, int W, int Cin, int CinP, int Cout, float dq);
const char* gd_nn_fp8_last_error(void);

const char* gd_nn_conv_last_error(void);
/* AutoencoderKL.quant_conv -- nn.Conv2d(8, 8, kernel_size=1) on the encoder's moments (diffusers AutoencoderKL.encode, called by
 * StableDiffusionGuidance.encode_images, threestudio/models/guidance/stable_diffusion_guidance.py:160-167): x, y NHWC bf16
 * [npix][8], weight bf16 [8][8] (Cout, Cin), bias bf16 [8] or NULL; fp32 accumulation.  transposed != 0: the input gradient
 * dx[p][ci] = sum_co weight[co][ci] dy[p][co] (bias ignored). */
int gd_nn_conv1x1_c8(void* stream, const void* x, const void* weight, const void* bias, void* y, int64_t npix, int transposed);
/* Row softmax of the VAE mid block's attention scores and its backward (diffusers Attention of AutoencoderKL's UNetMidBlock2D,
 * one head of 512 channels over 64x64 tokens; reached through StableDiffusionGuidance.encode_images,
 * threestudio/models/guidance/stable_diffusion_guidance.py:160-167): x / y / p / dp / ds bf16 [rows][L], L % 8 == 0, L <= 8192.
 * forward: y = softmax(x) per row (y may be x).  backward: ds = p * (dp - sum_j p_j dp_j) per row (ds may be dp). */
int gd_nn_softmax_rows_forward(void* stream, const void* x, void* y, int64_t rows, int L);
int gd_nn_softmax_rows_backward(void* stream, const void* p, const void* dp, void* ds, int64_t rows, int L);
const char* gd_nn_elementwise_last_error(void);
/* ---- rank-4 LoRA branch of the NeTF stage's trainable UNet (csrc/nn_lora.hip), fp32 accumulation, forward + backward.
 * Replaces diffusers 0.19 LoRALinearLayer.forward inside LoRAAttnProcessor -- hidden + scale * up(down(x)) -- and its
 * autograd backward (Garment_Deformer_NeTF/netf/vsd/lora_unet.py:119-160, 415-422; netf/trainer.py:215-256).
 *   rowdot:     h[m][r] = scale * sum_k a[m][k] w(r, k);   a [M][K] bf16, h [M][4] fp32, w fp32 as [4][K] (w_is_k_by_4 = 0:
 *               `down`) or [K][4] (1: `up`, for the gradient of h).  K % 8 == 0.
 *   rank4_add:  y[m][n] = base[m][n] + sum_r h[m][r] w(r, n);   base (may be NULL) / y [M][N] bf16, w fp32 as [N][4]
 *               (w_is_n_by_4 = 1: `up`) or [4][N] (0: `down`, for the gradient of x).  N % 8 == 0.
 *   colreduce:  g(r, j) = scale * sum_m a[m][j] v[m][r];   a [M][J] bf16, v [M][4] fp32, g fp32 as [4][J] (g_is_j_by_4 = 0:
 *               d down) or [J][4] (1: d up); scratch = gd_nn_lora_colreduce_scratch_floats(M, J) floats.  Fixed summation
 *               order (row chunks, then chunk order): bitwise reproducible, no atomics. */
int gd_nn_lora_rowdot(void* stream, const void* a, const float* w, float* h, int64_t M, int K, float scale, int w_is_k_by_4);
int gd_nn_lora_rank4_add(void* stream, const float* h, const float* w, const void* base, void* y, int64_t M, int N, int w_is_n_by_4);
/* rowdot + rank4_add of a row by one wave, ONE launch (bit-identical to the two calls): h[m] = scale * a[m] . w1, then
 * y[m] = base[m] + h[m] . w2; h (may be NULL) receives the [M][4] intermediate the backward pass / the weight gradients need.
 * backward = 0 (LoRAAttnProcessor forward, lora_unet.py:415-422): a = x [M][K], w1 = down [4][K], w2 = up [N][4], base = the frozen
 * projection.  backward = 1 (its autograd backward w.r.t. x): a = dy [M][K], w1 = up [K][4], w2 = down [4][N], base = the frozen
 * projection's own input gradient dy W (or NULL): dx leaves complete, no separate add of the two branches' gradients. */
int gd_nn_lora_row_fused(void* stream, const void* a, const float* w1, const float* w2, const void* base, float* h, void* y,
                         int64_t M, int K, int N, float scale, int backward);
size_t gd_nn_lora_colreduce_scratch_floats(int64_t M, int J);
int gd_nn_lora_colreduce(void* stream, const void* a, const float* v, float* scratch, float* g, int64_t M, int J, float scale,
                         int g_is_j_by_4);
/* Both weight gradients of one adapter in one launch per stage: d_up [N][4] = sum_m dy[m][n] hs[m][r] (hs = scale * down(x) as
 * rowdot returned it), d_down [4][K] = sum_m x[m][k] dh[m][r] (dh = scale * dy @ up); same summation order as two colreduce calls. */
size_t gd_nn_lora_colreduce_pair_scratch_floats(int64_t M, int N, int K);
int gd_nn_lora_colreduce_pair(void* stream, const void* dy, const float* hs, const void* x, const float* dh, float* scratch,
                              float* d_up, float* d_down, int64_t M, int N, int K);
/* The same with the two results ADDED onto d_up / d_down when accumulate != 0 (the sums are formed first, in the same fixed
 * order, then added once): the destination is a gradient buffer that accumulates over backward passes like torch's .grad --
 * here slices of the flat gradient buffer of garmentdreamer_amd.flat_adam.FlatAdam, so the 256 adapter gradients of a UNet
 * backward never become 256 tensors on the host (netf/trainer.py:252-256: loss.backward(); lora_unet_optimizer.step()). */
int gd_nn_lora_colreduce_pair_into(void* stream, const void* dy, const float* hs, const void* x, const float* dh, float* scratch,
                                   float* d_up, float* d_down, int64_t M, int N, int K, int accumulate);
/* The weight gradients of MANY adapters in one launch per stage (round 6; replaces one gd_nn_lora_colreduce_pair_into per adapted
 * projection -- 256 x 2 launches of ~5 us per LoRA UNet backward -- by 2 launches).  _group_desc fills ONE host-side table entry
 * (gd_nn_lora_colreduce_group_entry_bytes() bytes) for an adapter, arguments as gd_nn_lora_colreduce_pair_into, and returns that
 * adapter's grid extents {stage-1 x, stage-1 y, stage-2 x} in grid_xyz[3]; _group_launch takes the HOST table and the MAXIMA of
 * the extents and launches 32 adapters at a time with their entries by value in the kernel arguments (no device table, no copy:
 * safe inside a hipGraph capture).  Same kernels' bodies and summation orders: bit-identical results. */
size_t gd_nn_lora_colreduce_group_entry_bytes(void);
int gd_nn_lora_colreduce_group_desc(void* entry, const void* dy, const float* hs, const void* x, const float* dh, float* scratch,
                                    float* d_up, float* d_down, int64_t M, int N, int K, int accumulate, int* grid_xyz);
int gd_nn_lora_colreduce_group_launch(void* stream, const void* table_host, int n_entries, int grid1_x, int grid1_y, int grid2_x);
const char* gd_nn_lora_last_error(void);

const char* gd_nn_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
