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
 * stats_ws: N*G*2 doubles of scratch (zeroed by the call); mean_rstd: N*G*2 floats out
 * (saved for backward).  Replaces F.group_norm + F.silu (two kernels + two NCHW<->NHWC copies in
 * PyTorch-ROCm's native path). */
int gd_nn_groupnorm_silu_forward(void* stream, const void* x, void* y, const void* gamma, const void* beta, int N,
                                 int HW, int C, int G, float eps, int apply_silu, double* stats_ws,
                                 float* mean_rstd);

/* dx for the same op (weights frozen: no dgamma / dbeta, as in the guidance where
 * requires_grad_(False) is set on every VAE/UNet parameter,
 * stable_diffusion_guidance.py:99-102).  dy: gradient w.r.t. y. */
int gd_nn_groupnorm_silu_backward(void* stream, const void* x, const void* dy, const void* gamma,
                                  const void* beta, const float* mean_rstd, void* dx, int N, int HW, int C, int G,
                                  int apply_silu, double* stats_ws);

size_t gd_nn_groupnorm_ws_bytes(int N, int G);

const char* gd_nn_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
