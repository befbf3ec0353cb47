"""Deterministic stand-ins for the SD-2.1 UNet and VAE encoder, shared by tests/golden/make_golden_guidance.py (which
drives the REFERENCE's guidance code with them, in the build container) and tests/test_golden_fixtures.py (which drives
the build's guidance with the same functions): the fixture then pins everything AROUND the networks -- noise injection,
classifier-free / Perp-Neg combination, w(t), clipping, the reparameterised loss and its normalisation -- to the
reference's own code.  Written for this purpose; nothing here comes from the reference."""
import torch

_MIX = torch.tensor([[0.9, -0.3, 0.2], [0.1, 0.8, -0.5], [-0.4, 0.2, 0.7], [0.3, 0.3, 0.3]])


def unet_fn(x, t, ctx):
    """eps(x, t, context): smooth in all three, different for different prompts."""
    c = ctx.float().mean(dim=1)[:, :4]                       # [B, 4]
    tt = t.float().view(-1, 1, 1, 1) / 1000.0
    return (torch.tanh(0.7 * x.float() + 0.05) * (1.0 + tt) + 0.2 * c.view(-1, 4, 1, 1) * torch.cos(3.0 * x.float())).to(x.dtype)


def vae_mean(imgs):
    """Posterior mean of the stub encoder for a [B,3,512,512] image in [-1, 1]: 8x average pooling + a 3 -> 4 channel mix."""
    m = torch.nn.functional.avg_pool2d(imgs.float(), 8)
    return torch.einsum("oc,bchw->bohw", _MIX.to(m), m).to(imgs.dtype)


def q_unet_fn(x, t, ctx, pose, shading):
    """Stand-in for the pose-conditioned LoRA UNet of the NeTF stage (v-prediction output)."""
    s = {"albedo": 1.0, "normal": 0.5, "textureless": 0.25}.get(shading, 2.0)
    return (0.6 * unet_fn(x, t, ctx).float() - 0.1 * x.float() + 0.05 * s * pose.float().mean(dim=1).view(-1, 1, 1, 1)).to(x.dtype)


_UNMIX = torch.tensor([[0.6, 0.1, -0.3, 0.2], [-0.2, 0.7, 0.1, 0.2], [0.3, -0.4, 0.5, 0.2]])


def vae_decode(latents):
    """Stand-in decoder for the guidance_eval previews: a 4 -> 3 channel mix, tanh, 8x nearest upsampling ([-1, 1] image)."""
    img = torch.tanh(torch.einsum("oc,bchw->bohw", _UNMIX.to(latents.float()), latents.float()))
    return torch.nn.functional.interpolate(img, scale_factor=8, mode="nearest").to(latents.dtype)
