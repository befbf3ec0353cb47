"""One rank of tests/test_configs_gpu.py::test_config3_*: the view-sharded SDS loop on the real HIP rasterizer.

Launched by the test as ``WORLD_SIZE`` processes that share cuda:0 (``GD_DIST_BACKEND=gloo``); runs four iterations
(global steps 399..402, crossing the densify/prune event at 400) of 4 views @128^2 over 20 000 Gaussians with a
reduced-width SD-2.1-shaped UNet/VAE and writes this rank's state to ``sys.argv[1]``.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import garmentdreamer_amd  # noqa: E402,F401  (before the HIP runtime starts)
import torch  # noqa: E402

from garmentdreamer_amd import cameras as gcam  # noqa: E402
from garmentdreamer_amd import dist as gdist  # noqa: E402
from garmentdreamer_amd.gaussian_model import GaussianModel  # noqa: E402
from garmentdreamer_amd.guidance import sd21  # noqa: E402
from garmentdreamer_amd.guidance.stable_diffusion_guidance import PromptEmbeddings, StableDiffusionGuidance  # noqa: E402
from garmentdreamer_amd.scene import synthetic_gaussians  # noqa: E402
from garmentdreamer_amd.sds_loop import SDSLoop  # noqa: E402

V_TOTAL, P, HW, FIRST_STEP, N_STEPS = 4, 20000, 128, 399, 4
FULL = os.environ.get("GD_TEST_CFG") == "full"      # configs[3] at its stated shape: 8 views x 100k @512^2, full bf16 nets
if FULL:
    V_TOTAL, P, HW, FIRST_STEP, N_STEPS = 8, 100000, 512, 10, 2


def main():
    rk, _lr, ws = gdist.init_from_env()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if FULL:
        # guidance_scale: the classifier-free term multiplies (eps_text - eps_uncond), which at random-init weights is
        # bf16 rounding noise of the UNet -- and a UNet call on 8 latents picks other tiles than one on 16.  With the
        # reference's 100 the 2x4-view and 1x8-view buckets agree only to cosine 0.92 for that reason (measured);
        # GD_TEST_CFG_SCALE keeps the amplification at the level the comparison is about
        guidance = StableDiffusionGuidance({"guidance_scale": float(os.environ.get("GD_TEST_CFG_SCALE", "7.5")),
                                            "grad_clip": [0, 1.5, 2.0, 1000], "use_hip_graphs": True}, device=dev)
    else:
        with torch.device(dev):
            unet = sd21.init_random_(sd21.UNet2DConditionModel(block_out_channels=(64, 128, 256, 256),
                                                               attention_head_dim=(1, 2, 4, 4)))
            vae = sd21.init_random_(sd21.AutoencoderKLEncoder(block_out_channels=(32, 64, 128, 128)), 1)
        # fp32 UNet / VAE (torch ops): bf16 results depend on the batch composition (4 vs 8 UNet samples pick different
        # GEMM / conv tiles), which would blur a test whose subject is the sharded rasterizer + collectives + Adam path
        guidance = StableDiffusionGuidance({"guidance_scale": 7.5, "grad_clip": [0, 1.5, 2.0, 1000],
                                            "half_precision_weights": False}, device=dev, unet=unet, vae=vae)
    gm = GaussianModel.from_activated(synthetic_gaussians(P, seed=2), device=dev)
    loop = SDSLoop(gm, guidance, PromptEmbeddings.random(dev), torch.ones(3, device=dev), densify_seed=123,
                   batch_invariant=os.environ.get("GD_TEST_BATCH_INVARIANT") == "1")
    loop.global_step = FIRST_STEP
    view_ids = gdist.shard_views(V_TOTAL, rk, ws)
    g = torch.Generator().manual_seed(11)
    noise = torch.randn(N_STEPS, V_TOTAL, 4, 64, 64, generator=g).to(dev)
    vnoise = torch.randn(N_STEPS, V_TOTAL, 4, 64, 64, generator=g).to(dev)
    ts = torch.randint(20, 981, (N_STEPS, V_TOTAL), generator=g).to(dev)
    rec = {"grads": [], "radii": [], "P_history": [gm.get_xyz.shape[0]], "densified": []}
    for s in range(N_STEPS):
        batch = gcam.orbit_batch(V_TOTAL, height=HW, width=HW, azimuth_offset_deg=10.0 * s, view_ids=view_ids)
        if loop.global_step == 400:
            rec["flat_before_densify"] = gm._flat.detach().cpu().clone()
        bucket_before = gm.grad_bucket
        out = loop.step(batch, noise=noise[s, view_ids], timesteps=ts[s, view_ids], vae_noise=vnoise[s, view_ids])
        torch.cuda.synchronize()
        if not out["densified"]:
            rec["grads"].append(bucket_before.detach().cpu().clone())   # after the all-reduce, as Adam consumed it
            rec["radii"].append(gm.max_radii2D.detach().cpu().clone())
        rec["densified"].append(bool(out["densified"]))
        rec["P_history"].append(gm.get_xyz.shape[0])
    rec.update(backend=(torch.distributed.get_backend() if gdist.is_dist() else None), world_size=gdist.world_size())
    rec.update(P=gm.get_xyz.shape[0], flat=gm._flat.detach().cpu(), exp_avg=gm._exp_avg.detach().cpu(),
               exp_avg_sq=gm._exp_avg_sq.detach().cpu(), max_radii2D=gm.max_radii2D.detach().cpu(),
               xyz_gradient_accum=gm.xyz_gradient_accum.detach().cpu(), denom=gm.denom.detach().cpu())
    # only the pre-densify gradients are compared across world sizes
    n_pre = rec["densified"].index(True) if True in rec["densified"] else len(rec["grads"])
    rec["grads"], rec["radii"] = rec["grads"][:n_pre], rec["radii"][:n_pre]
    torch.save(rec, sys.argv[1])
    gdist.barrier()
    if gdist.is_dist():
        torch.distributed.destroy_process_group()
    print(f"rank {rk}/{ws}: P {rec['P_history']}, densified {rec['densified']}", flush=True)


if __name__ == "__main__":
    main()
