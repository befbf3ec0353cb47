"""Checks against golden vectors captured from the reference's importable Python
(tests/golden/make_golden.py; the reference itself is not needed at test time)."""
import json
import os

import numpy as np
import pytest
import torch

from garmentdreamer_amd import cameras as gcam
from tests import helpers as h

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_camera_matrices_match_reference_camera_bit_for_bit():
    d = np.load(os.path.join(G, "cameras.npz"))
    for i, (az, el, dist, fovy, H, W, fovx) in enumerate(d["params"]):
        c2w = gcam.c2w_3dgs(float(az), float(el), float(dist))
        np.testing.assert_array_equal(c2w.numpy(), d["c2w"][i])
        cam = gcam.Camera(torch.tensor(d["c2w"][i]), float(fovy), int(H), int(W), data_device="cpu")
        np.testing.assert_array_equal(cam.world_view_transform.numpy(), d["wvt"][i])
        np.testing.assert_array_equal(cam.full_proj_transform.numpy(), d["full"][i])
        np.testing.assert_array_equal(cam.camera_center.numpy(), d["center"][i])
        assert cam.FoVx == fovx
        # the row-vector convention the kernels index: last column of the view matrix is (0,0,0,1)
        np.testing.assert_allclose(cam.world_view_transform.numpy()[:, 3], [0, 0, 0, 1], atol=1e-6)


def test_camera_batch_packs_same_matrices():
    batch = gcam.orbit_batch(4, height=64, width=64)
    cams = [gcam.Camera(batch["c2w_3dgs"][i], batch["fovy"][i], 64, 64, data_device="cpu") for i in range(4)]
    cb = gcam.CameraBatch(cams, "cpu")
    for i, c in enumerate(cams):
        np.testing.assert_array_equal(cb.viewmatrix[i].numpy(), c.world_view_transform.numpy())
        np.testing.assert_array_equal(cb.projmatrix[i].numpy(), c.full_proj_transform.numpy())
        np.testing.assert_array_equal(cb.campos[i].numpy(), c.camera_center.numpy())
    assert cb.tanfovx == [c.tanfovx for c in cams]


def test_oracle_sh_colours_match_reference_eval_sh():
    """Oracle's computeColorFromSH restatement vs the reference's eval_sh (+0.5, clamp) for degrees 0-3."""
    d = np.load(os.path.join(G, "sh_eval.npz"))
    dirs = d["dirs"]
    P = dirs.shape[0]
    for deg in range(4):
        # put Gaussian g at position dirs[g]*2 with the camera at the origin -> direction = dirs[g]
        cam = h.make_camera(H=32, W=32)
        inp = h.raster_inputs(P=P, H=32, W=32)
        inp["means3D"] = (dirs * 2.0).astype(np.float32)
        inp["sh"] = d[f"sh{deg}"]
        inp["degree"] = deg
        inp["campos"] = np.zeros(3, np.float32)
        # make every Gaussian pass culling irrespective of the camera: use an orthonormal view that
        # looks down +z from far behind, huge image so rects are non-empty
        view = np.eye(4, dtype=np.float32)
        view[3, 2] = 10.0  # row-vector convention: translation in the last ROW
        inp["viewmatrix"] = view
        proj = cam.projection_matrix.numpy()
        inp["projmatrix"] = (view @ proj).astype(np.float32)
        st = h.oracle_forward(inp)
        vis = st.radii > 0
        assert vis.sum() > P // 2
        np.testing.assert_allclose(st.rgb[vis], d[f"rgb{deg}"][vis], rtol=2e-5, atol=2e-6)
        np.testing.assert_array_equal(st.clamped[vis].astype(bool), (d[f"raw{deg}"] < 0)[vis])


def test_python_op_marshals_like_the_reference(monkeypatch):
    """Our autograd.Function hands the native layer the same 19 / 24 positional arguments (kinds,
    shapes, order) and maps the native gradient tuple to the same inputs as the reference's op."""
    with open(os.path.join(G, "marshalling.json")) as f:
        gold = json.load(f)
    import garmentdreamer_amd.diff_gaussian_rasterization as dgr
    rec = {}
    P, H, W, M = 5, 8, 8, 1

    def kind(a):
        if isinstance(a, torch.Tensor):
            return ["tensor", list(a.shape), str(a.dtype).replace("torch.", "")]
        return [type(a).__name__, a if isinstance(a, (int, float, bool)) else None]

    def fwd(*args):
        rec["forward_args"] = [kind(a) for a in args]
        return (7, torch.zeros(3, H, W), torch.zeros(1, H, W), torch.zeros(1, H, W), torch.zeros(P, dtype=torch.int32),
                torch.zeros(3, dtype=torch.uint8), torch.zeros(4, dtype=torch.uint8), torch.zeros(5, dtype=torch.uint8))

    def bwd(*args):
        rec["backward_args"] = [kind(a) for a in args]
        shapes = [(P, 3), (P, 3), (P, 1), (P, 3), (P, 6), (P, M, 3), (P, 3), (P, 4)]
        return tuple(torch.full(s, float(i + 1)) for i, s in enumerate(shapes))

    monkeypatch.setattr(dgr._C, "rasterize_gaussians", fwd)
    monkeypatch.setattr(dgr._C, "rasterize_gaussians_backward", bwd)
    rs = dgr.GaussianRasterizationSettings(H, W, 0.5, 0.6, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0,
                                           torch.zeros(3), False, False)
    leaves = dict(means3D=torch.zeros(P, 3), means2D=torch.zeros(P, 3), opacities=torch.zeros(P, 1),
                  shs=torch.zeros(P, M, 3), scales=torch.zeros(P, 3), rotations=torch.zeros(P, 4))
    for v in leaves.values():
        v.requires_grad_(True)
    color, radii, depth, alpha = dgr.GaussianRasterizer(rs)(**leaves)
    (color.sum() + depth.sum() + alpha.sum()).backward()
    assert list(dgr.GaussianRasterizationSettings._fields) == gold["settings_fields"]
    assert rec["forward_args"] == gold["forward_args"]
    assert rec["backward_args"] == gold["backward_args"]
    got = {n: float(leaves[n].grad.flatten()[0]) for n in leaves}
    assert got == gold["grad_sentinel_by_input"]
    assert color.shape == (3, H, W) and radii.dtype == torch.int32 and depth.shape == (1, H, W)


def _pins():
    return np.load(os.path.join(G, "raster_pins.npz"))


def test_oracle_cov3D_matches_reference_build_covariance():
    """computeCov3D of the C oracle (forward.cu:118-152) vs the reference's Python
    ``build_covariance_from_scaling_rotation`` (scene/gaussian_model.py:27-31), fp32 tolerance: pins the
    S·R / GLM-transposition order and the (xx, xy, xz, yy, yz, zz) packing."""
    d = _pins()
    s, q = d["cov_scales"], d["cov_quats"]
    N = s.shape[0]
    for mi, mod in enumerate(d["cov_mods"]):
        inp = h.raster_inputs(P=N, H=64, W=64)
        # spread the Gaussians on a small sphere in front of the default camera so that none is culled
        inp["means3D"] = (d["ndc_points"][:N] * 0.3).astype(np.float32)
        inp["scales"], inp["rotations"], inp["scale_modifier"] = s, q, float(mod)
        st = h.oracle_forward(inp)
        vis = st.radii > 0
        assert vis.sum() >= N - 4, vis.sum()
        ref = d["cov3D"][mi]
        scale = np.abs(ref).max(axis=1, keepdims=True)
        err = np.abs(st.cov3D[vis] - ref[vis]) / scale[vis]
        assert err.max() < 4e-6, err.max()   # a few fp32 ulps of the largest entry (different summation order)


def test_oracle_projection_matches_reference_geom_transform_points():
    """transformPoint4x4 + 1/(w + 1e-7) + ndc2Pix of the C oracle (forward.cu:193-196, auxiliary.h:41-44) vs the
    reference's ``geom_transform_points`` (utils/graphics_utils.py:22-29) on the 20 fixture cameras."""
    d = _pins()
    cams = np.load(os.path.join(G, "cameras.npz"))
    pts = d["ndc_points"]
    N = pts.shape[0]
    checked = 0
    for i, (az, el, dist, fovy, H, W, fovx) in enumerate(cams["params"]):
        H, W = int(H), int(W)
        inp = h.raster_inputs(P=N, H=H, W=W)
        inp["means3D"] = pts
        inp["viewmatrix"], inp["projmatrix"] = cams["wvt"][i], cams["full"][i]
        inp["tanfovx"], inp["tanfovy"] = float(np.tan(fovx * 0.5)), float(np.tan(fovy * 0.5))
        inp["campos"] = cams["center"][i]
        st = h.oracle_forward(inp)
        vis = st.radii > 0
        ndc = d["ndc"][i].astype(np.float64)
        pix = np.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], 1)
        err = np.abs(st.means2D[vis] - pix[vis])
        # |ndc| <= ~1.3 -> one fp32 ulp of ndc is 1.2e-7, times S/2 pixels
        assert err.max() <= 4e-7 * max(H, W) + 1e-5, (i, err.max())
        # depth the keys are built from = view-space z = w of the projection (row-vector convention)
        w = (np.concatenate([pts, np.ones((N, 1), np.float32)], 1).astype(np.float64) @ cams["full"][i].astype(np.float64))[:, 3]
        np.testing.assert_allclose(st.depths[vis], w[vis], rtol=2e-6, atol=1e-6)
        checked += int(vis.sum())
    assert checked > 10 * N


# ---- SDS guidance algebra against the REFERENCE'S OWN code (tests/golden/make_golden_guidance.py: the reference's
# StableDiffusionGuidance.__call__ / compute_grad_sds / PromptProcessorOutput run with stub networks) ----
def _guidance_pins():
    return np.load(os.path.join(G, "guidance_pins.npz"))


class _PinUNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))

    def forward(self, x, t, encoder_hidden_states):
        from tests.golden import stub_nets
        return stub_nets.unet_fn(x, t, encoder_hidden_states)


class _PinVAE(torch.nn.Module):
    def __init__(self):
        super().__init__()
        from garmentdreamer_amd.guidance import sd21
        self.config = sd21._VAEConfig()
        self.w = torch.nn.Parameter(torch.ones(1))

    def encode(self, x):
        from garmentdreamer_amd.guidance import sd21
        from tests.golden import stub_nets
        m = stub_nets.vae_mean(x)
        return sd21._EncodeOutput(sd21.DiagonalGaussianDistribution(torch.cat([m, torch.full_like(m, -30.0)], 1)))


def _sub_and_moments(v):
    v = v.detach()
    return v[:, :, 1::4, 2::4].numpy(), np.array([v.double().sum().item(), v.double().abs().sum().item(),
                                                  (v.double() ** 2).sum().item()])


def test_guidance_call_reproduces_the_reference_guidance_code():
    """``StableDiffusionGuidance.__call__`` + ``compute_grad_sds`` (classifier-free and Perp-Neg combination, the three
    w(t) strategies, clipping, the reparameterised loss and its per-batch normalisation, the gradient that reaches the
    rendered image) against outputs of the reference's own functions run on the same stub networks, timesteps and noise
    (stable_diffusion_guidance.py:185-276,374-448; prompt_processors/base.py:52-160)."""
    from garmentdreamer_amd.guidance.stable_diffusion_guidance import PromptEmbeddings, StableDiffusionGuidance
    z = _guidance_pins()
    assert z["cfg100_clip/min_max_step"].tolist() == [20, 980]
    el, az, dist = (torch.from_numpy(z[k]) for k in ("elevation", "azimuth", "camera_distances"))
    t, noise = torch.from_numpy(z["t"]), torch.from_numpy(z["noise"])
    for name in ("cfg100_clip", "cfg7p5_uniform", "cfg20_fantasia", "perpneg7p5"):
        scale, clip = (float(v) for v in z[name + "/cfg"])
        gd = StableDiffusionGuidance({"guidance_scale": scale, "grad_clip": None, "half_precision_weights": False,
                                      "weighting_strategy": str(z[name + "/weighting"])},
                                     device="cpu", unet=_PinUNet(), vae=_PinVAE())
        gd.grad_clip_val = None if clip < 0 else clip
        assert (gd.min_step, gd.max_step) == (20, 980)
        prompt = PromptEmbeddings(torch.from_numpy(z[name + "/text_vd"]), torch.from_numpy(z[name + "/uncond_vd"]))
        prompt.use_perp_neg = name.startswith("perpneg")
        rgb = torch.from_numpy(z["rgb"]).clone().requires_grad_(True)
        seen = {}
        inner = gd.compute_grad_sds

        def spy(*a, **k):
            gr, u = inner(*a, **k)
            seen["grad"], seen["u"] = gr, u
            return gr, u
        gd.compute_grad_sds = spy
        out = gd(rgb, prompt, el, az, dist, noise=noise, timesteps=t, vae_noise=torch.zeros_like(noise))
        out["loss_sds"].backward()
        u = seen["u"]
        assert torch.allclose(u["text_embeddings"], torch.from_numpy(z[name + "/text_embeddings"]), rtol=0, atol=1e-6), name
        if prompt.use_perp_neg:
            assert np.allclose(u["neg_guidance_weights"].numpy(), z[name + "/neg_guidance_weights"], rtol=1e-6, atol=1e-7)
        for key, val in (("latents_noisy", u["latents_noisy"]), ("noise_pred", u["noise_pred"]), ("grad_sds", seen["grad"])):
            sub, mom = _sub_and_moments(val)
            ref_sub, ref_mom = z[f"{name}/{key}_sub"], z[f"{name}/{key}_moments"]
            scale_ = np.abs(ref_sub).max()
            assert np.abs(sub - ref_sub).max() <= 2e-5 * scale_, (name, key, np.abs(sub - ref_sub).max(), scale_)
            assert np.allclose(mom, ref_mom, rtol=2e-5), (name, key)
        assert abs(out["loss_sds"].item() - float(z[name + "/loss_sds"])) <= 2e-5 * float(z[name + "/loss_sds"]), name
        assert abs(out["grad_norm"].item() - float(z[name + "/grad_norm"])) <= 2e-5 * float(z[name + "/grad_norm"]), name
        ref_g = z[name + "/dloss_drgb"]
        assert np.abs(rgb.grad.numpy() - ref_g).max() <= 5e-5 * np.abs(ref_g).max(), name


def test_guidance_sjc_branch_reproduces_the_reference_code():
    """``use_sjc``: ``compute_grad_sjc`` (variance-exploding perturbation, the scaled UNet input, both forms of the
    gradient, the Perp-Neg combination) and ``__call__`` around it against outputs of the reference's own functions
    (stable_diffusion_guidance.py:278-372, 409-412; tests/golden/make_golden_guidance_sjc.py)."""
    from garmentdreamer_amd.guidance.stable_diffusion_guidance import PromptEmbeddings, StableDiffusionGuidance
    z0, z = _guidance_pins(), np.load(os.path.join(G, "guidance_sjc_pins.npz"))
    el, az, dist = (torch.from_numpy(z0[k]) for k in ("elevation", "azimuth", "camera_distances"))
    t, noise = torch.from_numpy(z["t"]), torch.from_numpy(z["noise"])
    for name in ("sjc_varred", "sjc_plain_clip", "sjc_perpneg"):
        scale, clip, var_red = (float(v) for v in z[name + "/cfg"])
        gd = StableDiffusionGuidance({"guidance_scale": scale, "grad_clip": None, "half_precision_weights": False,
                                      "use_sjc": True, "var_red": bool(var_red)},
                                     device="cpu", unet=_PinUNet(), vae=_PinVAE())
        gd.grad_clip_val = None if clip < 0 else clip
        prompt = PromptEmbeddings(torch.from_numpy(z[name + "/text_vd"]), torch.from_numpy(z[name + "/uncond_vd"]))
        prompt.use_perp_neg = name.endswith("perpneg")
        rgb = torch.from_numpy(z0["rgb"]).clone().requires_grad_(True)
        seen = {}
        inner = gd.compute_grad_sjc

        def spy(*a, **k):
            gr, u = inner(*a, **k)
            seen["grad"], seen["u"] = gr, u
            return gr, u
        gd.compute_grad_sjc = spy
        out = gd(rgb, prompt, el, az, dist, noise=noise, timesteps=t, vae_noise=torch.zeros_like(noise))
        out["loss_sds"].backward()
        u = seen["u"]
        for key, val in (("latents_noisy", u["latents_noisy"]), ("noise_pred", u["noise_pred"]), ("grad", seen["grad"])):
            sub, mom = _sub_and_moments(val)
            ref_sub, ref_mom = z[f"{name}/{key}_sub"], z[f"{name}/{key}_moments"]
            scale_ = np.abs(ref_sub).max()
            assert np.abs(sub - ref_sub).max() <= 2e-5 * scale_, (name, key, np.abs(sub - ref_sub).max(), scale_)
            assert np.allclose(mom, ref_mom, rtol=5e-5), (name, key, mom, ref_mom)
        assert abs(out["loss_sds"].item() - float(z[name + "/loss_sds"])) <= 2e-5 * float(z[name + "/loss_sds"]), name
        assert abs(out["grad_norm"].item() - float(z[name + "/grad_norm"])) <= 2e-5 * float(z[name + "/grad_norm"]), name
        ref_g = z[name + "/dloss_drgb"]
        assert np.abs(rgb.grad.numpy() - ref_g).max() <= 5e-5 * np.abs(ref_g).max(), name


def test_guidance_eval_previews_reproduce_the_reference_code():
    """``guidance_eval=True``: nearest-of-50 timestep search, one-step and full eta = 1 sampling per sample, the embedding
    rows picked per sample (plain and Perp-Neg), ``max_items_eval``, ``decode_latents`` and the preview dict with its texts
    against the reference's own ``guidance_eval`` / ``get_noise_pred`` (stable_diffusion_guidance.py:436-446, 452-579) run
    on the stub networks with an independently written DDIM step (tests/golden/make_golden_guidance_eval.py)."""
    from garmentdreamer_amd.guidance.stable_diffusion_guidance import PromptEmbeddings, StableDiffusionGuidance
    from tests.golden import stub_nets
    import types
    z0, z = _guidance_pins(), np.load(os.path.join(G, "guidance_eval_pins.npz"))
    el, az, dist = (torch.from_numpy(z0[k]) for k in ("elevation", "azimuth", "camera_distances"))

    class VAE(_PinVAE):
        def decode(self, lat):
            return types.SimpleNamespace(sample=stub_nets.vae_decode(lat))
    for name in ("eval_cfg", "eval_perpneg"):
        gd = StableDiffusionGuidance({"guidance_scale": 7.5, "half_precision_weights": False,
                                      "max_items_eval": int(z[name + "/max_items_eval"])},
                                     device="cpu", unet=_PinUNet(), vae=VAE())
        prompt = PromptEmbeddings(torch.from_numpy(z[name + "/text_vd"]), torch.from_numpy(z[name + "/uncond_vd"]))
        prompt.use_perp_neg = name.endswith("perpneg")
        noise = torch.from_numpy(z[name + "/noise"])
        out = gd(torch.from_numpy(z0["rgb"]).clone(), prompt, el, az, dist, noise=noise, timesteps=torch.from_numpy(z[name + "/t"]),
                 vae_noise=torch.zeros_like(noise), guidance_eval=True, eval_generator=torch.Generator().manual_seed(99))
        ev = out["eval"]
        assert ev["bs"] == int(z[name + "/bs"])
        assert np.allclose(np.array([float(v) for v in ev["noise_levels"]]), z[name + "/noise_levels"], rtol=0, atol=1e-6)
        assert list(ev["texts"]) == [str(t) for t in z[name + "/texts"]]
        for key in ("imgs_noisy", "imgs_1step", "imgs_1orig", "imgs_final"):
            assert list(ev[key].shape) == z[f"{name}/{key}_shape"].tolist(), key
            got = ev[key][:, 3::32, 5::32, :].numpy()
            # images in [0, 1]; the multi-step result goes through up to five guided sampler steps (fp32 here, the fixture's
            # scheduler in fp64, Perp-Neg projections in between): measured 1.3e-3
            bar = 5e-3 if key == "imgs_final" else 2e-4
            assert np.abs(got - z[f"{name}/{key}_sub"]).max() < bar, (name, key, np.abs(got - z[f"{name}/{key}_sub"]).max())
    # the restated encoder-only VAE has no decoder: a clear error, not an AttributeError deep inside
    gd = StableDiffusionGuidance({"half_precision_weights": False}, device="cpu", unet=_PinUNet(), vae=_PinVAE())
    with pytest.raises(RuntimeError, match="decoder"):
        gd.decode_latents(torch.zeros(1, 4, 8, 8))


def test_direction_rules_match_the_reference_prompt_processor():
    """front / side / back / overhead selection at the thresholds and across the azimuth wrap, against the index the
    reference's own DirectionConfig lambdas assign (prompt_processors/base.py:222-290 through get_text_embeddings)."""
    from garmentdreamer_amd.guidance.stable_diffusion_guidance import PromptEmbeddings
    z = _guidance_pins()
    p = PromptEmbeddings(torch.arange(4.0).view(4, 1, 1), torch.arange(4.0).view(4, 1, 1))
    el, az = torch.from_numpy(z["dir/elevation"]), torch.from_numpy(z["dir/azimuth"])
    got = p.get_text_embeddings(el, az, torch.ones_like(el))[: el.numel(), 0, 0].long().tolist()
    assert got == z["dir/index"].tolist()


def test_vsd_train_step_reproduces_the_reference_train_step():
    """NeTF-stage ``StableDiffusion.train_step`` (sd_vsd_utils.py:131-218: noise injection, classifier-free combination
    of the frozen UNet, v -> eps of the LoRA UNet's output, w(t), ``SpecifyGradient`` and its 1 / batch backward, the
    pseudo loss; t5 annealing and direction-dependent embeddings) against outputs of the reference's own function run on
    the stub networks of tests/golden/stub_nets.py with the same timestep and noise (tests/golden/make_golden_vsd.py)."""
    from garmentdreamer_amd.guidance.sd_vsd import StableDiffusionVSD
    from tests.golden import stub_nets
    z = np.load(os.path.join(G, "vsd_pins.npz"))
    for k in range(3):
        p = f"case{k}/"
        lo, hi, scale, t5, hor = (float(v) for v in z[p + "cfg"])
        gd = StableDiffusionVSD("cpu", fp16=False, t_range=(lo, hi), unet=_PinUNet(), vae=_PinVAE())
        assert (gd.min_step, gd.max_step) == (int(1000 * lo), int(1000 * hi))
        gd.set_text_embeds(*(torch.from_numpy(z[p + "emb_" + n]) for n in ("pos", "neg", "front", "side", "back")))
        pose = torch.from_numpy(z[p + "pose"])
        shading = str(z[p + "shading"])
        # the stub VAE sees the image through 8 x 8 average pooling only: a blocky image with the stored block means
        img = torch.from_numpy(z[p + "rgb_pooled"]).repeat_interleave(8, 2).repeat_interleave(8, 3).clone().requires_grad_(True)
        q = lambda x, t, text, c=None, shading=None: stub_nets.q_unet_fn(x, t, text, c, shading)    # noqa: E731
        noise = torch.from_numpy(z[p + "noise"])
        loss, pseudo, latents = gd.train_step(img, guidance_scale=scale, q_unet=q, pose=pose, shading=shading, t5=bool(t5),
                                              hors=None if hor < 0 else [hor], noise=noise, timesteps=torch.from_numpy(z[p + "t"]),
                                              vae_noise=torch.zeros_like(noise))
        loss.backward()
        assert torch.allclose(latents.detach(), torch.from_numpy(z[p + "latents"]), rtol=1e-5, atol=1e-6)
        ref_loss, ref_pseudo = float(z[p + "loss"]), float(z[p + "pseudo_loss"])
        assert abs(loss.item() - ref_loss) <= 3e-5 * abs(ref_loss) + 1e-3, (k, loss.item(), ref_loss)
        assert abs(pseudo.item() - ref_pseudo) <= 1e-4 * abs(ref_pseudo) + 1e-4, (k, pseudo.item(), ref_pseudo)
        blk = img.grad.view(1, 3, 64, 8, 64, 8)[:, :, :, 0, :, 0].numpy()
        ref_blk = z[p + "dloss_drgb_block"]
        assert np.abs(blk - ref_blk).max() <= 5e-5 * np.abs(ref_blk).max(), k


def test_densify_and_prune_reproduces_the_reference_gaussian_model(monkeypatch):
    """``GaussianModel.densify_and_prune`` (gaussiansplatting/scene/gaussian_model.py:398-413 with clone / split /
    densification_postfix / prune and the Adam-moment surgery of ``cat_tensors_to_optimizer`` / ``_prune_optimizer``)
    against outputs of the reference's own class run on CPU (tests/golden/make_golden_densify.py): point count, order,
    every parameter and both Adam moments of every group, the reset statistics.  The build's op-for-op torch sequence
    is what is held to the fixture here; the HIP path is held to that sequence bit for bit on the GPU
    (tests/test_scene_gpu.py::test_native_densify_and_prune_matches_the_torch_sequence)."""
    from garmentdreamer_amd.gaussian_model import GROUPS, GaussianModel
    z = np.load(os.path.join(G, "densify_pins.npz"))
    for case in range(3):
        p = f"case{case}/"
        max_grad, min_opacity, extent, max_screen, percent_dense, sh_degree = (float(v) for v in z[p + "args"])
        gm = GaussianModel(int(sh_degree), device="cpu")
        t = lambda k: torch.from_numpy(z[p + k])        # noqa: E731
        gm._pack({n: t("in/" + n) for n in GROUPS}, {n: t("in/exp_avg/" + n) for n in GROUPS},
                 {n: t("in/exp_avg_sq/" + n) for n in GROUPS})
        gm.percent_dense = percent_dense
        gm.xyz_gradient_accum, gm.denom, gm.max_radii2D = t("in/accum").clone(), t("in/denom").clone(), t("in/max_radii2D").clone()
        zs = [torch.from_numpy(z[p + "z"])]
        monkeypatch.setattr(torch, "normal", lambda mean=None, std=None, generator=None: mean + std * zs[0])
        gm.densify_and_prune_torch(max_grad, min_opacity, extent, int(max_screen))
        monkeypatch.undo()
        cur, m, v = gm._current()
        for n in GROUPS:
            ref = z[p + "out/" + n]
            assert cur[n].shape[0] == ref.shape[0], (case, n, cur[n].shape, ref.shape)
            got = cur[n].detach().reshape(ref.shape).numpy()
            # children's positions go through a bmm (accumulation order): 1e-6; everything else is copied or elementwise
            tol = 2e-6 if n == "xyz" else 0.0
            if ref.size:
                assert np.abs(got - ref).max() <= tol * max(1.0, np.abs(ref).max()), (case, n, np.abs(got - ref).max())
            assert np.array_equal(m[n].reshape(ref.shape).numpy(), z[p + "out/exp_avg/" + n]), (case, n)
            assert np.array_equal(v[n].reshape(ref.shape).numpy(), z[p + "out/exp_avg_sq/" + n]), (case, n)
        assert np.array_equal(gm.xyz_gradient_accum.numpy(), z[p + "out/accum"])
        assert np.array_equal(gm.denom.numpy(), z[p + "out/denom"])
        assert np.array_equal(gm.max_radii2D.numpy(), z[p + "out/max_radii2D"])
