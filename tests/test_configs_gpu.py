"""GPU tests at the shapes BASELINE.json's configs name (round-1 verdict: configs[2], [3] and [4] had no test).

configs[2]  100 000 Gaussians x 4 views @512^2 with the full-size SD-2.1 UNet/VAE: one graphed step vs the eager
            step (loss, parameter gradients, per-view viewspace gradients), and the rasterizer gradients of one of
            its views against the CPU oracle given the step's own dL/dimage.
configs[3]  the view-sharded loop on the REAL HIP rasterizer: 2 ranks (gloo, sharing this one GPU) x 2 views ==
            1 rank x 4 views; replicas stay bit-identical through a densify/prune event.  (RCCL itself needs two
            GPUs; the driver's multi-GPU bench runs it.)
configs[4]  the NeTF VSD iteration at full size (SD-2.1 UNet + LoRA UNet + VAE, 512^2), and the same step at reduced
            width against eager fp32 PyTorch.
"""
import math
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import helpers as h
from tests import parity_report

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cos(a, b):
    return F.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()


def _one_gradient_step(loop, batch, noise, t, vae_noise):
    """render -> guidance -> loss -> backward (no Adam): what SDSLoop.step does up to the optimizer."""
    gm = loop.gaussians
    out = loop.render_views(batch)
    for k in ("render", "depth_3dgs", "alpha"):
        out[k].retain_grad()
    g_out = loop.guidance(out["comp_rgb"], loop.prompt_utils, batch["elevation"], batch["azimuth"],
                          batch["camera_distances"], rgb_as_latents=False, guidance_eval=False, noise=noise,
                          timesteps=t, vae_noise=vae_noise)
    loss = g_out["loss_sds"] + (out["opacity"] ** 2 + 0.01).sqrt().mean()
    gm.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    return {"loss_sds": g_out["loss_sds"].detach().clone(), "flat_grad": gm.flat_grad.detach().clone(),
            "viewspace": out["viewspace_points"].grad.detach().clone(), "radii": out["radii"].clone(),
            "d_render": out["render"].grad.detach().clone(), "d_depth": out["depth_3dgs"].grad.detach().clone(),
            "d_alpha": None if out["alpha"].grad is None else out["alpha"].grad.detach().clone()}


def test_config2_100k_gaussians_4_views_full_nets_graph_vs_eager_and_oracle():
    _full_size_step_graph_vs_eager_and_oracle(V=4, check_view=0, label="configs[2]")


def test_config3_100k_gaussians_8_views_full_nets_graph_vs_eager_and_oracle():
    """BASELINE configs[3]'s workload at its stated shape on one GPU (what rank 0 of a 1-GPU run executes, and what the
    N ranks of the sharded run add up to): 8 views x 100 000 Gaussians @512^2 with the full-size SD-2.1 nets, one
    step; the rasterizer gradients of view 5 inside that step against the CPU oracle."""
    _full_size_step_graph_vs_eager_and_oracle(V=8, check_view=5, label="configs[3]")


def _full_size_step_graph_vs_eager_and_oracle(V, check_view, label):
    import argparse
    import bench
    from garmentdreamer_amd.cameras import Camera, CameraBatch
    from garmentdreamer_amd.diff_gaussian_rasterization import _C
    from garmentdreamer_amd.gaussian_model import GaussianModel
    from garmentdreamer_amd.guidance.stable_diffusion_guidance import PromptEmbeddings, StableDiffusionGuidance
    from garmentdreamer_amd.scene import synthetic_gaussians
    from garmentdreamer_amd.sds_loop import SDSLoop
    from oracle import gd_oracle
    dev = torch.device(DEV)
    P, HW = 100000, 512
    args = argparse.Namespace(views=V, gaussians=P, res=HW)
    gm = GaussianModel.from_activated(synthetic_gaussians(P, seed=0), device=dev)
    cfg = {"guidance_scale": 100.0, "grad_clip": [0, 1.5, 2.0, 1000]}
    g_graph = StableDiffusionGuidance({**cfg, "use_hip_graphs": True}, device=dev)
    g_eager = StableDiffusionGuidance({**cfg, "use_hip_graphs": False}, device=dev, unet=g_graph.unet, vae=g_graph.vae)
    prompt, bg = PromptEmbeddings.random(dev), torch.ones(3, device=dev)
    batch = bench.camera_batch(args, 0, list(range(V)))
    gen = torch.Generator(device=dev).manual_seed(5)
    noise = torch.randn(V, 4, 64, 64, device=dev, generator=gen)
    vn = torch.randn(V, 4, 64, 64, device=dev, generator=gen)
    t = torch.randint(20, 981, (V,), device=dev, generator=gen)
    res = {}
    for name, guid in (("eager", g_eager), ("graph", g_graph)):
        loop = SDSLoop(gm, guid, prompt, bg)
        guid.update_step(0, 0)
        for _ in range(2 if name == "graph" else 1):    # the second graphed call is a pure replay
            res[name] = _one_gradient_step(loop, batch, noise, t, vn)
    assert g_graph.cfg.use_hip_graphs, "hipGraph replay was switched off (capture failed or runtime flag missing)"
    e, g = res["eager"], res["graph"]
    # same kernels, same inputs (the rasterizer backward is atomic-free and bitwise reproducible): what can differ is
    # the order of the fp64 GroupNorm partial sums, amplified by guidance_scale = 100 on bf16 noise predictions
    d_loss = abs(float(e["loss_sds"]) - float(g["loss_sds"])) / abs(float(e["loss_sds"]))
    cos_flat, cos_vs = _cos(e["flat_grad"], g["flat_grad"]), _cos(e["viewspace"], g["viewspace"])
    cos_img = _cos(e["d_render"], g["d_render"])
    parity_report.record(f"{label} graph vs eager, 100k x {V} views, full SD-2.1", "step",
                         rel_dloss=d_loss, cos_flat_grad=cos_flat, cos_viewspace=cos_vs, cos_dL_dimage=cos_img)
    assert torch.equal(e["radii"], g["radii"])
    assert d_loss < 1e-3, d_loss
    assert cos_img > 0.9999 and cos_flat > 0.9999 and cos_vs > 0.9999, (cos_img, cos_flat, cos_vs)
    assert torch.isfinite(g["flat_grad"]).all() and float(g["flat_grad"].abs().max()) > 0

    # ---- rasterizer gradients of one view inside this step vs the CPU oracle, given the step's own dL/dimage ----
    k = check_view
    cams = [Camera(batch["c2w_3dgs"][i], batch["fovy"][i], HW, HW, data_device="cpu") for i in range(V)]
    cb = CameraBatch(cams, dev)
    with torch.no_grad():
        shs, opac, scales, rots = gm.activated()
    n = lambda x: x.detach().cpu().numpy()
    inp = dict(bg=n(bg), means3D=n(gm.get_xyz), colors_precomp=None, opacities=n(opac), scales=n(scales),
               rotations=n(rots), scale_modifier=1.0, cov3D_precomp=None, viewmatrix=n(cb.viewmatrix[k]),
               projmatrix=n(cb.projmatrix[k]), tanfovx=float(cb.tanfovx[k]), tanfovy=float(cb.tanfovy[k]),
               image_height=HW, image_width=HW, sh=n(shs), degree=0, campos=n(cb.campos[k]))
    st = h.oracle_forward(inp)
    gc, gd = n(g["d_render"][k]), n(g["d_depth"][k])
    ga = np.zeros((1, HW, HW), np.float32) if g["d_alpha"] is None else n(g["d_alpha"][k])
    ref = gd_oracle.backward(st, gc, gd, ga)
    a = h.to_torch(inp, DEV)
    out = _C.rasterize_gaussians(*a)
    R, color, depth, alpha, radii, geom, binning, img = out
    assert R == st.num_rendered and torch.equal(radii, g["radii"][k])
    tt = lambda x: torch.as_tensor(np.ascontiguousarray(x), device=DEV)
    (bg_, means3D, colors, op_, sc_, ro_, smod, cov, vm, pm, tx, ty, H, W, sh_, degree, campos, _, _) = a
    grads = _C.rasterize_gaussians_backward(bg_, means3D, radii, colors, sc_, ro_, smod, cov, vm, pm, tx, ty, tt(gc),
                                            tt(gd), tt(ga), sh_, degree, campos, geom, R, binning, img, alpha, False)
    torch.cuda.synchronize()
    names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
             "dL_drotations")
    from tests.test_raster_gpu import _check_grads
    for nm, gr in zip(names, grads):
        # the forward pass is bit-exact against the oracle, so the GPU's own alpha image gives the standard tolerance
        _check_grads(nm, gr, ref[nm], rtol=1e-3, atol_scale=5e-6,
                     case=f"{label} view {k} of the {V}-view step vs oracle (GPU alpha, SDS dL/dimage)")
    # ... and the gradient the LOOP saw for that view (batched launch) is the single-view one
    vs0 = g["viewspace"][k]
    s = float(grads[0].abs().max())
    assert float((vs0 - grads[0]).abs().max()) <= 2e-4 * s


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_workers(world, out_dir, extra_env=None, timeout=1500):
    port = _free_port()
    procs = []
    for rk in range(world):
        env = dict(os.environ, RANK=str(rk), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"),
                                       os.path.join(out_dir, f"w{world}_r{rk}.pt")], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-4000:]
    return [torch.load(os.path.join(out_dir, f"w{world}_r{rk}.pt"), map_location="cpu") for rk in range(world)]


def test_config3_two_ranks_share_one_gpu_real_rasterizer_matches_single_rank(tmp_path):
    """2 ranks x 2 views == 1 rank x 4 views through the HIP rasterizer, the guidance and the HIP Adam, with the
    flat in-place all-reduce (GaussianDreamer.py:189-191,268-279 is the coupling being reproduced); steps 399-402
    cross the densify_and_prune event at step 400 (:279-283) with the seeded generator."""
    single = _run_workers(1, str(tmp_path))[0]
    r0, r1 = _run_workers(2, str(tmp_path))
    # replicas: bit-identical state on both ranks after every step, densification included
    for k in ("flat", "exp_avg", "exp_avg_sq", "max_radii2D", "xyz_gradient_accum", "denom"):
        assert torch.equal(r0[k], r1[k]), k
    assert r0["P"] == r1["P"] and r0["P_history"] == r1["P_history"]
    assert r0["P_history"][0] != r0["P_history"][-1], "densify_and_prune did not change P"
    assert any(r0["densified"])
    # sharded == single rank, before the densify event (after it a borderline threshold may legitimately flip)
    for s in range(len(single["grads"])):
        a, b = single["grads"][s], r0["grads"][s]
        assert a.shape == b.shape
        cos = _cos(a, b)
        rel = float((a - b).abs().max() / a.abs().max())
        parity_report.record("configs[3] 2 ranks x 2 views vs 1 rank x 4 views (gloo, one GPU)", f"step {s} grad bucket",
                             cos=cos, max_err_over_scale=rel)
        assert cos > 0.9995, (s, cos)
        assert torch.equal(single["radii"][s], r0["radii"][s])
    assert abs(single["P_history"][-1] - r0["P_history"][-1]) <= 0.02 * single["P_history"][-1]
    # parameters after the Adam step before the event: Adam normalises the gradient, so an element whose gradient is
    # rounding noise (different batch shapes in the guidance: 2 views vs 4) can move by +-lr in one run and not in the
    # other -- bound the bulk tightly and the outliers by the largest learning rate (opacity, 0.05) x 2
    d = (single["flat_before_densify"] - r0["flat_before_densify"]).abs().float()
    assert float(torch.quantile(d[torch.randperm(d.numel())[:1000000]], 0.999)) < 1e-3
    assert float(d.max()) < 0.11


def test_config3_single_rank_rccl_group_runs_every_collective_on_the_gpu(tmp_path):
    """The builder gets one GPU at a time, so RCCL cannot be exercised across devices here; what CAN be checked on
    hardware is that every collective of the sharded loop -- the flat fp32 gradient bucket (sum), the int32 radii
    (max), the scalar depth maximum and its backward -- executes on the "nccl" (= RCCL) backend with the tensors and
    dtypes the loop passes, next to the hipGraph replays: a process group of ONE rank on cuda:0 must reproduce the
    group-less run (to the run-to-run noise of two processes: the GroupNorm statistics use fp64 atomics)."""
    plain = _run_workers(1, str(tmp_path))[0]
    rccl = _run_workers(1, str(tmp_path), extra_env={"GD_DIST_BACKEND": "nccl", "GD_DIST_SINGLE": "1"})[0]
    assert rccl.get("backend") == "nccl" and rccl.get("world_size") == 1 and plain.get("backend") is None
    assert len(plain["grads"]) == len(rccl["grads"]) > 0
    for s in range(len(plain["grads"])):
        cos = _cos(plain["grads"][s], rccl["grads"][s])
        parity_report.record("configs[3] one-rank RCCL group vs no group (one GPU)", f"step {s} grad bucket", cos=cos)
        assert cos > 0.9999, (s, cos)
        assert torch.equal(plain["radii"][s], rccl["radii"][s])
    assert any(rccl["densified"]) and rccl["P_history"][0] != rccl["P_history"][-1]
    assert abs(plain["P_history"][-1] - rccl["P_history"][-1]) <= 0.02 * plain["P_history"][-1]


def _vsd_objects(kw_unet, kw_vae, dtype, graphs=False, fp32_adapters=True, **gd_kw):
    from garmentdreamer_amd.guidance import sd21
    from garmentdreamer_amd.guidance.sd_vsd import LoraUnet, StableDiffusionVSD
    with torch.device(DEV):
        unet = sd21.init_random_(sd21.UNet2DConditionModel(**kw_unet))
        vae = sd21.init_random_(sd21.AutoencoderKLEncoder(**kw_vae), 1)
        lora = sd21.init_random_(sd21.LoraUNet2DConditionModel(**kw_unet), 2)
    gd = StableDiffusionVSD(DEV, fp16=dtype == torch.bfloat16, unet=unet, vae=vae, use_hip_graphs=graphs, **gd_kw)
    lora = lora.to(dtype).to(memory_format=torch.channels_last)
    if dtype == torch.bfloat16 and fp32_adapters:
        lora.adapters_to_fp32()
    train = lora.freeze_base()
    return gd, lora, train, LoraUnet(lora)


def _vsd_step(gd, q, train, seed, res=512, zero_grad=None, lora_stream=False):
    g = torch.Generator(DEV).manual_seed(seed)
    gd.set_text_embeds(torch.randn(1, 77, 1024, device=DEV, generator=g), torch.randn(1, 77, 1024, device=DEV, generator=g))
    leaf = torch.rand(1, 3, res, res, device=DEV, generator=g).requires_grad_(True)
    # a 1024^2 render is reduced to the 512^2 the NeTF guidance asserts (sd_vsd_utils.py:146), the Garment_3DGS way
    img = leaf if res == 512 else torch.nn.functional.interpolate(leaf, (512, 512), mode="bilinear", align_corners=False)
    pose = torch.randn(1, 16, device=DEV, generator=g)
    noise = torch.randn(1, 4, 64, 64, device=DEV, generator=g)
    vn = torch.randn(1, 4, 64, 64, device=DEV, generator=g)
    t = torch.tensor([317], device=DEV)
    loss, pseudo, latents = gd.train_step(img, guidance_scale=7.5, q_unet=q, pose=pose, shading="albedo", noise=noise,
                                          timesteps=t, vae_noise=vn)
    loss.backward()
    n2 = torch.randn(1, 4, 64, 64, device=DEV, generator=g)
    lu = gd.lora_train_loss(q, latents, pose, shading="albedo", unet_bs=1, timesteps=torch.tensor([611], device=DEV),
                            noise=n2, drop_pose=False)
    if zero_grad is not None:
        zero_grad()                  # an optimizer that owns the gradient buffers (flat_adam.FlatAdam)
    else:
        for p in train:
            p.grad = None
    if lora_stream:                  # the adapters' backward pass on the guidance's LoRA stream, the caller's stream not waiting
        with gd.lora_stream():
            lu.backward()
        gd.join_lora_stream()
    else:
        lu.backward()
    torch.cuda.synchronize()
    return leaf.grad.detach().float(), latents.detach().float(), float(lu), \
        {i: p.grad.detach().float().clone() for i, p in enumerate(train) if p.grad is not None}


def test_config4_vsd_iteration_full_size():
    """BASELINE configs[4] at its stated shape: the NeTF VSD iteration (trainer.py:158-262) with the full-size SD-2.1
    UNet, the full-size LoRA UNet (rank-4 adapters, camera + shading embeddings) and the VAE encoder on a 512^2 render,
    through the HIP kernels in bf16: the image gradient and every trainable gradient finite and non-zero."""
    gd, lora, train, q = _vsd_objects({}, {}, torch.bfloat16)
    assert sum(p.numel() for p in gd.unet.parameters()) == 865910724
    dimg, lat, lu, grads = _vsd_step(gd, q, train, seed=3)
    assert torch.isfinite(dimg).all() and float(dimg.abs().sum()) > 0
    assert math.isfinite(lu) and lu > 0
    assert len(grads) > 0.9 * len(train)
    assert all(torch.isfinite(v).all() for v in grads.values())
    names = [n for n, p in lora.named_parameters() if p.requires_grad]
    assert any("lora" in n for n in names) and any(n.startswith("camera_emb") for n in names)
    assert sum(float(v.abs().sum()) > 0 for v in grads.values()) > 0.5 * len(grads)


def test_config4_vsd_iteration_full_size_fp8_on_1024_render():
    """BASELINE configs[4] as stated: 1024^2 render (reduced to the 512^2 the NeTF guidance asserts), full-size SD-2.1 +
    LoRA UNets, the three no-grad UNet forwards with e4m3 MFMA convolutions (StableDiffusionVSD(fp8_unet=True)).
    Checked against the bf16 iteration on the same weights and inputs: the calibration step IS the bf16 step; on the
    next step the SDS gradient that reaches the image keeps its direction (cosine bar) and the LoRA loss its value."""
    res = {}
    for fp8 in (False, True):
        gd, lora, train, q = _vsd_objects({}, {}, torch.bfloat16, fp8_unet=fp8, fp8_calibration_steps=1)
        res[fp8] = [_vsd_step(gd, q, train, seed=sd, res=1024) for sd in (3, 4)]
        if fp8:
            assert gd.unet.fp8.mode == "run" and lora.fp8 is not None and lora.fp8.mode == "run"
            sites = gd.unet.fp8.sites_run + lora.fp8.sites_run
            # one step in "run" mode: the frozen UNet (batch 2) runs its 64^2 and 32^2 ResnetBlock convolutions in e4m3 (20),
            # the LoRA UNet (batch 1) its 64^2 ones (10); smaller maps stay bf16 (Fp8State.min_pixels: no gain there)
            assert sites >= 30, sites
        del gd, lora, train, q
        torch.cuda.empty_cache()
    (d0b, l0b, u0b, _), (d1b, l1b, u1b, g1b) = res[False]
    (d0f, l0f, u0f, _), (d1f, l1f, u1f, g1f) = res[True]
    assert d0b.shape == (1, 3, 1024, 1024)
    assert torch.equal(l0b, l0f) and _cos(d0b, d0f) > 0.9999     # calibration forward = the bf16 kernels
    c_img, c_lat = _cos(d1b, d1f), _cos(l1b, l1f)
    parity_report.record("configs[4] VSD step, full size, 1024^2 leaf: fp8 vs bf16 no-grad UNet forwards", "step",
                         cos_dL_dimage=c_img, cos_latents=c_lat, rel_dlora_loss=abs(u1b - u1f) / abs(u1b))
    assert torch.isfinite(d1f).all() and float(d1f.abs().sum()) > 0
    assert c_lat > 0.99999 and c_img > 0.99, (c_lat, c_img)
    assert abs(u1b - u1f) <= 2e-2 * abs(u1b)
    assert all(torch.isfinite(v).all() for v in g1f.values())


def test_config4_vsd_step_reduced_width_matches_eager_fp32():
    """The same iteration at reduced width: bf16 through the HIP kernels vs the fp32 PyTorch ops on identical weights
    (the reference runs fp32, sd_vsd_utils.py:35).  Latents, the image gradient and the LoRA loss agree to bf16 level."""
    kw_u = dict(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4))
    kw_v = dict(block_out_channels=(64, 64, 128, 128))
    out = {}
    for dt in (torch.float32, torch.bfloat16):
        gd, lora, train, q = _vsd_objects(kw_u, kw_v, dt)
        out[dt] = _vsd_step(gd, q, train, seed=9)
    (di32, lat32, lu32, g32), (di16, lat16, lu16, g16) = out[torch.float32], out[torch.bfloat16]
    c_lat, c_img = _cos(lat32, lat16), _cos(di32, di16)
    # per-tensor LoRA gradients are ill-conditioned at random init (the up-projections start at zero, so each
    # gradient is a sum over tokens that cancels to ~1e-3 of its terms: bf16 autograd through plain torch ops gives
    # median cosine ~0 against fp32, tools/dbg_vsd.py) -- compare all of them as ONE vector
    keys = [i for i in g32 if float(g32[i].abs().max()) > 0]
    c_lora = _cos(torch.cat([g32[i].flatten() for i in keys]), torch.cat([g16[i].flatten() for i in keys]))
    parity_report.record("configs[4] VSD step, reduced width: bf16 HIP vs fp32 eager", "step", cos_latents=c_lat,
                         cos_dL_dimage=c_img, rel_dlora_loss=abs(lu32 - lu16) / abs(lu32), cos_all_lora_grads=c_lora)
    assert c_lat > 0.999 and c_img > 0.98, (c_lat, c_img)
    assert abs(lu32 - lu16) <= 2e-2 * abs(lu32)
    assert c_lora > 0.8, c_lora


_KW_U = dict(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4))
_KW_V = dict(block_out_channels=(64, 64, 128, 128))
_TRAINED = []


def _trained_fp32_vsd():
    """The reduced-width fp32 eager VSD objects after five Adam steps (lr 1e-3) on the LoRA denoising loss: the state in which the
    adapter gradients are a signal (non-zero up-projections that fit the data), built once per test session."""
    if not _TRAINED:
        gd32, lora32, train32, q32 = _vsd_objects(_KW_U, _KW_V, torch.float32)
        opt = torch.optim.Adam(train32, lr=1e-3)
        for it in range(5):
            _vsd_step(gd32, q32, train32, seed=100 + it)        # leaves the LoRA loss's gradients in .grad
            opt.step()
        _TRAINED.append((gd32, lora32, train32, q32))
    return _TRAINED[0]


def test_config4_lora_gradients_at_a_trained_state_match_eager_fp32():
    """The adapter gradients where they can discriminate (round-4 review, weak 2): at random init every up-projection is zero,
    each adapter gradient is a token sum that cancels to ~1e-3 of its terms, and the bar above (cos > 0.8 on all of them as
    one vector) is all two runs of ANY implementation agree to.  Here the fp32 eager network first takes five Adam steps
    (lr 1e-3) on the LoRA denoising loss -- the up-projections are then non-zero and the gradient is a signal, not a
    cancellation residue --, the trained adapters / embeddings are copied into the bf16 HIP network, and the gradients of one
    more iteration on identical inputs are compared: bar cos >= 0.999 on all adapter gradients as one vector and on the median
    tensor, >= 0.99 on the worst tensor (measured 0.99997 / 0.99997 / 0.9998)."""
    kw_u, kw_v = _KW_U, _KW_V
    gd32, lora32, train32, q32 = _trained_fp32_vsd()
    name_of = {id(p): n for n, p in lora32.named_parameters()}
    names = [name_of[id(p)] for p in train32]                # index i of the gradient dicts below = train32[i]
    ups = [p for p, n in zip(train32, names) if n.endswith("up.weight")]
    assert ups and all(float(p.detach().abs().max()) > 0 for p in ups)          # the state the test is about
    gd16, lora16, train16, q16 = _vsd_objects(kw_u, kw_v, torch.bfloat16)
    with torch.no_grad():
        for p16, p32 in zip(train16, train32):
            p16.copy_(p32.to(p16.dtype))
    di32, lat32, lu32, g32 = _vsd_step(gd32, q32, train32, seed=9)
    di16, lat16, lu16, g16 = _vsd_step(gd16, q16, train16, seed=9)
    lora_idx = [i for i, n in enumerate(names) if "lora" in n and i in g32 and i in g16 and float(g32[i].abs().max()) > 0]
    assert len(lora_idx) > 0.9 * sum("lora" in n for n in names)
    c_all = _cos(torch.cat([g32[i].flatten() for i in lora_idx]), torch.cat([g16[i].flatten() for i in lora_idx]))
    per = sorted(_cos(g32[i], g16[i]) for i in lora_idx)
    c_med, c_min = per[len(per) // 2], per[0]
    parity_report.record("configs[4] VSD step, reduced width, TRAINED adapters (5 fp32 Adam steps): bf16 HIP vs fp32 eager", "step",
                         cos_all_lora_grads=c_all, cos_median_per_tensor=c_med, cos_min_per_tensor=c_min,
                         cos_latents=_cos(lat32, lat16), cos_dL_dimage=_cos(di32, di16), rel_dlora_loss=abs(lu32 - lu16) / abs(lu32))
    assert abs(lu32 - lu16) <= 2e-2 * abs(lu32)
    # measured 0.99997 / 0.99997 / 0.9998 (profiles/r05_parity_report.json); the review's bar was 0.99
    assert c_all >= 0.999, (c_all, c_med, c_min)
    assert c_med >= 0.999 and c_min >= 0.99, (c_all, c_med, c_min)


def test_config4_vsd_step_hipgraph_replay_matches_eager():
    """The graphed VSD iteration (frozen UNet, LoRA UNet no-grad forward, VAE forward/backward and the LoRA training
    forward/backward as hipGraphs) against eager launches of the same kernels on the same weights: the same numbers,
    on the capture step and on a replay with different inputs."""
    kw_u = dict(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4))
    kw_v = dict(block_out_channels=(64, 64, 128, 128))
    res = {}
    for graphs in (False, True):
        gd, lora, train, q = _vsd_objects(kw_u, kw_v, torch.bfloat16, graphs=graphs)
        res[graphs] = [_vsd_step(gd, q, train, seed=sd) for sd in (9, 10, 11)]
        if graphs:
            assert gd.use_hip_graphs, "capture fell back to eager launches"
            assert len(gd._graphs) == 4
    # the adapter gradients are sums over all tokens that cancel to ~1e-3 of their terms, and the library GEMMs'
    # stream-K reductions make the residue differ run to run: EAGER AGAINST EAGER they agree only to cosine 0.86-0.89
    # as one vector, single tensors down to -0.9 (tools/dbg_vsd_graph.py) -- so the replay is held to that level on
    # them, and to bf16 rounding on everything that is reproducible (latents, image gradient, LoRA loss).
    for (di_e, lat_e, lu_e, g_e), (di_g, lat_g, lu_g, g_g) in zip(res[False], res[True]):
        assert _cos(lat_e, lat_g) > 0.99999 and _cos(di_e, di_g) > 0.9999
        assert abs(lu_e - lu_g) <= 1e-3 * abs(lu_e)
        keys = [i for i in g_e if float(g_e[i].abs().max()) > 0]
        assert set(keys) <= set(g_g)
        assert _cos(torch.cat([g_e[i].flatten() for i in keys]), torch.cat([g_g[i].flatten() for i in keys])) > 0.8


def test_config4_frozen_and_lora_forwards_on_two_streams_give_the_one_after_the_other_result(monkeypatch):
    """Round 5: in ``train_step`` the frozen UNet (2 latents) and the LoRA UNet's no-grad forward (1 latent) replay their hipGraphs
    CONCURRENTLY on two streams (each has its own capture stream, hence its own library-GEMM workspace, and its own GroupNorm
    accumulators).  Full-size networks, six iterations on fresh inputs each way: the image gradient -- which mixes both networks'
    outputs -- and the latents agree with the one-after-the-other run to the level two runs of the SAME schedule agree (the
    library's stream-K GEMMs are not bitwise reproducible), every time."""
    from garmentdreamer_amd.guidance import sd_vsd
    out = {}
    for conc in (False, True):
        monkeypatch.setattr(sd_vsd, "_CONCURRENT", conc)
        gd, lora, train, q = _vsd_objects({}, {}, torch.bfloat16, graphs=True)
        out[conc] = [_vsd_step(gd, q, train, seed=40 + i) for i in range(6)]
        assert gd.use_hip_graphs, "capture fell back to eager launches"
        del gd, lora, train, q
        torch.cuda.empty_cache()
    worst = 1.0
    for (di_s, lat_s, lu_s, _), (di_c, lat_c, lu_c, _) in zip(out[False], out[True]):
        assert torch.isfinite(di_c).all() and float(di_c.abs().sum()) > 0
        assert _cos(lat_s, lat_c) > 0.99999
        worst = min(worst, _cos(di_s, di_c))
        assert abs(lu_s - lu_c) <= 2e-3 * abs(lu_s)
    parity_report.record("configs[4] VSD step, full size: frozen + LoRA no-grad forwards on two streams vs one after the other", "step",
                         min_cos_dL_dimage=worst)
    assert worst > 0.9999, worst


def test_config4_lora_backward_on_the_lora_stream_gives_the_callers_stream_gradients():
    """``StableDiffusionVSD.lora_stream()`` (round 5): the adapters' backward pass runs on the LoRA side stream without the caller's
    stream waiting for it.  At the trained state of the test above (adapter gradients are a signal there; with untrained
    up-projections two runs of ANY schedule give uncorrelated adapter gradients, tools/lora_grad_repro.py; profiles/r05_lora_grad_repro.txt) the gradients of
    four iterations on fresh inputs -- zeroed on the CALLER's stream right before each backward pass, which the side stream
    therefore has to wait for -- agree with the caller's-stream schedule as well as that schedule reproduces itself."""
    _, _, train32, _ = _trained_fp32_vsd()
    runs = {}
    for name, ls in (("plain", False), ("plain again", False), ("lora stream", True)):
        gd, lora, train, q = _vsd_objects(_KW_U, _KW_V, torch.bfloat16, graphs=True)
        with torch.no_grad():
            for p16, p32 in zip(train, train32):
                p16.copy_(p32.to(p16.dtype))
        runs[name] = [_vsd_step(gd, q, train, seed=60 + i, lora_stream=ls) for i in range(4)]
        assert gd.use_hip_graphs
        names = {id(p): n for n, p in lora.named_parameters()}
        idx = [i for i, p in enumerate(train) if "lora" in names[id(p)]]
        del gd, lora, train, q
        torch.cuda.empty_cache()
    cat = lambda gr: torch.cat([gr[i].flatten() for i in idx])
    worst_self, worst_ls = 1.0, 1.0
    for it in range(4):
        ref, again, ls = (runs[k][it] for k in ("plain", "plain again", "lora stream"))
        assert set(idx) <= set(ls[3]) and all(torch.isfinite(ls[3][i]).all() for i in idx)
        c_self, c_ls = _cos(cat(ref[3]), cat(again[3])), _cos(cat(ref[3]), cat(ls[3]))
        worst_self, worst_ls = min(worst_self, c_self), min(worst_ls, c_ls)
        assert _cos(ref[0], ls[0]) > 0.9999 and abs(ref[2] - ls[2]) <= 1e-2 * abs(ref[2])
    parity_report.record("configs[4] VSD step, reduced width, trained adapters: LoRA backward on the LoRA stream vs the caller's stream",
                         "step", min_cos_all_lora_grads=worst_ls, min_cos_same_schedule_twice=worst_self)
    assert worst_self > 0.99, worst_self                     # the state the comparison needs: reproducible gradients
    assert worst_ls > 0.99 and worst_ls > worst_self - 5e-3, (worst_self, worst_ls)


def test_config4_three_stream_schedule_against_one_stream_at_bit_level_where_the_step_is_reproducible(monkeypatch):
    """The round-5 review's missing bar: cosines would let a race that perturbs one adapter by 1e-3 through.  At the trained
    state, with FIXED noise and timesteps, the one-stream schedule is run twice and the three-stream schedule (frozen | LoRA
    no-grad | training pass on their own streams, backward on the LoRA stream) once, on fresh objects each.  Whatever the
    one-stream schedule reproduces bit for bit between its two runs (latents, image gradient, LoRA loss, every adapter
    gradient -- the library's stream-K GEMMs decide that, tools/lora_grad_repro.py), the three-stream schedule must reproduce bit
    for bit too; what it does not reproduce exactly must agree across schedules to within four times the one-stream run-to-run
    difference at the worst element."""
    from garmentdreamer_amd.guidance import sd_vsd
    _, _, train32, _ = _trained_fp32_vsd()
    runs = {}
    for name, conc, ls in (("one", False, False), ("one again", False, False), ("three", True, True)):
        monkeypatch.setattr(sd_vsd, "_CONCURRENT", conc)
        gd, lora, train, q = _vsd_objects(_KW_U, _KW_V, torch.bfloat16, graphs=True)
        with torch.no_grad():
            for p16, p32 in zip(train, train32):
                p16.copy_(p32.to(p16.dtype))
        runs[name] = [_vsd_step(gd, q, train, seed=80 + i, lora_stream=ls) for i in range(3)]      # step 0 captures, 1-2 replay
        assert gd.use_hip_graphs
        del gd, lora, train, q
        torch.cuda.empty_cache()
    exact, total, worst_ratio = 0, 0, 0.0
    for it in range(3):
        a, b, c = runs["one"][it], runs["one again"][it], runs["three"][it]
        items = [("dL/dimage", a[0], b[0], c[0]), ("latents", a[1], b[1], c[1]),
                 ("lora loss", torch.tensor(a[2]), torch.tensor(b[2]), torch.tensor(c[2]))]
        items += [(f"adapter grad {i}", a[3][i], b[3][i], c[3][i]) for i in sorted(a[3]) if i in b[3] and i in c[3]]
        for name, x1, x2, y in items:
            total += 1
            d_self = float((x1 - x2).abs().max())
            d_cross = float((x1 - y).abs().max())
            if d_self == 0.0:
                exact += 1
                assert d_cross == 0.0, f"iteration {it}, {name}: reproducible on one stream, differs on three ({d_cross:.3e})"
            else:
                scale = float(x1.abs().max()) + 1e-30
                worst_ratio = max(worst_ratio, d_cross / d_self)
                assert d_cross <= 4.0 * d_self + 1e-5 * scale, f"iteration {it}, {name}: {d_cross:.3e} vs run-to-run {d_self:.3e}"
    parity_report.record("configs[4] VSD step, reduced width, trained adapters: three-stream vs one-stream schedule, element level",
                         "step", quantities=total, bitwise_reproducible_on_one_stream=exact, worst_cross_over_self=worst_ratio)
    assert exact > 0, "nothing of the step is bitwise reproducible: the bit-level half of this test did not run"


def test_config4_two_lora_training_passes_after_one_train_step_see_the_optimizer_step_in_between(monkeypatch):
    """trainer.py's K loop: ``lora_train_loss`` -> backward -> optimizer step TWICE after one ``train_step``, the optimizer on the
    caller's stream (the reference's unedited sequence).  The training pass runs on its own stream behind an event recorded after
    the latents -- good for one call only: the second pass has to wait for the optimizer step the caller ran in between.  With a
    large learning rate the second loss depends on that step: the three-stream schedule reproduces the one-stream losses."""
    from garmentdreamer_amd.guidance import sd_vsd
    _, _, train32, _ = _trained_fp32_vsd()
    out = {}
    for conc in (False, True):
        monkeypatch.setattr(sd_vsd, "_CONCURRENT", conc)
        gd, lora, train, q = _vsd_objects(_KW_U, _KW_V, torch.bfloat16, graphs=True)
        with torch.no_grad():
            for p16, p32 in zip(train, train32):
                p16.copy_(p32.to(p16.dtype))
        opt = torch.optim.Adam(train, lr=3e-2)
        g = torch.Generator(DEV).manual_seed(5)
        gd.set_text_embeds(torch.randn(1, 77, 1024, device=DEV, generator=g), torch.randn(1, 77, 1024, device=DEV, generator=g))
        leaf = torch.rand(1, 3, 512, 512, device=DEV, generator=g).requires_grad_(True)
        pose = torch.randn(1, 16, device=DEV, generator=g)
        losses = []
        for it in range(3):                          # iteration 0 captures the graphs
            torch.manual_seed(1000 + it)
            loss, _, latents = gd.train_step(leaf, guidance_scale=7.5, q_unet=q, pose=pose, shading="albedo")
            loss.backward()
            for k in range(2):
                lu = gd.lora_train_loss(q, latents, pose, shading="albedo", unet_bs=1, drop_pose=False)
                opt.zero_grad(set_to_none=True)
                lu.backward()
                opt.step()
                losses.append(float(lu))
        torch.cuda.synchronize()
        assert gd.use_hip_graphs and all(math.isfinite(v) for v in losses)
        out[conc] = losses
        del gd, lora, train, q, opt
        torch.cuda.empty_cache()
    a, b = out[False], out[True]
    assert max(abs(x - y) / abs(x) for x, y in zip(a, b)) <= 3e-2, (a, b)
    # the state the test needs: at this learning rate the losses move by far more than that from pass to pass (1.25 ... 0.2 measured),
    # so a pass that read the adapters before (or while) the optimizer step in front of it wrote them would not land within 3 %
    assert max(a) / min(a) > 3.0, a


def test_config4_vsd_graphed_iteration_with_flat_adam_gradient_sinks():
    """The graphed VSD iteration with flat_adam.FlatAdam owning the adapters' gradients (the LoRA backward kernels inside the
    hipGraph add into slices of its flat buffer) against eager launches with plain autograd gradients, same weights (lr = 0).
    Everything reproducible agrees to bf16 rounding.  The adapter gradients of two INSTANCES are uncorrelated whatever the path
    (tools/vsd_sink_probe.py: cosine 0.1 between two plain eager instances alive at once -- sums that cancel to ~1e-3 of their
    terms, decided by bf16-level differences upstream), so they are held to what is reproducible: the same size as the eager
    ones (the warm-up passes torch runs before a capture must not stay in the accumulating buffer: that would be 5x), and the
    capture step's gradients reproduced by a later replay on the same inputs."""
    from garmentdreamer_amd.flat_adam import FlatAdam
    kw_u = dict(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4))
    kw_v = dict(block_out_channels=(64, 64, 128, 128))
    seeds = (9, 10, 9)
    gd_e, _, train_e, q_e = _vsd_objects(kw_u, kw_v, torch.bfloat16, graphs=False)
    eager = [_vsd_step(gd_e, q_e, train_e, seed=sd) for sd in seeds]
    gd_g, _, train_g, q_g = _vsd_objects(kw_u, kw_v, torch.bfloat16, graphs=True)
    opt = FlatAdam.for_lora_unet(q_g.unet, train_g, lr=0.0)              # before any capture: it re-seats the adapters
    flat = [p for p in train_g if hasattr(p, "_gd_grad_sink")]
    assert len(flat) >= 32 and all(p.grad is p._gd_grad_sink for p in flat)
    got = []
    for (di_e, lat_e, lu_e, g_e), sd in zip(eager, seeds):
        di_g, lat_g, lu_g, g_g = _vsd_step(gd_g, q_g, train_g, seed=sd, zero_grad=opt.zero_grad)
        assert gd_g.use_hip_graphs, "capture fell back to eager launches"
        assert _cos(lat_e, lat_g) > 0.99999 and _cos(di_e, di_g) > 0.9999
        assert abs(lu_e - lu_g) <= 1e-3 * abs(lu_e)
        keys = [i for i in g_e if float(g_e[i].abs().max()) > 0 and hasattr(train_g[i], "_gd_grad_sink")]
        assert len(keys) >= 32
        a = torch.cat([g_e[i].flatten() for i in keys])
        b = torch.cat([g_g[i].flatten() for i in keys])
        assert torch.isfinite(b).all()
        ratio = float(b.norm() / a.norm())
        assert 0.7 < ratio < 1.4, (sd, ratio)
        got.append(b)
        opt.step()
        assert all(p.grad is p._gd_grad_sink for p in flat)
    # capture step (warm-up passes undone, first replay) vs a later replay on the same inputs: the library GEMMs' reductions
    # may differ run to run (see the test above), the accumulation must not
    c = _cos(got[0], got[2])
    parity_report.record("configs[4] VSD step, reduced width: graphed + FlatAdam gradient sinks", "capture step vs replay",
                         cos_adapter_grads=c, norm_ratio=float(got[2].norm() / got[0].norm()))
    assert c > 0.8 and 0.9 < float(got[2].norm() / got[0].norm()) < 1.1, c


def test_config3_full_shape_two_ranks_x_4_views_vs_one_rank_x_8_views(tmp_path):
    """configs[3] at its stated shape through the sharded loop: 8 views x 100 000 Gaussians @512^2 with the FULL-SIZE
    bf16 SD-2.1 nets (HIP kernels, hipGraphs), 2 ranks x 4 views (gloo over the one GPU this box has) against 1 rank x
    8 views.  Replicas end bit-identical; the all-reduced gradient bucket matches the single-rank one to the level
    the bf16 guidance allows when its batch is split (8 vs 4 latents per UNet call pick different tiles)."""
    env = {"GD_TEST_CFG": "full"}
    single = _run_workers(1, str(tmp_path), env)[0]
    r0, r1 = _run_workers(2, str(tmp_path), env)
    for k in ("flat", "exp_avg", "exp_avg_sq", "max_radii2D", "xyz_gradient_accum", "denom"):
        assert torch.equal(r0[k], r1[k]), k
    assert single["P"] == r0["P"] == 100000
    for s in range(len(single["grads"])):
        a, b = single["grads"][s], r0["grads"][s]
        cos = _cos(a, b)
        parity_report.record("configs[3] full shape: 2 ranks x 4 views vs 1 rank x 8 views (gloo, one GPU, bf16 nets)",
                             f"step {s} grad bucket", cos=cos, max_err_over_scale=float((a - b).abs().max() / a.abs().max()))
        assert cos > 0.995, (s, cos)
        if s == 0:      # same parameters on both sides: the max over views of the integer radii is exact
            assert torch.equal(single["radii"][s], r0["radii"][s])
        else:           # after an Adam step on slightly different gradients a radius may round the other way
            diff = (single["radii"][s] - r0["radii"][s]).abs()
            assert float((diff > 0).float().mean()) < 5e-3 and float(diff.max()) <= 1
    assert torch.isfinite(r0["flat"]).all()
    # SDSLoop(batch_invariant=True): each rank selects its guidance kernels for the whole 8-view batch and hands the
    # library GEMMs the single-rank row set (its own rows at their global positions), so every view's SDS gradient
    # carries the bits of the single-rank run (tools/guidance_invariance.py: dL/drgb max|d| 0)
    (tmp_path / "bi").mkdir()
    b0, b1 = _run_workers(2, str(tmp_path / "bi"), dict(env, GD_TEST_BATCH_INVARIANT="1"))
    assert torch.equal(b0["flat"], b1["flat"])
    for s in range(len(single["grads"])):
        a, b = single["grads"][s], b0["grads"][s]
        cos_bi = _cos(a, b)
        parity_report.record("configs[3] full shape: 2 ranks x 4 views vs 1 rank x 8 views (gloo, one GPU, bf16 nets)",
                             f"step {s} grad bucket, batch-invariant kernel selection", cos=cos_bi,
                             max_err_over_scale=float((a - b).abs().max() / a.abs().max()))
        # measured: step 0 (same parameters on both sides) cos 1 - 2e-13, max error 9e-8 of the largest entry -- the fp32
        # all-reduce sums 2 x 4 views in another order than the single rank's 8.  Step 1 follows an Adam step (eps 1e-15) on
        # those gradients: an entry whose sign the 9e-8 decides moves its parameter by +-lr, and guidance_scale = 100
        # amplifies that like any other perturbation -- usually 1 - 3e-10, but 0.99997 and 0.9987 were seen in 2 of 9
        # repeated runs of the suite, so only step 0 carries the bit-level bar
        assert cos_bi > (0.999999 if s == 0 else 0.995), (s, cos_bi)
