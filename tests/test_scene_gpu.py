"""GPU parity of the scene-side HIP kernels (include/gd_scene.h): distCUDA2 bit-exact against the CPU oracle,
the flat multi-group Adam against torch.optim.Adam, the densification statistics against the torch ops of the
loop, and the SDS loop running on GaussianModel's flat buffers."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("P,seed", [(1, 0), (3, 1), (4, 2), (700, 3), (1024, 4), (1025, 5), (20000, 6), (100000, 7)])
def test_dist2_bit_exact_vs_oracle(P, seed):
    from garmentdreamer_amd.gaussian_model import distCUDA2
    from oracle import gd_oracle
    rng = np.random.default_rng(seed)
    pts = (rng.normal(size=(P, 3)) * np.array([1.0, 0.4, 2.5]) + np.array([0.3, -0.2, 1.5])).astype(np.float32)
    if P >= 16:
        pts[: P // 8] = pts[P // 8: 2 * (P // 8)]       # duplicates: zero distances, equal Morton codes
    ref = gd_oracle.dist2(pts)
    out = distCUDA2(torch.from_numpy(pts).to(DEV)).cpu().numpy()
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), np.abs(out - ref).max()


def test_dist2_clustered_far_from_origin_and_rejects_cpu():
    from garmentdreamer_amd.gaussian_model import distCUDA2
    from oracle import gd_oracle
    rng = np.random.default_rng(11)
    centres = rng.uniform(5.0, 9.0, size=(20, 3))
    pts = (centres[rng.integers(0, 20, size=30000)] + rng.normal(scale=0.01, size=(30000, 3))).astype(np.float32)
    ref = gd_oracle.dist2(pts)
    out = distCUDA2(torch.from_numpy(pts).to(DEV)).cpu().numpy()
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    with pytest.raises(RuntimeError, match="no CPU path"):
        distCUDA2(torch.from_numpy(pts))


def _model(P, deg=0, seed=0):
    from garmentdreamer_amd import gaussian_model as gm
    g = torch.Generator().manual_seed(seed)
    m = gm.GaussianModel(sh_degree=deg, device=DEV)
    M = (deg + 1) ** 2
    m._pack({"xyz": torch.randn(P, 3, generator=g), "f_dc": torch.randn(P, 1, 3, generator=g),
             "f_rest": torch.randn(P, M - 1, 3, generator=g), "opacity": torch.randn(P, 1, generator=g),
             "scaling": torch.randn(P, 3, generator=g) * 0.3 - 3.0, "rotation": torch.randn(P, 4, generator=g)})
    m.spatial_lr_scale = 1.0
    m.training_setup()
    return m


def test_flat_adam_matches_torch_adam_over_several_steps():
    """One launch over the flat buffer with per-group learning rates == torch.optim.Adam(l, lr=0.0, eps=1e-15)
    over the six groups (gaussian_model.py:156-167), including a learning-rate change between steps."""
    from garmentdreamer_amd.gaussian_model import GROUPS
    P = 3000
    m = _model(P, deg=1, seed=2)
    ref_params = {n: t.detach().clone().requires_grad_(True) for n, t in m._current()[0].items()}
    opt = torch.optim.Adam([{"params": [ref_params[n]], "lr": m.lrs[n], "name": n} for n in GROUPS], lr=0.0, eps=1e-15)
    g = torch.Generator(DEV).manual_seed(5)
    for it in range(6):
        m.zero_grad()
        grads = {n: torch.randn(ref_params[n].shape, device=DEV, generator=g) * (10.0 ** (it - 3)) for n in GROUPS}
        for n, p in (("xyz", m._xyz), ("f_dc", m._features_dc), ("f_rest", m._features_rest), ("opacity", m._opacity),
                     ("scaling", m._scaling), ("rotation", m._rotation)):
            p.grad.copy_(grads[n])
            ref_params[n].grad = grads[n].clone()
        if it == 3:
            lr = m.update_learning_rate(1234)
            for grp in opt.param_groups:
                if grp["name"] == "xyz":
                    grp["lr"] = lr
        m.step()
        opt.step()
        cur = m._current()[0]
        for n in GROUPS:
            a, b = cur[n], ref_params[n].detach()
            # one fp32 ulp of the parameter or of the update (|lr * m / denom| <= lr), op order differs from torch's
            assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), (it, n, (a - b).abs().max().item())
    st = opt.state[ref_params["scaling"]]
    ea, ev = st["exp_avg"], st["exp_avg_sq"]
    assert torch.allclose(m._group_views(m._exp_avg)["scaling"].view_as(ea), ea, rtol=1e-5, atol=1e-6 * ea.abs().max().item())
    assert torch.allclose(m._group_views(m._exp_avg_sq)["scaling"].view_as(ev), ev, rtol=1e-5,
                          atol=1e-6 * ev.abs().max().item())


def test_flat_adam_over_many_small_tensors_matches_torch_adam():
    """garmentdreamer_amd.flat_adam.FlatAdam (the LoRA UNet's optimizer, trainer.py:137: torch.optim.Adam over ~260 small
    tensors): the fp32 parameters and their .grad are views of two flat buffers, a step is one gd_scene_adam_step launch,
    zero_grad one memset; gradients accumulate in place like torch's; bf16 / excluded parameters go through torch.optim.Adam."""
    from garmentdreamer_amd.flat_adam import FlatAdam
    g = torch.Generator(DEV).manual_seed(11)
    shapes = [(4, 320), (320, 4), (4, 1024), (640, 4), (7,), (3, 5, 2), (1280, 4)]
    mine = [torch.nn.Parameter(torch.randn(*sh, device=DEV, generator=g)) for sh in shapes]
    mine.append(torch.nn.Parameter(torch.randn(16, 8, device=DEV, generator=g).to(torch.bfloat16)))      # -> torch.optim.Adam
    sparse = torch.nn.Parameter(torch.randn(9, device=DEV, generator=g))                                    # excluded, never a gradient
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    unused = torch.nn.Parameter(torch.randn(12, device=DEV, generator=g))      # fp32, NOT named in `flat`: torch's Adam, skipped while .grad is None
    # fp32, not in the flat set, a gradient on SOME steps only (the shading embedding of the step): torch's skip semantics --
    # no moment decay and no step count on the steps without one -- through the HIP kernel, one launch per tensor with a gradient
    solo = torch.nn.Parameter(torch.randn(33, 5, device=DEV, generator=g))
    solo_ref = torch.nn.Parameter(solo.detach().clone())
    opt = FlatAdam(mine + [sparse, unused, solo], lr=3e-3, flat=mine + [sparse], exclude=[sparse])
    ropt = torch.optim.Adam(ref + [solo_ref], lr=3e-3)
    assert opt._rest is not None and len(opt._solo) == 3 and not hasattr(solo, "_gd_grad_sink")
    assert all(p.data_ptr() % 256 == 0 and p.grad is p._gd_grad_sink for p in mine[:-1])    # re-seated, aligned views of one buffer
    assert mine[-1].grad is None and not hasattr(sparse, "_gd_grad_sink") and not hasattr(unused, "_gd_grad_sink")
    unused0 = unused.detach().clone()
    assert opt.param_groups is opt.param_groups        # ONE list of ONE dict: editing it is how the learning rate changes
    assert torch.equal(torch.cat([p.detach().flatten().float() for p in mine]), torch.cat([p.detach().flatten().float() for p in ref]))
    sparse0 = sparse.detach().clone()
    for it in range(6):
        opt.zero_grad()
        ropt.zero_grad()
        assert float(opt.flat_grad.abs().sum()) == 0.0
        for p, q in zip(mine, ref):
            gr = (torch.randn(p.shape, device=DEV, generator=g) * 10.0 ** (it - 3)).to(p.dtype)
            q.grad = gr.clone()
            if p.dtype != torch.float32 or it == 2:
                p.grad = gr.clone()              # a foreign tensor in .grad: step() folds it into the view and re-seats it
            elif it % 2:
                p.grad.add_(gr)                  # what autograd's AccumulateGrad does with an existing .grad
            else:
                p._gd_grad_sink.add_(gr / 2)     # what the LoRA backward kernels do, twice (accumulation)
                p._gd_grad_sink.add_(gr / 2)
        if it in (0, 2, 3, 5):
            gr = torch.randn(solo.shape, device=DEV, generator=g)
            solo.grad, solo_ref.grad = gr.clone(), gr.clone()
        else:
            assert solo.grad is None and solo_ref.grad is None
        if it == 4:
            for grp in opt.param_groups:         # the torch idiom (it used to edit a throw-away dict)
                grp["lr"] = 1e-3
            assert opt.lr == 1e-3
            for grp in ropt.param_groups:
                grp["lr"] = 1e-3
        opt.step()
        ropt.step()
        for p, q in zip(mine, ref):
            tol = dict(rtol=2e-6, atol=2e-7) if p.dtype == torch.float32 else dict(rtol=0, atol=0)
            assert torch.allclose(p.detach().float(), q.detach().float(), **tol), (it, p.shape)
        assert all(p.grad is p._gd_grad_sink for p in mine[:-1])
        assert torch.allclose(solo.detach(), solo_ref.detach(), rtol=2e-6, atol=2e-7), it
    assert opt._solo_state[id(solo)][0] == 4
    assert torch.equal(sparse.detach(), sparse0) and torch.equal(unused.detach(), unused0)
    assert opt.flat_grad.numel() == sum((p.numel() + 63) // 64 * 64 for p in mine[:-1])


def test_densify_stats_kernel_matches_torch_ops():
    P = 5000
    m = _model(P)
    g = torch.Generator(DEV).manual_seed(1)
    mr, acc, den = m.max_radii2D.clone(), m.xyz_gradient_accum.clone(), m.denom.clone()
    for _ in range(3):
        radii = torch.randint(-2, 40, (P,), device=DEV, generator=g, dtype=torch.int32).clamp_min(0)
        vg = torch.randn(P, 3, device=DEV, generator=g)
        m.add_densification_stats(vg, radii)
        vis = radii > 0
        mr = torch.where(vis, torch.max(mr, radii.float()), mr)
        acc = acc + torch.where(vis[:, None], vg[:, :2].norm(dim=-1, keepdim=True), torch.zeros_like(acc))
        den = den + vis[:, None].float()
    assert torch.equal(m.max_radii2D, mr) and torch.equal(m.denom, den)
    assert torch.allclose(m.xyz_gradient_accum, acc, rtol=1e-6, atol=1e-7)


def test_create_from_pcd_and_loop_on_flat_buffers():
    """create_from_pcd (HIP distCUDA2 -> initial scales), then SDS-loop iterations with the flat-buffer model:
    gradients land in the flat buffer, one HIP Adam launch moves every group, densification keeps running."""
    from garmentdreamer_amd import cameras as gcam, gaussian_model as gm
    from garmentdreamer_amd.sds_loop import SDSLoop
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(4000, 3)).astype(np.float32) * 0.4
    cols = rng.uniform(size=(4000, 3)).astype(np.float32)
    m = gm.GaussianModel(sh_degree=0, device=DEV)
    m.create_from_pcd(pts, cols, spatial_lr_scale=5.0)
    d2 = np.maximum(__import__("oracle.gd_oracle", fromlist=["x"]).dist2(pts), 1e-7)
    assert torch.allclose(m._scaling.data[:, 0].cpu(), torch.from_numpy(np.log(np.sqrt(d2))), rtol=1e-6, atol=1e-6)
    assert torch.allclose(m.get_opacity, torch.full_like(m.get_opacity, 0.1), atol=1e-6)
    m.training_setup()

    class ToyGuidance:   # deterministic stand-in: pulls the render towards grey
        def __call__(self, rgb, *a, **k):
            return {"loss_sds": ((rgb - 0.5) ** 2).sum() / rgb.shape[0], "grad_norm": torch.zeros((), device=rgb.device)}

        def set_min_max_steps(self, **k):
            pass

    loop = SDSLoop(m, ToyGuidance(), None, torch.ones(3, device=DEV))
    assert loop.native_scene
    before = m._flat.clone()
    for step in range(3):
        batch = gcam.orbit_batch(2, elevation_deg=15.0, camera_distance=2.75, fovy_deg=55.0, height=64, width=64,
                                 azimuth_offset_deg=10.0 * step)
        out = loop.step(batch)
    assert torch.isfinite(out["loss"]) and torch.isfinite(m._flat).all()
    assert (m._flat != before).float().mean() > 0.3 and m.flat_grad.abs().sum() > 0
    assert m.denom.sum() > 0 and m.max_radii2D.max() > 0
    P0 = m._xyz.shape[0]
    m.densify_and_prune(max_grad=1e-9, min_opacity=0.005, extent=5.0, max_screen_size=None)
    assert m._xyz.shape[0] > P0
    loop.step(gcam.orbit_batch(2, elevation_deg=15.0, camera_distance=2.75, fovy_deg=55.0, height=64, width=64))
    assert torch.isfinite(m._flat).all()


def test_loop_without_the_forward_host_sync_moves_the_scene_exactly_like_the_synchronising_loop():
    """SDSLoop(sync_free=True) (the default: gd_raster_forward_batched_capacity from the second iteration on, the capacity
    re-sized from every iteration's deferred count) against SDSLoop(sync_free=False) on the same scene and cameras: the flat
    parameter buffers agree bit for bit after six iterations, and the sync-free loop really ran sync-free."""
    from garmentdreamer_amd import cameras as gcam
    from garmentdreamer_amd.gaussian_model import GaussianModel
    from garmentdreamer_amd.scene import synthetic_gaussians
    from garmentdreamer_amd.sds_loop import SDSLoop

    class ToyGuidance:
        def __call__(self, rgb, *a, **k):
            return {"loss_sds": ((rgb - 0.5) ** 2).sum() / rgb.shape[0], "grad_norm": torch.zeros((), device=rgb.device)}

        def set_min_max_steps(self, **k):
            pass

    flats, loops = [], []
    for sync_free in (False, True):
        m = GaussianModel.from_activated(synthetic_gaussians(6000, seed=4), device=DEV)
        loop = SDSLoop(m, ToyGuidance(), None, torch.ones(3, device=DEV), sync_free=sync_free, densify=False)
        for step in range(6):
            loop.step(gcam.orbit_batch(3, elevation_deg=15.0, camera_distance=2.75, fovy_deg=55.0, height=96, width=96,
                                       azimuth_offset_deg=17.0 * step))
        torch.cuda.synchronize()
        flats.append(m._flat.clone())
        loops.append(loop)
    assert loops[0].capacity is None and loops[1].capacity.calls_sync_free == 5
    assert torch.isfinite(flats[0]).all() and torch.equal(flats[0], flats[1])
    loops[1].capacity.collect()
    assert loops[1].capacity.last_count > 0


def test_sync_free_loop_survives_reference_range_cameras_an_overflow_never_reaches_the_optimizer():
    """The reference draws camera_distance in [1.5, 4.0] and fovy in [40, 70] PER VIEW (configs/gaussiandreamer-sd.yaml:11-12):
    the instance count moves several-fold from one iteration to the next, so a capacity seeded from one iteration overflows.
    SDSLoop.step looks at the deferred overflow flag after the guidance forward and repeats the render on the synchronising
    path; backward, densification statistics and Adam only ever see a complete render: after ten iterations on random
    reference-range batches (a far batch first, so the seed is small) the parameters equal the synchronising loop's bit for
    bit, and the overflow path was really taken."""
    from garmentdreamer_amd import cameras as gcam
    from garmentdreamer_amd.gaussian_model import GaussianModel
    from garmentdreamer_amd.scene import synthetic_gaussians
    from garmentdreamer_amd.sds_loop import SDSLoop

    class ToyGuidance:
        def __call__(self, rgb, *a, **k):
            return {"loss_sds": ((rgb - 0.5) ** 2).sum() / rgb.shape[0], "grad_norm": torch.zeros((), device=rgb.device)}

        def set_min_max_steps(self, **k):
            pass

    flats, loops = [], []
    for sync_free in (False, True):
        m = GaussianModel.from_activated(synthetic_gaussians(6000, seed=4), device=DEV)
        loop = SDSLoop(m, ToyGuidance(), None, torch.ones(3, device=DEV), sync_free=sync_free, densify=False,
                       capacity_margin=1.1, capacity_quantum=256)
        g = torch.Generator().manual_seed(11)
        # oracle counts of this scene at 256^2: 14.6k instances per far view, 19.3k per near view (1.32x > the 1.1 margin)
        far = dict(camera_distance_range=(3.9, 4.0), fovy_range_deg=(40.0, 41.0))
        near = dict(camera_distance_range=(1.5, 1.55), fovy_range_deg=(69.0, 70.0))
        for step in range(10):
            kw = far if step in (0, 5) else near if step in (1, 6) else {}
            loop.step(gcam.random_batch(3, generator=g, height=256, width=256, **kw))
        torch.cuda.synchronize()
        flats.append(m._flat.clone())
        loops.append(loop)
    assert loops[1].render_overflow_retries >= 1 and loops[1].capacity.overflows == loops[1].render_overflow_retries
    assert loops[1].capacity.calls_sync_free >= 3
    assert torch.isfinite(flats[0]).all() and torch.equal(flats[0], flats[1])
    loops[1].capacity.collect()          # nothing pending: the loop looked at every count itself


def test_simple_knn_import_path_is_served_by_the_hip_kernel():
    from simple_knn._C import distCUDA2
    pts = torch.randn(500, 3, device=DEV)
    d = distCUDA2(pts)
    D = torch.cdist(pts.double(), pts.double()) ** 2
    D.fill_diagonal_(float("inf"))
    ref = D.topk(3, largest=False).values.mean(dim=1)
    assert d.shape == (500,) and torch.allclose(d.double(), ref, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("sh_degree", [0, 2])
def test_fused_activations_match_torch_accessors_and_their_gradients(sh_degree):
    """GaussianModel.activated() (gd_scene_activate_forward / _backward, one launch each, gradients accumulated
    straight into the flat gradient buffer) vs get_features / get_opacity / get_scaling / get_rotation through
    torch autograd (scene/gaussian_model.py:95-115)."""
    from garmentdreamer_amd.gaussian_model import GaussianModel
    from garmentdreamer_amd.scene import synthetic_gaussians
    gm = GaussianModel.from_activated(synthetic_gaussians(5000, seed=3, sh_degree=sh_degree), sh_degree=sh_degree, device=DEV)
    with torch.no_grad():
        gm._rotation[:7] *= 3.7                  # un-normalised quaternions exercise the normalize Jacobian
        gm._rotation[7] = 0.0                    # and the clamped denominator
    outs = gm.activated()
    refs = (gm.get_features, gm.get_opacity, gm.get_scaling, gm.get_rotation)
    for o, r in zip(outs, refs):
        assert o.shape == r.shape
        torch.testing.assert_close(o, r, rtol=2e-6, atol=1e-7)
    g = torch.Generator(DEV).manual_seed(1)
    ws = [torch.randn(o.shape, device=DEV, generator=g) for o in outs]
    gm.zero_grad()
    sum((o * w).sum() for o, w in zip(outs, ws)).backward()
    fused = gm.flat_grad.clone()
    assert fused.abs().sum().item() > 0
    gm.zero_grad()
    sum((r * w).sum() for r, w in zip(refs, ws)).backward()
    ref = gm.flat_grad.clone()
    ok = torch.isfinite(ref)
    torch.testing.assert_close(fused[ok], ref[ok], rtol=1e-5, atol=1e-6)
    # twice through the same buffer accumulates, like AccumulateGrad on an existing .grad
    gm.zero_grad()
    for _ in range(2):
        o2 = gm.activated()
        sum((o * w).sum() for o, w in zip(o2, ws)).backward()
    torch.testing.assert_close(gm.flat_grad[ok], 2 * fused[ok], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("P,sh_degree,screen", [(5000, 0, None), (30000, 0, 20), (7777, 2, 20), (257, 0, None)])
def test_native_densify_and_prune_matches_the_torch_sequence(P, sh_degree, screen):
    """gd_scene_densify_plan / _apply vs the reference's op-for-op sequence (``densify_and_prune_torch``: clone, split,
    prune with torch indexing, scene/gaussian_model.py:283-413) on the same scene, statistics and generator seed: same
    point count and order, parameters and Adam moments bit-identical except the children's positions (R(q) . sample
    is a bmm in the reference: summation order unspecified -> 1e-6)."""
    from garmentdreamer_amd import gaussian_model as gm
    from garmentdreamer_amd.scene import synthetic_gaussians
    models = []
    for _ in range(2):
        sc = synthetic_gaussians(P, seed=3, sh_degree=sh_degree)
        rng = np.random.default_rng(5)
        sc["scales"] = (sc["scales"] * rng.uniform(0.5, 6.0, size=(P, 1))).astype(np.float32)   # both sides of percent_dense * extent
        sc["opacities"] = rng.uniform(0.001, 0.9, size=(P, 1)).astype(np.float32)              # some below min_opacity
        m = gm.GaussianModel.from_activated(sc, sh_degree=sh_degree, device=DEV)
        g = torch.Generator(device="cpu").manual_seed(11)
        m._exp_avg.copy_(torch.randn(m._exp_avg.shape, generator=g))
        m._exp_avg_sq.copy_(torch.rand(m._exp_avg_sq.shape, generator=g))
        acc = torch.rand((P, 1), generator=g) * 6e-4
        den = torch.randint(0, 4, (P, 1), generator=g).float()       # zeros -> NaN / inf gradients
        m.xyz_gradient_accum.copy_(acc * den)
        m.denom.copy_(den)
        m.max_radii2D.copy_(torch.rand(P, generator=g) * 40)
        models.append(m)
    a, b = models
    ga = torch.Generator(device=DEV).manual_seed(77)
    gb = torch.Generator(device=DEV).manual_seed(77)
    gen0 = b.generation
    a.densify_and_prune_torch(0.0002, 0.05, 4.0, screen, generator=ga)
    b.densify_and_prune(0.0002, 0.05, 4.0, screen, generator=gb)
    assert b.generation == gen0 + 1
    Pa, Pb = a._xyz.shape[0], b._xyz.shape[0]
    assert Pa == Pb and Pa != P
    for n in ("_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        assert torch.equal(getattr(a, n).data, getattr(b, n).data), n
    assert torch.allclose(a._xyz.data, b._xyz.data, rtol=1e-6, atol=1e-6)
    same = (a._xyz.data == b._xyz.data).all(dim=1)
    assert same.float().mean() > 0.3          # originals and clones are copies
    assert torch.equal(a._exp_avg, b._exp_avg) and torch.equal(a._exp_avg_sq, b._exp_avg_sq)
    for t in (b.xyz_gradient_accum, b.denom, b.max_radii2D):
        assert t.shape[0] == Pb and float(t.abs().sum()) == 0.0
    assert b.viewspace_grad.shape == (Pb, 3) and b._xyz.grad.data_ptr() == b._grad.data_ptr()
    # the two generators consumed the same amount of the stream
    assert torch.equal(torch.rand(4, device=DEV, generator=ga), torch.rand(4, device=DEV, generator=gb))
