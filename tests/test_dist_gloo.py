"""world_size-2 gloo tests of the view-sharded SDS loop (CPU).

The HIP rasterizer has no CPU path, so these tests drive ``SDSLoop`` with a small differentiable
stand-in for ``render_batch`` (pure torch) and a tiny SD-shaped UNet/VAE: what is under test is the
sharding + collectives -- ``shard_views``, ``GradBucket`` (one flat all-reduce, 1/N scaling),
``global_max`` (depth.max() across ranks, gradient routed to the owner), radii max-reduce -- by
checking that 2 ranks x 2 views reproduce 1 rank x 4 views.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from garmentdreamer_amd import cameras as gcam
from garmentdreamer_amd import dist as gdist
from garmentdreamer_amd.guidance import sd21
from garmentdreamer_amd.guidance.stable_diffusion_guidance import PromptEmbeddings, StableDiffusionGuidance
from garmentdreamer_amd.scene import GaussianParams, synthetic_gaussians
from garmentdreamer_amd.sds_loop import SDSLoop

V_TOTAL, P, HW = 4, 48, 16


def toy_render_batch(cb, pc, bg):
    """Differentiable dense splat in torch: isotropic blobs, order-independent (enough to exercise
    every gradient path the loop reduces: xyz, colour, opacity, scale, rotation, viewspace)."""
    xyz, V = pc.get_xyz, cb.viewmatrix.shape[0]
    vs = torch.zeros((V,) + tuple(xyz.shape), requires_grad=True) + 0
    vs.retain_grad()
    hom = torch.cat([xyz, torch.ones(xyz.shape[0], 1)], 1)
    ys, xs = torch.meshgrid(torch.arange(HW, dtype=torch.float32), torch.arange(HW, dtype=torch.float32), indexing="ij")
    imgs, deps, alps, rads = [], [], [], []
    col = torch.sigmoid(pc.get_features[:, 0, :])
    sig = pc.get_scaling.mean(1) * 40 + pc.get_rotation.pow(2).sum(1) * 0.0 + 1.0
    for v in range(V):
        pv = hom @ cb.viewmatrix[v]
        ph = hom @ cb.projmatrix[v]
        ndc = ph[:, :2] / (ph[:, 3:4] + 1e-7)
        mx = ((ndc[:, 0] + 1) * HW - 1) * 0.5 + vs[v, :, 0]
        my = ((ndc[:, 1] + 1) * HW - 1) * 0.5 + vs[v, :, 1]
        w = pc.get_opacity[:, 0] * torch.exp(-0.5 * ((xs[..., None] - mx) ** 2 + (ys[..., None] - my) ** 2) / sig ** 2)
        wsum = w.sum(-1)
        imgs.append((w @ col).permute(2, 0, 1) / (1 + wsum) + bg[:, None, None] / (1 + wsum))
        deps.append(((w * pv[:, 2]).sum(-1) / (1 + wsum))[None])
        alps.append((wsum / (1 + wsum))[None])
        rads.append((sig.detach() * 3 + (cb.campos[v, 0] * 3).round().abs()).to(torch.int32))  # view-dependent
    return {"render": torch.stack(imgs), "viewspace_points": vs, "radii": torch.stack(rads),
            "depth_3dgs": torch.stack(deps), "alpha": torch.stack(alps), "visibility_filter": None}


def build_loop():
    torch.manual_seed(0)
    unet = sd21.init_random_(sd21.UNet2DConditionModel(block_out_channels=(32, 32, 32, 32),
                                                       attention_head_dim=(1, 1, 1, 1), cross_attention_dim=1024))
    vae = sd21.init_random_(sd21.AutoencoderKLEncoder(block_out_channels=(32, 32, 32, 32)), seed=1)
    guidance = StableDiffusionGuidance({"half_precision_weights": False, "grad_clip": [0, 1.5, 2.0, 1000]},
                                       device="cpu", unet=unet, vae=vae)
    gaussians = GaussianParams(synthetic_gaussians(P, seed=3), device="cpu")
    loop = SDSLoop(gaussians, guidance, PromptEmbeddings.random("cpu"), torch.ones(3),
                   render_batch_fn=toy_render_batch, lr_scale=100.0)
    return loop


def run_steps(loop, view_ids, n_steps=1):
    g = torch.Generator().manual_seed(11)
    noise = torch.randn(n_steps, V_TOTAL, 4, 64, 64, generator=g)
    vnoise = torch.randn(n_steps, V_TOTAL, 4, 64, 64, generator=g)
    ts = torch.randint(20, 981, (n_steps, V_TOTAL), generator=g)
    outs = []
    for s in range(n_steps):
        batch = gcam.orbit_batch(V_TOTAL, height=HW, width=HW, azimuth_offset_deg=10.0 * s, view_ids=view_ids)
        outs.append(loop.step(batch, noise=noise[s, view_ids], timesteps=ts[s, view_ids], vae_noise=vnoise[s, view_ids]))
    return outs


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    rk, lr, ws = gdist.init_from_env(backend="gloo")
    assert (rk, ws) == (rank, world) and gdist.world_size() == world
    view_ids = gdist.shard_views(V_TOTAL, rk, ws)
    assert view_ids == list(range(rank, V_TOTAL, world))
    loop = build_loop()
    run_steps(loop, view_ids)
    # replicas must stay bit-identical (same reduced gradients, same Adam state)
    flat = torch.cat([p.detach().reshape(-1) for p in loop.params])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], g) for g in gathered)
    if rank == 0:
        ret["params"] = flat.clone()
        ret["grads"] = torch.cat([p.grad.detach().reshape(-1) for p in loop.params])
        ret["max_radii"] = loop.max_radii2D.clone()
        ret["accum"] = loop.xyz_gradient_accum.clone()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_reproduce_single_rank_iteration():
    torch.set_num_threads(4)
    loop = build_loop()
    run_steps(loop, list(range(V_TOTAL)))
    ref = torch.cat([p.detach().reshape(-1) for p in loop.params])
    ref_g = torch.cat([p.grad.detach().reshape(-1) for p in loop.params])
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    got, got_g = ret["params"], ret["grads"]
    # reduced gradients of 2 ranks x 2 views == gradients of 1 rank x 4 views
    assert ref_g.abs().max() > 1e-4
    # (tolerance: guidance_scale = 100 amplifies the fp32 batch-size-dependent rounding of the UNet's
    # eps_text - eps_uncond to ~1e-4 relative)
    assert torch.allclose(got_g, ref_g, rtol=5e-3, atol=1e-4 * float(ref_g.abs().max())), (got_g - ref_g).abs().max()
    # the parameters actually moved; Adam(eps=1e-15) turns noise-level gradients into +-lr steps, so
    # compare the update only where the gradient is well above fp noise
    init = torch.cat([p.detach().reshape(-1) for p in build_loop().params])
    assert (ref - init).abs().max() > 1e-3
    solid = ref_g.abs() > 1e-2 * ref_g.abs().max()
    assert torch.allclose(got[solid], ref[solid], rtol=1e-3, atol=1e-5), (got - ref)[solid].abs().max()
    assert torch.equal(ret["max_radii"], loop.max_radii2D)
    assert torch.allclose(ret["accum"], loop.xyz_gradient_accum, rtol=5e-3, atol=1e-4 * float(loop.xyz_gradient_accum.max()))


def _gmax_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    gdist.init_from_env(backend="gloo")
    x = torch.tensor([1.0, 5.0, 2.0] if rank == 0 else [3.0, 4.0, 0.5], requires_grad=True)
    m = gdist.global_max(x.max())
    loss = (x / (m + 1e-5)).sum()
    loss.backward()
    ret[rank] = (float(m), x.grad.clone())
    # the same maximum riding in the tail slot of the asynchronous radii collective (PendingMax)
    x2 = x.detach().clone().requires_grad_(True)
    radii = torch.tensor([3, 0, 7, 1] if rank == 0 else [2, 9, 0, 1], dtype=torch.int32)
    pend = gdist.PendingMax(radii, x2.max())
    side = (x2 * 2.0).sum()                       # work between start and finish
    radii_all, m2 = pend.finish()
    ((x2 / (m2 + 1e-5)).sum() + 0.0 * side).backward()
    ret[f"p{rank}"] = (float(m2), x2.grad.clone(), radii_all.clone())
    # the sync-free rasterizer's overflow flag as a third payload: raised on rank 1 only, seen by both (SDSLoop.step)
    x3 = x.detach().clone().requires_grad_(True)
    pend = gdist.PendingMax(radii, x3.max(), torch.tensor([rank], dtype=torch.int32))
    any_flag = pend.flag_any()
    radii_f, m3 = pend.finish()
    pend0 = gdist.PendingMax(radii, x3.max().detach(), torch.zeros(1, dtype=torch.int32))
    ret[f"f{rank}"] = (any_flag, float(m3), radii_f.clone(), pend0.flag_any(), gdist.PendingMax(radii, x3.max().detach()).flag_any())
    pend0.finish()
    b = gdist.GradBucket([torch.zeros(3), torch.zeros(2, 2)])
    a1, a2 = torch.full((3,), float(rank + 1)), torch.full((2, 2), 10.0 * (rank + 1))
    b.all_reduce_mean_([a1, a2])
    ret[f"b{rank}"] = (a1, a2)
    dist.destroy_process_group()


def test_global_max_routes_gradient_to_owner_and_bucket_averages():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gmax_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    # single-process reference: concatenated tensor
    x = torch.tensor([1.0, 5.0, 2.0, 3.0, 4.0, 0.5], requires_grad=True)
    (x / (x.max() + 1e-5)).sum().backward()
    assert ret[0][0] == 5.0 and ret[1][0] == 5.0
    assert torch.allclose(torch.cat([ret[0][1], ret[1][1]]), x.grad, rtol=1e-6)
    for r in (0, 1):
        m2, g2, radii_all = ret[f"p{r}"]
        assert m2 == 5.0 and torch.equal(g2, ret[r][1]) and radii_all.tolist() == [3, 9, 7, 1]
    for r in (0, 1):
        any_flag, m3, radii_f, none_raised, no_flag = ret[f"f{r}"]
        assert any_flag is True and m3 == 5.0 and radii_f.tolist() == [3, 9, 7, 1] and not none_raised and not no_flag
    for r in (0, 1):
        a1, a2 = ret[f"b{r}"]
        assert torch.equal(a1, torch.full((3,), 1.5)) and torch.equal(a2, torch.full((2, 2), 15.0))


def test_shard_views_contract():
    assert gdist.shard_views(8, 0, 1) == list(range(8))
    assert gdist.shard_views(8, 3, 4) == [3, 7]
    assert sorted(sum((gdist.shard_views(8, r, 8) for r in range(8)), [])) == list(range(8))
    with pytest.raises(ValueError):
        gdist.shard_views(8, 0, 3)
