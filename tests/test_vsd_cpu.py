"""CPU checks of the VSD guidance algebra (NeTF stage) with stub networks.  The reference ships no tests
and its arithmetic lives in diffusers (absent) -> only what sd_vsd_utils.py itself states can be pinned."""
import torch

from garmentdreamer_amd.guidance import sd21
from garmentdreamer_amd.guidance.sd_vsd import LoraUnet, SpecifyGradient, StableDiffusionVSD
from tests.test_guidance_cpu import _StubVAE


class _Fixed(torch.nn.Module):
    def __init__(self, out):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.out = out
        self.seen = []

    def forward(self, x, t, encoder_hidden_states=None, c=None, shading=None, **kw):
        self.seen.append((tuple(x.shape), t.clone(), None if c is None else c.clone(), shading))
        return self.out.to(x.dtype)


def test_specify_gradient_divides_by_batch():
    lat = torch.zeros(3, 4, 2, 2, requires_grad=True)
    g = torch.arange(48.0).view(3, 4, 2, 2)
    loss = SpecifyGradient.apply(lat, g)
    assert loss.item() == g.sum().item()
    loss.backward()
    assert torch.equal(lat.grad, g / 3)


def test_vsd_gradient_algebra():
    gen = torch.Generator().manual_seed(0)
    e_c, e_u, v_q = (torch.randn(1, 4, 64, 64, generator=gen) for _ in range(3))
    unet = _Fixed(torch.cat([e_c, e_u]))
    q = _Fixed(v_q)
    gd = StableDiffusionVSD("cpu", fp16=False, unet=unet, vae=_StubVAE())
    assert (gd.min_step, gd.max_step) == (20, 500)
    pos, neg = torch.randn(1, 77, 1024, generator=gen), torch.randn(1, 77, 1024, generator=gen)
    gd.set_text_embeds(pos, neg)
    img = torch.rand(1, 3, 512, 512, requires_grad=True)
    noise = torch.randn(1, 4, 64, 64, generator=gen)
    t = torch.tensor([321])
    pose = torch.randn(1, 16, generator=gen)
    loss, pseudo, lat = gd.train_step(img, guidance_scale=7.5, q_unet=q, pose=pose, shading="albedo", noise=noise,
                                      timesteps=t)
    a = gd.alphas[t].view(1, 1, 1, 1)
    x_t = a.sqrt() * lat.detach() + (1 - a).sqrt() * noise
    eps_cfg = e_u + 7.5 * (e_c - e_u)                       # usual CFG form here (sd_vsd_utils.py:188-190)
    eps_q = a.sqrt() * v_q + (1 - a).sqrt() * x_t           # v-prediction -> eps (:199-207)
    grad = (1 - a) * (eps_cfg - eps_q)
    (g_lat,) = torch.autograd.grad(loss, lat, retain_graph=True)
    assert torch.allclose(g_lat, grad, rtol=1e-5, atol=1e-6)   # batch size 1 -> gt_grad / 1
    assert torch.allclose(pseudo, ((1 - a) * eps_cfg * lat.detach()).sum(), rtol=1e-4)
    loss.backward()
    assert img.grad is not None and img.grad.abs().sum() > 0
    # frozen UNet saw [x_t; x_t] with [t; t]; q-UNet saw x_t, t, the pose and the shading tag
    assert unet.seen[0][0] == (2, 4, 64, 64) and torch.equal(unet.seen[0][1], torch.tensor([321, 321]))
    assert q.seen[0][0] == (1, 4, 64, 64) and torch.equal(q.seen[0][2], pose) and q.seen[0][3] == "albedo"


def test_lora_unet_trains_only_adapters_and_matches_base_at_init():
    torch.manual_seed(0)
    kw = dict(block_out_channels=(32, 64, 64, 64), attention_head_dim=(1, 2, 2, 2), cross_attention_dim=64)
    lora = sd21.init_random_(sd21.LoraUNet2DConditionModel(**kw))
    base = sd21.UNet2DConditionModel(**kw)
    base.load_state_dict({k: v for k, v in lora.state_dict().items() if k in base.state_dict()})
    train = lora.freeze_base()
    names = {n for n, p in lora.named_parameters() if p.requires_grad}
    assert all(("lora" in n) or n.startswith("camera_emb") or n.endswith("_emb") for n in names)
    assert len(train) == len(names) and not lora.conv_in.weight.requires_grad
    x, t, c = torch.randn(2, 4, 16, 16), torch.tensor([10, 700]), torch.randn(2, 77, 64)
    # LoRA "up" matrices start at zero -> adapters are inert; with the camera MLP zeroed out too the
    # network equals the base UNet
    with torch.no_grad():
        for p in lora.camera_emb.parameters():
            p.zero_()
        assert torch.allclose(lora(x, t, c, c=torch.randn(2, 16), shading="albedo"), base(x, t, c), atol=1e-5)
    q = LoraUnet(lora)
    gd = StableDiffusionVSD("cpu", fp16=False, unet=base, vae=_StubVAE())
    gd.set_text_embeds(torch.randn(1, 77, 64), torch.randn(1, 77, 64))
    lat = torch.randn(1, 4, 16, 16)
    loss = gd.lora_train_loss(q, lat, torch.randn(1, 16), shading="normal", unet_bs=2, drop_pose=False)
    loss.backward()
    got = {n for n, p in lora.named_parameters() if p.grad is not None and p.grad.abs().sum() > 0}
    assert any("to_out_lora.up" in n for n in got) and any(n.startswith("camera_emb") for n in got)
    assert "normal_emb" in got and all(lora.get_parameter(n).requires_grad for n in got)
    # velocity target of the scheduler
    s = sd21.DDIMScheduler()
    x0, n, tt = torch.randn(2, 4, 4, 4), torch.randn(2, 4, 4, 4), torch.tensor([3, 900])
    a = s.alphas_cumprod[tt].view(-1, 1, 1, 1)
    assert torch.allclose(s.get_velocity(x0, n, tt), a.sqrt() * n - (1 - a).sqrt() * x0, atol=1e-6)
