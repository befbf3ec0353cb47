"""Diagnostic: LoRA-UNet training gradients -- fp32 torch vs bf16 (HIP kernels) vs bf16 (torch ops only)."""
import sys, torch
sys.path.insert(0, '.')
import garmentdreamer_amd
import torch.nn.functional as F
from garmentdreamer_amd.guidance import sd21
DEV = "cuda:0"
def cos(a, b): return F.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()
kw = dict(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4))
def build(dt):
    with torch.device(DEV):
        lora = sd21.init_random_(sd21.LoraUNet2DConditionModel(**kw), 2)
    lora = lora.to(dt).to(memory_format=torch.channels_last)
    train = lora.freeze_base()
    return lora, train
g = torch.Generator(DEV).manual_seed(1)
x0 = torch.randn(1, 4, 64, 64, device=DEV, generator=g)
ctx = torch.randn(1, 77, 1024, device=DEV, generator=g)
pose = torch.randn(1, 16, device=DEV, generator=g)
tgt = torch.randn(1, 4, 64, 64, device=DEV, generator=g)
t = torch.tensor([611], device=DEV)
def run(dt, hip=True):
    saved = {}
    if not hip:
        for n in ("conv3x3_supported", "gn_conv3x3_supported", "conv3x3_s2_supported", "upsample2x_conv3x3_supported",
                  "resnet_block_frozen_supported", "attention_d64_supported"):
            saved[n] = getattr(sd21, n); setattr(sd21, n, lambda *a, **k: False)
        saved["group_norm_silu"] = sd21.group_norm_silu
        def gns(x, w, b, groups, eps, silu=True):
            y = F.group_norm(x, groups, w, b, eps); return F.silu(y) if silu else y
        sd21.group_norm_silu = gns
    lora, train = build(dt)
    x = x0.clone().requires_grad_(True)
    out = lora(x.to(dt) if dt != torch.float32 else x, t, encoder_hidden_states=ctx.to(dt), c=pose.to(dt), shading="albedo").float()
    loss = F.mse_loss(out, tgt)
    loss.backward()
    torch.cuda.synchronize()
    for n, f in saved.items(): setattr(sd21, n, f)
    return out.detach(), x.grad.detach().float(), [p.grad.detach().float() if p.grad is not None else None for p in train], float(loss)
ref = run(torch.float32)
for label, r in (("bf16 HIP kernels", run(torch.bfloat16, True)), ("bf16 torch ops", run(torch.bfloat16, False))):
    cs = [cos(a, b) for a, b in zip(ref[2], r[2]) if a is not None and float(a.abs().max()) > 0]
    tot_a = torch.cat([a.flatten() for a in ref[2] if a is not None]); tot_b = torch.cat([b.flatten() for a, b in zip(ref[2], r[2]) if a is not None])
    cs.sort()
    print(f"{label}: loss {ref[3]:.5f} vs {r[3]:.5f}; cos out {cos(ref[0], r[0]):.6f}; cos dL/dx {cos(ref[1], r[1]):.6f}; "
          f"param-grad cos: min {cs[0]:.3f} p10 {cs[len(cs)//10]:.3f} median {cs[len(cs)//2]:.3f} global {cos(tot_a, tot_b):.4f} (n={len(cs)})")
