#!/usr/bin/env python
"""What does torch's flash SDPA on ROCm return as logsumexp, and does its backward accept an externally computed one?
(question behind using the own d64 forward kernel in the LoRA UNet's training pass)"""
import torch
torch.manual_seed(0)
B, H, S, D = 2, 5, 256, 64
q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
scale = D ** -0.5
out = torch.ops.aten._scaled_dot_product_flash_attention(q, k, v, 0.0, False, False, scale=scale)
print("n outputs", len(out), [getattr(o, "shape", o) for o in out[:2]], [type(o).__name__ for o in out])
o, lse = out[0], out[1]
s = (q.float() @ k.float().transpose(-1, -2)) * scale
ref_nat = torch.logsumexp(s, -1)
print("lse dtype", lse.dtype, "shape", tuple(lse.shape))
print("max |lse - natural|", (lse.float() - ref_nat).abs().max().item(), " max |lse - log2 form|", (lse.float() - ref_nat * 1.4426950408889634).abs().max().item())
do = torch.randn_like(o)
try:
    g = torch.ops.aten._scaled_dot_product_flash_attention_backward(do, q, k, v, o, ref_nat.contiguous(), out[2], out[3], out[4], out[5], 0.0, False, out[6], out[7], scale=scale)
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    of = torch.softmax((qf @ kf.transpose(-1, -2)) * scale, -1) @ vf
    of.backward(do.float())
    for name, a, b in zip("qkv", g, (qf.grad, kf.grad, vf.grad)):
        print("d" + name, "rel err with external natural-log LSE:", ((a.float() - b).abs().max() / b.abs().max()).item())
except Exception as e:
    print("backward with external LSE failed:", type(e).__name__, str(e)[:300])
