import sys, torch
sys.path.insert(0, "/root/repo")
import garmentdreamer_amd
from tests import test_configs_gpu as t
kw_u = dict(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4)); kw_v = dict(block_out_channels=(64, 64, 128, 128))
out = {}
for name, dt, f32 in (("fp32", torch.float32, False), ("bf16", torch.bfloat16, False), ("bf16+fp32adapters", torch.bfloat16, True), ("bf16+fp32adapters#2", torch.bfloat16, True)):
    gd, lora, train, q = t._vsd_objects(kw_u, kw_v, dt, fp32_adapters=f32)
    out[name] = t._vsd_step(gd, q, train, seed=9)
g32 = out["fp32"][3]
keys = [i for i in g32 if float(g32[i].abs().max()) > 0]
cat = lambda g: torch.cat([g[i].flatten().float() for i in keys])
for name in out:
    print(name, "cos all lora grads vs fp32:", t._cos(cat(g32), cat(out[name][3])), " cos dimg", t._cos(out["fp32"][0], out[name][0]))
print("bf16+fp32adapters run-to-run:", t._cos(cat(out["bf16+fp32adapters"][3]), cat(out["bf16+fp32adapters#2"][3])))
