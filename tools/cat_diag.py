import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import garmentdreamer_amd, torch
from garmentdreamer_amd.guidance import sd21
dev = "cuda:0"
with torch.device(dev):
    unet = sd21.init_random_(sd21.UNet2DConditionModel(), 1)
unet = unet.to(torch.bfloat16).to(memory_format=torch.channels_last).requires_grad_(False)
orig = torch.cat
def cat(ts, dim=0, **kw):
    ts = list(ts)
    if ts[0].dim() == 4:
        print("cat", [tuple(t.shape) for t in ts], [t.is_contiguous(memory_format=torch.channels_last) for t in ts], [t.stride() for t in ts])
    return orig(ts, dim, **kw)
torch.cat = cat
x = torch.randn(16, 4, 64, 64, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
t = torch.full((16,), 500, device=dev, dtype=torch.long)
ctx = torch.randn(16, 77, 1024, device=dev).to(torch.bfloat16)
with torch.no_grad():
    unet(x, t, encoder_hidden_states=ctx)
