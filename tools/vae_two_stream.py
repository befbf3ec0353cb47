"""Experiment: VAE encoder forward + backward for 8 views as ONE batch vs TWO half batches on two HIP streams (the
MFMA-bound convolutions of one half can overlap the HBM-bound GroupNorm passes of the other)."""
import sys
import time
import torch
sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
from garmentdreamer_amd.guidance import sd21

dev = "cuda:0"
torch.manual_seed(0)
vae = sd21.init_random_(sd21.AutoencoderKLEncoder()).to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
for p in vae.parameters():
    p.requires_grad_(False)
V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
img = torch.rand(V, 3, 512, 512, device=dev).to(torch.bfloat16)
gy = torch.randn(V, 4, 64, 64, device=dev).to(torch.bfloat16)


def one_batch():
    x = img.clone().requires_grad_(True)
    z = vae.encode(x).latent_dist.mean
    z.backward(gy)
    return x.grad


streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def two_streams():
    cur = torch.cuda.current_stream()
    grads = []
    xs = []
    for k, s in enumerate(streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            x = img[k * V // 2:(k + 1) * V // 2].clone().requires_grad_(True)
            z = vae.encode(x).latent_dist.mean
            xs.append((x, z))
    for k, s in enumerate(streams):
        with torch.cuda.stream(s):
            x, z = xs[k]
            z.backward(gy[k * V // 2:(k + 1) * V // 2])
            grads.append(x.grad)
    for s in streams:
        cur.wait_stream(s)
    return torch.cat(grads)


def halves_one_stream():
    grads = []
    for k in range(2):
        x = img[k * V // 2:(k + 1) * V // 2].clone().requires_grad_(True)
        z = vae.encode(x).latent_dist.mean
        z.backward(gy[k * V // 2:(k + 1) * V // 2])
        grads.append(x.grad)
    return torch.cat(grads)


def timeit(fn, n=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


g1 = one_batch()
g2 = two_streams()
print("max |grad diff| one batch vs two streams:", (g1.float() - g2.float()).abs().max().item(), "of", g1.float().abs().max().item())
for name, fn in (("one batch", one_batch), ("two halves, one stream", halves_one_stream), ("two halves, two streams", two_streams),
                 ("one batch", one_batch), ("two halves, two streams", two_streams)):
    print(f"{name:28s} {timeit(fn):7.2f} ms")
