"""Stress of the streaming GEMM's counted waits: many launches on fresh data, alone and with another stream keeping the
memory system busy, every result checked against fp32 (a tile consumed before its DMA landed would be off by O(1))."""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
from garmentdreamer_amd import nn_ops

dev = "cuda:0"
g = torch.Generator(dev).manual_seed(0)
side = torch.cuda.Stream()
big = torch.randn(64 * 1024 * 1024, device=dev)
bad = 0
n = 0
for it in range(240):
    M = [65536, 16384, 4096 + 32 * (it % 7) + (it % 3), 32768][it % 4]
    N = [320, 640, 2560][it % 3]
    x = torch.randn(M, 320, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, 320, device=dev, generator=g) / 18).to(torch.bfloat16)
    b = torch.randn(N, device=dev, generator=g).to(torch.bfloat16)
    if it % 2:
        with torch.cuda.stream(side):       # competing HBM traffic
            for _ in range(4):
                big.mul_(1.0001)
    y = nn_ops.linear_320(x, w, b)
    ref = F.linear(x.float(), w.float(), b.float())
    err = (y.float() - ref).abs()
    ok = bool((err <= 2.0 ** -7 * ref.abs() + 1e-5).all())
    if N == 2560:
        z = nn_ops.linear_320_geglu(x, w, b)
        ok = ok and torch.equal(z, nn_ops.geglu(y))
    n += 1
    if not ok:
        rows = (err > 2.0 ** -7 * ref.abs() + 1e-5).any(dim=1).nonzero().flatten()
        print(f"it {it} M {M} N {N} busy {it % 2}: {rows.numel()} bad rows, first {rows[:6].tolist()} last {rows[-3:].tolist()}, max err {float(err.max()):.3f}", flush=True)
    bad += not ok
torch.cuda.synchronize()
print(f"{n} launches, {bad} wrong")
sys.exit(1 if bad else 0)
