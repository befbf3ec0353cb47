"""K = N = 320 nn.Linear on long row sets: weights-in-registers streaming kernel (csrc/nn_linear.hip) vs hipBLASLt."""
import sys
import time
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
from garmentdreamer_amd import nn_ops

DEV = "cuda:0"


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


g = torch.Generator(DEV).manual_seed(0)
for M, N in ((65536, 320), (32768, 320), (8192, 320), (65536 + 37, 320), (65536, 640), (65536, 2560), (8192, 2560)):
    x = torch.randn(M, 320, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, 320, device=DEV, generator=g) / 18).to(torch.bfloat16)
    b = torch.randn(N, device=DEV, generator=g).to(torch.bfloat16)
    with torch.no_grad():
        ref = F.linear(x.float(), w.float(), b.float())
        lib = F.linear(x, w, b)
        own = nn_ops.linear_320(x, w, b)
        e_own = (own.float() - ref).abs().max().item() / ref.abs().max().item()
        e_lib = (lib.float() - ref).abs().max().item() / ref.abs().max().item()
        t0 = timeit(lambda: F.linear(x, w, b))
        t1 = timeit(lambda: nn_ops.linear_320(x, w, b))
    gb = M * (320 + N) * 2 / 1e3
    print(f"M {M:6d} N {N:4d}: hipBLASLt {t0:6.1f} us ({gb / t0 / 1e3:4.2f} TB/s)   own {t1:6.1f} us ({gb / t1 / 1e3:4.2f} TB/s)   "
          f"rel err own {e_own:.1e} lib {e_lib:.1e}")

M = 65536
x = torch.randn(M, 320, device=DEV, generator=g).to(torch.bfloat16)
w = (torch.randn(2560, 320, device=DEV, generator=g) / 18).to(torch.bfloat16)
b = torch.randn(2560, device=DEV, generator=g).to(torch.bfloat16)
with torch.no_grad():
    t_lib = timeit(lambda: nn_ops.geglu(F.linear(x, w, b)))
    t_two = timeit(lambda: nn_ops.geglu(nn_ops.linear_320(x, w, b)))
    t_one = timeit(lambda: nn_ops.linear_320_geglu(x, w, b))
print(f"GEGLU(320, 1280) on {M} rows: hipBLASLt + geglu {t_lib:6.1f} us   own GEMM + geglu {t_two:6.1f} us   fused {t_one:6.1f} us")
