"""fp8 implicit-GEMM kernel vs the bf16 paths on the UNet-forward shapes (16 samples): TFLOP/s per shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import garmentdreamer_amd  # noqa
import torch
import torch.nn.functional as F
from garmentdreamer_amd import nn_ops

DEV = "cuda:0"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def main():
    g = torch.Generator(DEV).manual_seed(0)
    print("linear  M      K     N      bf16(hipBLASLt) us  TF/s |  fp8 us  TF/s | quantize us")
    for M, K, N in [(65536, 320, 320), (65536, 320, 960), (65536, 320, 2560), (65536, 1280, 320),
                    (16384, 640, 640), (16384, 640, 1920), (16384, 640, 5120), (16384, 2560, 640),
                    (4096, 1280, 1280), (4096, 1280, 3840), (4096, 1280, 10240), (4096, 5120, 1280), (1024, 1280, 10240)]:
        x = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(torch.bfloat16)
        b = torch.randn(N, device=DEV, generator=g).to(torch.bfloat16)
        x8, w8 = nn_ops.fp8_quantize(x, 0.01), nn_ops.fp8_pack_weights(w, 0.001)
        t0 = timeit(lambda: F.linear(x, w, b))
        t1 = timeit(lambda: nn_ops.fp8_linear(x8, w8, b, None, K, 1e-5))
        t2 = timeit(lambda: nn_ops.fp8_quantize(x, 0.01))
        fl = 2.0 * M * K * N
        print(f"{M:7d} {K:5d} {N:6d}   {t0*1e6:9.1f} {fl/t0/1e12:7.0f}   | {t1*1e6:8.1f} {fl/t1/1e12:7.0f} | {t2*1e6:7.1f}")
    print("conv3x3  N  H   Cin  Cout    bf16 us  TF/s |  fp8 us  TF/s")
    for N, H, Cin, Cout in [(16, 64, 320, 320), (16, 64, 640, 320), (16, 64, 960, 320), (16, 32, 640, 640), (16, 32, 1280, 640),
                            (16, 32, 320, 640), (16, 16, 1280, 1280), (16, 16, 2560, 1280), (16, 16, 640, 1280), (16, 8, 1280, 1280)]:
        x = torch.randn(N, Cin, H, H, device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Cout, Cin, 3, 3, device=DEV, generator=g) / (9 * Cin) ** 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        b = torch.randn(N, Cout, device=DEV, generator=g).to(torch.bfloat16)
        x8 = nn_ops.fp8_quantize(x.permute(0, 2, 3, 1), 0.01).permute(0, 3, 1, 2)
        w8 = nn_ops.fp8_pack_weights(w.permute(0, 2, 3, 1).reshape(Cout * 9, Cin), 0.001)
        with torch.no_grad():
            t0 = timeit(lambda: nn_ops.conv3x3(x, w, b, None))
            t1 = timeit(lambda: nn_ops.fp8_conv3x3(x8, w8, b, None, Cin, 1e-5))
        fl = 2.0 * N * H * H * Cout * 9 * Cin
        print(f"{N:8d} {H:3d} {Cin:5d} {Cout:5d}  {t0*1e6:8.1f} {fl/t0/1e12:6.0f} | {t1*1e6:8.1f} {fl/t1/1e12:6.0f}")


if __name__ == "__main__":
    main()
