#!/usr/bin/env python
"""128 -> 128 3x3 convolution with the filter bank in registers (tools/experimental/nn_conv_regw.h -- a measured negative,
DESIGN.md 3.11; since round 5 NOT in libgd_nn.so: build it with `tools/regw_variants.sh regw:` and point GD_NN_LIB at
ablate/libgd_nn_regw.so): parity against fp32 PyTorch and time against what the routing table picks today (wide tile).
usage: GD_NN_LIB=ablate/libgd_nn_regw.so python tools/regw_conv_bench.py [N H W]"""
import ctypes as C_
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import tools.ablib  # noqa: F401,E402
from garmentdreamer_amd import nn_ops  # noqa: E402

_vp, _i = C_.c_void_p, C_.c_int
for _name, (_res, _args) in {
        "gd_nn_conv3x3_regw_supported": (_i, [_i, _i, _i, _i, _i]),
        "gd_nn_conv3x3_regw_weights": (_i, [_vp, _vp, _vp]),
        "gd_nn_conv3x3_regw_weights_bytes": (C_.c_size_t, []),
        "gd_nn_conv3x3_regw_forward": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp])}.items():
    _fn = getattr(nn_ops.lib(), _name)          # AttributeError = the library was not built with -DGD_NN_EXPERIMENTAL_REGW
    _fn.restype, _fn.argtypes = _res, _args


def _regw(weight):
    """Cached re-packing of a frozen 128 -> 128 conv weight as the register fragments of csrc/nn_conv_regw.h."""
    u = getattr(weight, "_gd_regw", None)
    key = (weight.data_ptr(), weight._version)
    if u is None or u.device != weight.device or getattr(weight, "_gd_regw_key", None) != key:
        u = torch.empty(nn_ops.lib().gd_nn_conv3x3_regw_weights_bytes() // 2, dtype=torch.bfloat16, device=weight.device)
        with torch.cuda.device(weight.device):
            ret = nn_ops.lib().gd_nn_conv3x3_regw_weights(torch.cuda.current_stream(weight.device).cuda_stream,
                                                   weight.data_ptr(), u.data_ptr())
        nn_ops._check(ret, "gd_nn_conv3x3_regw_weights", "gd_nn_conv_last_error")
        weight._gd_regw, weight._gd_regw_key = u, key
    return u


def _regw_launch(x, w_khwc, bias, residual, out_channels, stat_part=None):
    """3x3/s1/p1 convolution, 128 -> 128 channels, filter bank resident in registers (csrc/nn_conv_regw.h).  Not on the
    default route (parity with the wide tile); tools/regw_conv_bench.py and the GPU tests call it."""
    N, Cin, H, W = x.shape
    L = nn_ops.lib()
    y = torch.empty((N, out_channels, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    bias, stride = nn_ops._bias_and_stride(bias)
    u = _regw(w_khwc)
    with torch.cuda.device(x.device):
        ret = L.gd_nn_conv3x3_regw_forward(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), u.data_ptr(),
                                           None if bias is None else bias.data_ptr(), stride,
                                           None if residual is None else residual.data_ptr(), y.data_ptr(), N, H, W, Cin,
                                           out_channels, None if stat_part is None else stat_part.data_ptr())
    nn_ops._check(ret, "gd_nn_conv3x3_regw_forward", "gd_nn_conv_last_error")
    return y



nn_ops._regw_launch = _regw_launch

N, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 512, 512)
dev = "cuda"
torch.manual_seed(0)
C = 128


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def check(name, got, ref):
    d = (got.float() - ref).abs()
    scale = ref.abs().max().item()
    cos = F.cosine_similarity(got.float().flatten().double(), ref.flatten().double(), dim=0).item()
    print(f"  {name}: max err {d.max().item() / scale:.3e} of scale, cos {cos:.7f}")
    return d.max().item() / scale, cos


for (n, h, w_) in ((1, 32, 64), (2, 48, 96), (N, H, W)):
    x = torch.randn(n, C, h, w_, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, C, 3, 3, device=dev) / (3 * C ** 0.5)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = torch.randn(C, device=dev).to(torch.bfloat16)
    res = torch.randn(n, C, h, w_, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    print(f"N {n} {h}x{w_}")
    big = n * h * w_ > 1 << 20
    with torch.no_grad():
        if not big:
            ref = F.conv2d(x.float(), w.float(), b.float(), padding=1)
            y = nn_ops._regw_launch(x, w, b, None, C)
            check("plain", y, ref)
            # epilogue statistics: partial rows against the sums of the stored tensor
            rows = ((h + 15) // 16) * ((w_ + 15) // 16) * 8
            part = torch.full((n, C // 4, rows, 2), float("nan"), device=dev)
            y = nn_ops._regw_launch(x, w, b, None, C, stat_part=part)
            yq = y.float().permute(0, 2, 3, 1).reshape(n, h * w_, C // 4, 4)
            s1, s2 = yq.sum((1, 3)), (yq * yq).sum((1, 3))
            g1, g2 = part[..., 0].sum(2), part[..., 1].sum(2)
            print(f"  stats: finite {bool(torch.isfinite(part).all())}, sum err {((g1 - s1).abs().max() / s1.abs().max()).item():.2e}, "
                  f"sumsq err {((g2 - s2).abs().max() / s2.abs().max()).item():.2e}")
        else:
            y = nn_ops._regw_launch(x, w, b, None, C)
            y2 = nn_ops._wide_launch(x, w, b, None, C)
            check("plain vs wide kernel", y, y2.float())
        t_new = timeit(lambda: nn_ops._regw_launch(x, w, b, None, C))
        t_old = timeit(lambda: nn_ops._wide_launch(x, w, b, None, C))
        fl = 2.0 * n * h * w_ * C * C * 9
        print(f"  regw {t_new:8.1f} us {fl / t_new / 1e6:6.0f} TF | wide {t_old:8.1f} us {fl / t_old / 1e6:6.0f} TF | {t_old / t_new:.2f}x")
