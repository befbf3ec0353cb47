#!/usr/bin/env python
"""Weight-streaming small-map convolution (tools/experimental/nn_conv_stream.h -- a measured negative; NOT in
libgd_nn.so: build it with `tools/stream_variants.sh` and point GD_NN_LIB at ablate/libgd_nn_stream.so) against the split-K implicit-GEMM route on the UNet's 8^2 / 16^2
layers at 1, 2 and 4 latents; device time per call from a hipGraph of 20 calls.   python tools/stream_conv_bench.py [waves ...]
(GD_NN_STREAM_WAVES is read once per process: run the script once per value to sweep it)"""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import tools.ablib  # noqa: F401,E402
import garmentdreamer_amd  # noqa: F401
import ctypes as C
import os
from garmentdreamer_amd import nn_ops

L = nn_ops.lib()
_vp, _i = C.c_void_p, C.c_int
for _name, (_res, _args) in {
        "gd_nn_conv3x3_stream_supported": (_i, [_i, _i, _i, _i, _i]),
        "gd_nn_conv3x3_stream_weights_bytes": (C.c_size_t, [_i, _i]),
        "gd_nn_conv3x3_stream_weights": (_i, [_vp, _vp, _vp, _i, _i]),
        "gd_nn_conv3x3_stream_ws_bytes": (C.c_size_t, [_i, _i, _i, _i, _i]),
        "gd_nn_conv3x3_stream_forward": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, C.c_size_t])}.items():
    _fn = getattr(L, _name)          # AttributeError = the library was not built with -DGD_NN_EXPERIMENTAL_STREAM
    _fn.restype, _fn.argtypes = _res, _args

# GD_NN_STREAM=0: the small-map layers on the split-K implicit-GEMM kernel as before round 5 (same-box A/B; never set in tests)
_STREAM = os.environ.get("GD_NN_STREAM", "1") != "0"
_STREAM_MAX_M = int(os.environ.get("GD_NN_STREAM_MAX_M", "512"))


def _stream(weight):
    """Cached re-packing of a frozen conv weight into MFMA fragment order for the weight-streaming kernel (csrc/nn_conv_stream.h)."""
    u = getattr(weight, "_gd_stream", None)
    key = (weight.data_ptr(), weight._version)
    if u is None or u.device != weight.device or getattr(weight, "_gd_stream_key", None) != key:
        Cout, Cin = weight.shape[0], weight.shape[1]
        u = torch.empty(nn_ops.lib().gd_nn_conv3x3_stream_weights_bytes(Cout, Cin) // 2, dtype=torch.bfloat16, device=weight.device)
        with torch.cuda.device(weight.device):
            ret = nn_ops.lib().gd_nn_conv3x3_stream_weights(torch.cuda.current_stream(weight.device).cuda_stream,
                                                     weight.data_ptr(), u.data_ptr(), Cout, Cin)
        nn_ops._check(ret, "gd_nn_conv3x3_stream_weights", "gd_nn_conv_last_error")
        weight._gd_stream, weight._gd_stream_key = u, key
    return u


def _stream_route(N, H, W, Cin, Cout) -> bool:
    """Maps of a few hundred pixels under deep, wide filters (the UNet's 8^2 / 16^2 levels at one or two latents): the layer is
    bound by reading its filter bank once -- the weight-streaming kernel (tools/stream_conv_bench.py)."""
    return (_STREAM and nn_ops._ROUTE_SCALE == 1 and N * H * W <= _STREAM_MAX_M and Cin >= 320 and Cout >= 640 and
            bool(nn_ops.lib().gd_nn_conv3x3_stream_supported(N, H, W, Cin, Cout)))


def _stream_launch(x, w_khwc, bias, residual, out_channels):
    N, Cin, H, W = x.shape
    L = nn_ops.lib()
    y = torch.empty((N, out_channels, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    bias, stride = nn_ops._bias_and_stride(bias)
    wp = _stream(w_khwc)
    ws_bytes = L.gd_nn_conv3x3_stream_ws_bytes(N, H, W, Cin, out_channels)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        ret = L.gd_nn_conv3x3_stream_forward(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), wp.data_ptr(),
                                             None if bias is None else bias.data_ptr(), stride,
                                             None if residual is None else residual.data_ptr(), y.data_ptr(), N, H, W, Cin,
                                             out_channels, ws.data_ptr(), ws_bytes)
    nn_ops._check(ret, "gd_nn_conv3x3_stream_forward", "gd_nn_conv_last_error")
    return y





def graph_time(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * reps)


for N in (1, 2, 4):
    for ci, co, hw in ((1280, 1280, 8), (2560, 1280, 8), (640, 1280, 16), (1280, 1280, 16), (2560, 1280, 16), (1920, 1280, 16),
                       (320, 640, 8), (640, 640, 16)):
        if N * hw * hw > 512 or not L.gd_nn_conv3x3_stream_supported(N, hw, hw, ci, co):
            continue
        x = torch.randn(N, ci, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(co, ci, 3, 3, device="cuda") / (3 * ci ** 0.5)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        b = torch.randn(co, device="cuda").to(torch.bfloat16)
        fl = 2.0 * N * hw * hw * co * 9 * ci
        wb = 2.0 * co * 9 * ci
        with torch.no_grad():
            ref = nn_ops._conv_launch(x, w, b, None, co).float()
            t_old = graph_time(lambda: nn_ops._conv_launch(x, w, b, None, co))
            got = _stream_launch(x, w, b, None, co).float()
            t_new = graph_time(lambda: _stream_launch(x, w, b, None, co))
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        print(f"N{N} {ci:4d}->{co:4d} @{hw:2d} (M={N * hw * hw:3d}): split-K {t_old:6.1f} us | stream {t_new:6.1f} us {fl / t_new / 1e6:5.0f} TF "
              f"{wb / t_new / 1e6:5.2f} TB/s of filter  {t_old / t_new:.2f}x  err {err:.1e}", flush=True)
