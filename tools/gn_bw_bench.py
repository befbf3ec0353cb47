"""Effective HBM bandwidth of the GroupNorm kernels on the VAE encoder's tensors (8 views)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import garmentdreamer_amd  # noqa
import torch
from garmentdreamer_amd import nn_ops

def ev_time(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n

L = nn_ops.lib()
SHAPES = [(8, 128, 512), (8, 256, 256), (8, 128, 256), (8, 512, 128), (8, 256, 128), (8, 512, 64), (16, 320, 64), (16, 640, 32), (16, 1280, 16)]
if os.environ.get("GN_SHAPE"):
    SHAPES = [SHAPES[int(os.environ["GN_SHAPE"])]]
for (N, C, H) in SHAPES:
    cl = torch.channels_last
    x = torch.randn(N, C, H, H, device="cuda").to(torch.bfloat16).contiguous(memory_format=cl)
    dy = torch.randn(N, C, H, H, device="cuda").to(torch.bfloat16).contiguous(memory_format=cl)
    add = torch.randn(N, C, H, H, device="cuda").to(torch.bfloat16).contiguous(memory_format=cl)
    gw = torch.ones(C, device="cuda", dtype=torch.bfloat16); gb = torch.zeros(C, device="cuda", dtype=torch.bfloat16)
    ws = nn_ops._gn_workspace(x, N, 32)
    mr = torch.empty(N * 64, dtype=torch.float32, device="cuda")
    y = torch.empty_like(x); dx = torch.empty_like(x); sums = torch.empty(N * 64, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    nb = x.numel() * 2
    t_stats = ev_time(lambda: L.gd_nn_groupnorm_stats(st, x.data_ptr(), N, H * H, C, 32, 1e-6, ws.data_ptr(), mr.data_ptr()))
    t_fwd = ev_time(lambda: L.gd_nn_groupnorm_silu_forward(st, x.data_ptr(), y.data_ptr(), gw.data_ptr(), gb.data_ptr(), N, H * H, C, 32, 1e-6, 1, ws.data_ptr(), mr.data_ptr()))
    t_bwd = ev_time(lambda: L.gd_nn_groupnorm_silu_backward(st, x.data_ptr(), dy.data_ptr(), gw.data_ptr(), gb.data_ptr(), mr.data_ptr(), dx.data_ptr(), N, H * H, C, 32, 1, ws.data_ptr(), sums.data_ptr(), None))
    t_bwda = ev_time(lambda: L.gd_nn_groupnorm_silu_backward(st, x.data_ptr(), dy.data_ptr(), gw.data_ptr(), gb.data_ptr(), mr.data_ptr(), dx.data_ptr(), N, H * H, C, 32, 1, ws.data_ptr(), sums.data_ptr(), add.data_ptr()))
    print(f"N{N} C{C:4d} @{H:3d} ({nb/1e6:6.1f} MB): stats {t_stats*1e6:6.1f} us {nb/t_stats/1e12:4.2f} TB/s | fwd (stats+apply, 3 passes) {t_fwd*1e6:6.1f} us {3*nb/t_fwd/1e12:4.2f} TB/s | "
          f"bwd (5 passes) {t_bwd*1e6:6.1f} us {5*nb/t_bwd/1e12:4.2f} TB/s | bwd+add (6 passes) {t_bwda*1e6:6.1f} us {6*nb/t_bwda/1e12:4.2f} TB/s", flush=True)
