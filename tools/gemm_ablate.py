#!/usr/bin/env python
"""Where the own GEMM's K loop spends its time: timing-only builds of csrc/nn_gemm.hip (tools/gemm_variants.sh) run on the big
shapes of the 8-view step, same process, interleaved.   python tools/gemm_ablate.py name1 name2 ..."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
import garmentdreamer_amd  # noqa: F401

names = sys.argv[1:] or ["base", "nodma", "noread", "nomfma", "nobar"]
libs = {}
for n in names:
    L = C.CDLL(f"tools/variants/libgd_gemm_{n}.so")
    L.gd_nn_gemm_forward.restype = C.c_int
    L.gd_nn_gemm_forward.argtypes = [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_int]
    L.gd_nn_gemm_geglu_forward.restype = C.c_int
    L.gd_nn_gemm_geglu_forward.argtypes = [C.c_void_p] * 5 + [C.c_int64, C.c_int, C.c_int]
    libs[n] = L


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


shapes = [(16384, 640, 5120), (4096, 1280, 10240), (16384, 2560, 640), (65536, 1280, 320), (8192, 8192, 8192), (4096, 4096, 4096)]
print("# us per call (TFLOP/s); variants: " + " ".join(names))
for M, K, N in shapes:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * K * N
    row = []
    best = {n: 1e30 for n in names}
    t_lib = 1e30
    for rep in range(3):             # interleaved, best of three (the first entry of a row otherwise pays the clock ramp)
        for n in names:
            L = libs[n]
            t = graph_time(lambda: L.gd_nn_gemm_forward(torch.cuda.current_stream().cuda_stream, x.data_ptr(), w.data_ptr(), None,
                                                        None, y.data_ptr(), M, K, N))
            best[n] = min(best[n], t)
        t_lib = min(t_lib, graph_time(lambda: torch.nn.functional.linear(x, w)))
    for n in names:
        row.append(f"{n} {best[n]:7.1f} ({fl / best[n] / 1e6:5.0f})")
    print(f"M{M:6d} K{K:5d} N{N:6d}: " + " | ".join(row) + f" | hipBLASLt {t_lib:7.1f} ({fl / t_lib / 1e6:5.0f})", flush=True)
