"""Time of one densify_and_prune event: the two HIP passes (GaussianModel.densify_and_prune) against the reference's
torch op sequence (densify_and_prune_torch) on the same scene and statistics.  python tools/densify_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import garmentdreamer_amd  # noqa
import numpy as np
import torch
from garmentdreamer_amd.gaussian_model import GaussianModel
from garmentdreamer_amd.scene import synthetic_gaussians

dev = torch.device("cuda", 0)


def make(P):
    sc = synthetic_gaussians(P, seed=3)
    rng = np.random.default_rng(5)
    sc["scales"] = (sc["scales"] * rng.uniform(0.5, 6.0, size=(P, 1))).astype(np.float32)
    m = GaussianModel.from_activated(sc, device=dev)
    g = torch.Generator().manual_seed(1)
    m.xyz_gradient_accum.copy_(torch.rand((P, 1), generator=g) * 6e-4)
    m.denom.fill_(1.0)
    return m


for P in (100000, 400000):
    for name in ("densify_and_prune", "densify_and_prune_torch"):
        ts = []
        for rep in range(6):
            m = make(P)
            gen = torch.Generator(device=dev).manual_seed(7)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            getattr(m, name)(0.0002, 0.05, 4.0, 20, generator=gen)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"P = {P}: {name:26s} first call {ts[0]*1e3:7.2f} ms, median of the next five {sorted(ts[1:])[2]*1e3:6.2f} ms -> P = {m._xyz.shape[0]}")
