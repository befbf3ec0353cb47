"""The round-5 review's item 1(b), measured before it is built: the 8-view SDS step as TWO 4-view chains on two HIP streams
of ONE process, so that one half's HBM-bound passes (GroupNorm, LayerNorm / GEGLU, elementwise) can sit under the other half's
MFMA-bound convolutions.  Timing experiment only: each half has its own scene replica, guidance instance (own hipGraphs, own
capture stream = own library-GEMM workspace, own GroupNorm accumulators) and optimizer; the cross-half gradient sum a product
version needs (one 9 MB add) is NOT included, i.e. the two-stream figure is a lower bound of what the built form would cost.

    python tools/sds_two_stream.py [--steps 20] [--warmup 4]

Prints, same process and box, interleaved repetitions:
    one chain x 8 views          (the headline workload)
    two chains x 4 views, one after the other on one stream
    two chains x 4 views on two streams
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import garmentdreamer_amd  # noqa: E402,F401
from garmentdreamer_amd import cameras as gcam, nn_ops  # noqa: E402
from garmentdreamer_amd.gaussian_model import GaussianModel  # noqa: E402
from garmentdreamer_amd.guidance.stable_diffusion_guidance import PromptEmbeddings, StableDiffusionGuidance  # noqa: E402
from garmentdreamer_amd.scene import synthetic_gaussians  # noqa: E402
from garmentdreamer_amd.sds_loop import SDSLoop  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=4)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--gaussians", type=int, default=100000)
ap.add_argument("--res", type=int, default=512)
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
VIEWS = 8


class Chain:
    """One SDS chain on `view_ids` of the 8-view ring, with everything it captures kept apart from the other chains'."""

    def __init__(self, view_ids, tag, stream):
        self.view_ids, self.tag, self.stream = list(view_ids), tag, stream
        self.cap = torch.cuda.Stream(device=dev)
        scene = synthetic_gaussians(args.gaussians, seed=0, sh_degree=0)
        self.gaussians = GaussianModel.from_activated(scene, sh_degree=0, device=dev)
        self.guidance = StableDiffusionGuidance({"guidance_scale": 100.0, "grad_clip": [0, 1.5, 2.0, 1000],
                                                 "use_hip_graphs": True}, device=dev)
        self.prompt = PromptEmbeddings.random(dev)
        self.loop = SDSLoop(self.gaussians, self.guidance, self.prompt, torch.ones(3, device=dev), densify=False)
        self.gen = torch.Generator(device=dev)

    def step(self, step):
        V = len(self.view_ids)
        batch = gcam.orbit_batch(VIEWS, elevation_deg=15.0, camera_distance=2.75, fovy_deg=55.0, height=args.res, width=args.res,
                                 azimuth_offset_deg=7.0 * step, view_ids=self.view_ids)
        self.gen.manual_seed(1234 + 1000 * step + self.view_ids[0])
        noise = torch.randn(V, 4, 64, 64, device=dev, generator=self.gen)
        vae_noise = torch.randn(V, 4, 64, 64, device=dev, generator=self.gen)
        t = torch.randint(20, 981, (V,), device=dev, generator=self.gen)
        prev = torch.cuda.graph.default_capture_stream
        torch.cuda.graph.default_capture_stream = self.cap       # only matters while this chain still captures
        try:
            with nn_ops.workspace_tag(self.tag):
                self.loop.step(batch, noise=noise, timesteps=t, vae_noise=vae_noise)
        finally:
            torch.cuda.graph.default_capture_stream = prev

    def healthy(self):
        return all(bool(torch.isfinite(p).all()) for p in self.gaussians.parameters())


quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
if quiet is not None:
    quiet(False)
main = torch.cuda.current_stream(dev)
sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
one8 = Chain(range(8), "one8", main)
halfA = Chain([0, 2, 4, 6], "halfA", sA)
halfB = Chain([1, 3, 5, 7], "halfB", sB)

# warm-up / capture, every chain on the stream it will be timed on
for s in range(args.warmup):
    one8.step(s)
torch.cuda.synchronize()
for ch in (halfA, halfB):
    with torch.cuda.stream(ch.stream):
        for s in range(args.warmup):
            ch.step(s)
    torch.cuda.synchronize()


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        fn(args.warmup + s)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.steps * 1e3


def run_one8(s):
    one8.step(s)


def run_serial(s):
    halfA.step(s)
    halfB.step(s)


def run_streams(s):
    with torch.cuda.stream(sA):
        halfA.step(s)
    with torch.cuda.stream(sB):
        halfB.step(s)


def timed_threads():
    """Each chain driven by its own host thread (a graph launch holds the calling thread while earlier work of the process
    drains -- DESIGN 3.13 -- so one thread feeding two streams may serialise them on the host side)."""
    import threading
    bar = threading.Barrier(3)

    def work(ch):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(ch.stream):
            bar.wait()
            for s in range(args.steps):
                ch.step(args.warmup + s)
        bar.wait()

    th = [threading.Thread(target=work, args=(ch,)) for ch in (halfA, halfB)]
    torch.cuda.synchronize()
    for t in th:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    bar.wait()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps * 1e3
    for t in th:
        t.join()
    return dt


rows = {"one chain x 8 views": [], "two chains x 4 views, one stream": [], "two chains x 4 views, two streams": [],
        "two chains x 4 views, two streams, two host threads": []}
for rep in range(args.reps):
    rows["one chain x 8 views"].append(timed(run_one8))
    rows["two chains x 4 views, one stream"].append(timed(run_serial))
    rows["two chains x 4 views, two streams"].append(timed(run_streams))
    rows["two chains x 4 views, two streams, two host threads"].append(timed_threads())
ok = one8.healthy() and halfA.healthy() and halfB.healthy()
print(f"# tools/sds_two_stream.py: {args.gaussians} Gaussians @{args.res}^2, {args.steps} timed steps per entry, ms per 8-view "
      f"iteration; parameters finite: {ok}")
for k, v in rows.items():
    print(f"{k:55s} " + "  ".join(f"{x:7.3f}" for x in v))
