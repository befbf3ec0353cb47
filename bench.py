#!/usr/bin/env python
"""bench.py -- SDS iterations/sec of the GarmentDreamer inner loop on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full SDS iteration on synthetic data (BASELINE.json metric): rasterize V views
(fwd) -> bilinear 512^2 -> VAE encode (fwd, with grad) -> UNet on 2V samples (no grad) -> SDS loss
-> VAE dgrad -> rasterize bwd -> gradient all-reduce (N > 1) -> Adam.  Workload = BASELINE.json
configs[3]: 100k Gaussians x 8 views @ 512^2, the 8 views sharded V/N per GPU ("strong" scaling:
total work is fixed).  Random-init SD-2.1 weights (no network for checkpoints), bf16.

``--gpus N`` with N > 1 and no torchrun environment makes this process the launcher: it re-executes itself under
``torch.distributed.run`` with N ranks, one per GPU (backend nccl = RCCL).  In every mode the run is REFUSED (non-zero
exit, nothing on stdout) when WORLD_SIZE != --gpus or fewer than --gpus devices are visible; ``n_gpus`` in the line is
the process group's size and ``rccl_ranks`` an all-reduced counter (tests/test_bench_launch.py, gloo, stubbed step).

Rank 0 prints ONE JSON line.  Extra objects on that line:
  roofline      the DOMINANT kernel of the step = the hand-written bf16 MFMA conv3x3 kernel (largest share
                of GPU time): algorithmic FLOPs (2*N*H*W*Cout*9*Cin per launch) / its summed launch
                duration, from HIP events recorded around every launch inside the timed region; peak =
                2.5 PFLOP/s dense bf16 MFMA.
  roofline_raster_bwd  the rasterizer backward blend kernel (the kernel north_star names): algorithmic
                FLOPs per launch (14 per visited pair + 87 per contributing pair of the reference's
                backward, counted on the GPU from n_contrib / pair_counts; derivation in DESIGN.md) /
                its average launch duration from HIP events; peak = 157.3 TFLOP/s fp32 VALU.
  roofline_dense  UNet + VAE part: analytic FLOPs (0.804 TF/UNet sample, 1.117 TF/VAE image, dgrad
                = 1x fwd) / event time of guidance fwd + its backward, vs the 2.5 PF bf16 MFMA roof.
  cpu_baseline  the CPU oracle (rasterizer fwd+bwd, 1 thread) + fp32 PyTorch-CPU UNet/VAE for ONE of
                the 8 views, scaled to iterations/s (rank 0, N=1 only; skip with --no-cpu-baseline).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import garmentdreamer_amd  # noqa: E402,F401  (first: sets a HIP runtime flag before the runtime starts, _runtime_env.py)

import torch  # noqa: E402

UNET_TFLOP_PER_SAMPLE = 0.80425746432     # FlopCounterMode on sd21.UNet2DConditionModel @64x64 latents
VAE_TFLOP_PER_IMAGE = 1.116658466816      # FlopCounterMode on sd21.AutoencoderKLEncoder @512x512
FLOPS_VISITED_PAIR = 14                   # backward.cu:522-532 per visited (pixel, Gaussian) pair
FLOPS_CONTRIB_PAIR = 87                   # backward.cu:534-598 per contributing pair (incl. 10 adds)
PEAK_FP32_TFLOPS = 157.3                  # MI355X fp32 vector = f32-input MFMA rate (MI355X_MICROARCH.md)
PEAK_BF16_TFLOPS = 2500.0                 # dense bf16 MFMA
# What the VALU can ISSUE, measured on this chip (tools/probes/valu_rate_probe.hip): a wave64 v_pk_fma_f32 holds
# its SIMD 5.3 cycles -> 2 FMA x 2 flop x 64 lanes / 5.3 cycles x 1024 SIMDs x 2.4 GHz; the 157.3 datasheet figure would need 4.0
MEASURED_PACKED_FP32_TFLOPS = 2 * 2 * 64 / 5.3 * 1024 * 2.4e9 / 1e12
METRIC = "SDS iters/sec (rasterize+UNet+bwd), 100k Gaussians ×8 views @512², 1/2/4/8 GPU"


class Telemetry:
    """Shader clock and socket power of THIS rank's GPU during the timed region, sampled from the amdgpu hwmon files by a side
    thread (outside the step: two small sysfs reads every 20 ms).  The convolution family that dominates the step is
    power-bound (profiles/r04_regw_ablation.txt), so two boxes of the pool differ by several % on the same tree: the line
    carries what the box held while it was timed."""

    def __init__(self, device_index: int):
        self.samples = []          # (sclk MHz, power W)
        self.source = None
        self._stop = None
        self._thread = None
        try:
            self._hwmon = self._find(device_index)
        except Exception as e:     # telemetry is reporting only
            self._hwmon, self.source = None, f"unavailable: {type(e).__name__}: {e}"

    @staticmethod
    def _cards():
        import glob
        out = []
        for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            if os.path.exists(os.path.join(h, "freq1_input")) and os.path.exists(os.path.join(h, "power1_input")):
                out.append(h)
        return out

    def _find(self, device_index: int):
        cards = self._cards()
        if not cards:
            raise FileNotFoundError("no amdgpu hwmon with freq1_input + power1_input under /sys/class/drm")
        bus = None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            bus = "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            pass
        if bus is not None:
            for h in cards:
                real = os.path.realpath(os.path.join(h, "..", ".."))      # .../0000:05:00.0
                if os.path.basename(real).lower().startswith(bus):
                    self.source = f"{h} (PCI {bus})"
                    return h
        if len(cards) == 1:
            self.source = cards[0]
            return cards[0]
        self.source = "busiest of %d amdgpu hwmon nodes (PCI id of the HIP device not matched)" % len(cards)
        return cards           # sample all, keep the one that drew the most power

    @staticmethod
    def _read(h):
        with open(os.path.join(h, "freq1_input")) as f:
            mhz = int(f.read()) / 1e6
        with open(os.path.join(h, "power1_input")) as f:
            w = int(f.read()) / 1e6
        return mhz, w

    def start(self):
        if self._hwmon is None:
            return
        import threading
        self._stop = threading.Event()
        many = isinstance(self._hwmon, list)
        per = {h: [] for h in self._hwmon} if many else None

        def run():
            while not self._stop.is_set():
                try:
                    if many:
                        for h in self._hwmon:
                            per[h].append(self._read(h))
                    else:
                        self.samples.append(self._read(self._hwmon))
                except Exception:
                    pass
                self._stop.wait(0.02)
            if many:
                best = max(per.values(), key=lambda v: sum(w for _, w in v) if v else 0.0)
                self.samples = best
        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()

    def stop(self) -> dict:
        if self._thread is not None:
            self._stop.set()
            self._thread.join()
        n = len(self.samples)
        if n == 0:
            return {"clock_mhz_mean": None, "power_w_mean": None, "samples": 0, "source": self.source}
        return {"clock_mhz_mean": sum(m for m, _ in self.samples) / n, "clock_mhz_min": min(m for m, _ in self.samples),
                "power_w_mean": sum(w for _, w in self.samples) / n, "power_w_max": max(w for _, w in self.samples),
                "samples": n, "period_ms": 20, "source": self.source,
                "what": "amdgpu hwmon freq1_input (sclk) / power1_input (socket) during the timed region"}


def kernel_source_hash() -> str:
    """sha256 over the HIP sources + headers of both libraries: identifies WHICH kernels a counter file measured
    (.git does not travel to the GPU box, so a commit id cannot be read there)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "garmentdreamer_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def latest_profile(pattern: str):
    """Newest profiles/rNN_<pattern> (by round number), or None."""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + pattern)))
    return c[-1] if c else None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--gaussians", type=int, default=100000)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true")
    ap.add_argument("--per-view-raster", action="store_true", help="reference-style Python loop over views")
    ap.add_argument("--raster-only", action="store_true", help="time only rasterizer fwd+bwd (diagnostic)")
    ap.add_argument("--no-graphs", action="store_true", help="eager launches instead of hipGraph replay of UNet/VAE")
    ap.add_argument("--fp8", action="store_true",
                    help="e4m3 3x3 convolutions in the no-grad UNet forward (second, non-headline line: dtype says so)")
    ap.add_argument("--torch-adam", action="store_true",
                    help="six nn.Parameters + torch.optim.Adam(fused) instead of the flat-buffer GaussianModel")
    ap.add_argument("--nn-lib", default=None, help="A/B timing: another build of libgd_nn.so (recorded in the line's config)")
    ap.add_argument("--raster-lib", default=None, help="A/B timing: another build of libgd_raster.so (recorded likewise)")
    ap.add_argument("--host-sync-raster", action="store_true",
                    help="A/B: the rasterizer forward pass reads its instance count back to the host every iteration (the "
                         "reference's one stream synchronisation, rasterizer_impl.cu:282) instead of the capacity-bounded "
                         "sync-free form that is the default since round 5")
    ap.add_argument("--batch-invariant", action="store_true",
                    help="SDSLoop(batch_invariant=True): kernels (and bf16 summation orders) selected for the WHOLE camera batch "
                         "on every rank, so a sharded run reproduces the single-rank gradients per view bit for bit; off by "
                         "default (each rank tunes its launches for its own share); recorded in the line's config")
    ap.add_argument("--simulate-world", type=int, default=0,
                    help="diagnostic, one process: with --batch-invariant route the kernels as rank 0 of a K-rank run would (what "
                         "the bit-for-bit sharded mode costs a rank with 1/K of the views); recorded in the line's config")
    ap.add_argument("--stub-step", action="store_true",
                    help="TEST ONLY (tests/test_bench_launch.py): replace the SDS iteration by one small all-reduce so the "
                         "launch / rank-accounting logic of --gpus N can be exercised on CPU over gloo; the line it prints "
                         "is labelled metric='stub' and is not a measurement")
    ap.add_argument("--vsd", action="store_true",
                    help="BASELINE configs[4] diagnostic: NeTF VSD iteration (VAE + 2 frozen UNet + LoRA UNet fwd, "
                         "LoRA UNet fwd+bwd) on a synthetic 512^2 render, one view per GPU")
    return ap.parse_args()


def observed_host_syncs(step_fn) -> dict:
    """One extra, untimed step under torch's sync debug mode: the calls of the step in which the HOST waited for the GPU (pageable
    copies, .item(), nonzero, ...), by source line.  It sees torch's own calls; the rasterizer's read-back of the instance count goes
    through the C-ABI and is reported separately (``raster_forward_host_syncs_in_timed_region``)."""
    import collections
    import warnings
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("warn")
    try:
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            step_fn()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    sites = collections.Counter(f"{os.path.basename(w.filename)}:{w.lineno}" for w in caught if "synchroniz" in str(w.message))
    return {"count": sum(sites.values()), "sites": dict(sites)}


def camera_batch(args, step, view_ids):
    from garmentdreamer_amd import cameras as gcam
    # SURVEY 8d benchmark default orbit; the ring of azimuths rotates a little every step
    return gcam.orbit_batch(args.views, elevation_deg=15.0, camera_distance=2.75, fovy_deg=55.0, height=args.res,
                            width=args.res, azimuth_offset_deg=7.0 * step, view_ids=view_ids)


def count_pairs(loop, batch, device):
    """Visited / contributing pair counts of this rank's views (one extra untimed forward)."""
    import ctypes as C
    from garmentdreamer_amd import _native
    from garmentdreamer_amd.cameras import Camera, CameraBatch
    from garmentdreamer_amd.diff_gaussian_rasterization import _C
    g = loop.gaussians
    cams = [Camera(batch["c2w_3dgs"][i], batch["fovy"][i], batch["height"], batch["width"], data_device="cpu")
            for i in range(batch["c2w_3dgs"].shape[0])]
    cb = CameraBatch(cams, device)
    with torch.no_grad():
        out = _C.rasterize_gaussians_batched(
            loop.bg, g.get_xyz, torch.Tensor([]), g.get_opacity, g.get_scaling, g.get_rotation, 1.0,
            torch.Tensor([]), cb.viewmatrix, cb.projmatrix, cb.tanfovx, cb.tanfovy, cb.image_height, cb.image_width,
            g.get_features, g.active_sh_degree, cb.campos, False, False)
    R, _c, _d, _a, radii, geom, binning, img = out
    V, P = radii.shape
    lay = _native.Layout()
    _native.lib().gd_raster_get_layout(geom.data_ptr(), img.data_ptr(), binning.data_ptr(), P, V, cb.image_width,
                                       cb.image_height, R, C.byref(lay))
    npix = V * cb.image_height * cb.image_width
    n_contrib = img[lay.n_contrib:lay.n_contrib + 4 * npix].view(torch.int32)
    pc = img[lay.pair_counts:lay.pair_counts + 8 * npix].view(torch.int32).view(npix, 2)
    return dict(num_rendered=int(R), visible=int((radii > 0).sum()), pairs_visited_bwd=int(n_contrib.sum(dtype=torch.int64)),
                pairs_visited_fwd=int(pc[:, 0].sum(dtype=torch.int64)), pairs_contrib=int(pc[:, 1].sum(dtype=torch.int64)),
                views=V)


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args):
    """CPU oracle rasterizer (1 thread, and tiles over all host cores with OpenMP) + fp32 PyTorch-CPU VAE/UNet for
    ONE view, scaled to the whole iteration.  Test infrastructure used as the reported baseline only (never as the
    product path).  ``value`` uses the multi-core rasterizer time; the single-thread time is reported beside it."""
    from garmentdreamer_amd.guidance import sd21
    from oracle import gd_oracle
    from tests import helpers as h
    nthreads = torch.get_num_threads()
    ncores = os.cpu_count() or nthreads
    inp = h.raster_inputs(P=args.gaussians, H=args.res, W=args.res, seed=0, azimuth=-157.5, elevation=15.0,
                          distance=2.75, fovy_deg=55.0)
    gc, gd, ga = h.random_image_grads(args.res, args.res)
    t_raster = {}
    for omp in (False, True):
        t0 = time.perf_counter()
        st = gd_oracle.forward(inp["bg"], inp["means3D"], inp["colors_precomp"], inp["opacities"], inp["scales"],
                               inp["rotations"], inp["scale_modifier"], inp["cov3D_precomp"], inp["viewmatrix"],
                               inp["projmatrix"], inp["tanfovx"], inp["tanfovy"], inp["image_height"],
                               inp["image_width"], inp["sh"], inp["degree"], inp["campos"], omp=omp)
        gd_oracle.backward(st, gc, gd, ga)
        t_raster[omp] = time.perf_counter() - t0
    torch.manual_seed(0)
    vae = sd21.init_random_(sd21.AutoencoderKLEncoder()).float().eval()
    unet = sd21.init_random_(sd21.UNet2DConditionModel()).float().eval()
    for p in list(vae.parameters()) + list(unet.parameters()):
        p.requires_grad_(False)
    img = torch.rand(1, 3, 512, 512, requires_grad=True)
    t0 = time.perf_counter()
    lat = vae.encode(img * 2 - 1).latent_dist.sample() * 0.18215
    with torch.no_grad():
        eps = unet(torch.cat([lat.detach()] * 2), torch.tensor([500, 500]), torch.randn(2, 77, 1024))
    (lat * eps[:1].detach()).sum().backward()
    t_dense = time.perf_counter() - t0
    t_view = t_raster[True] + t_dense
    return {"value": 1.0 / (args.views * t_view), "unit": "iters/s", "cores": nthreads, "kind": "port",
            "cpu_model": _cpu_model(), "host_cores": ncores,
            "raster_1thread_s_per_view": t_raster[False], "raster_openmp_s_per_view": t_raster[True],
            "dense_fp32_s_per_view": t_dense,
            "value_1thread_raster": 1.0 / (args.views * (t_raster[False] + t_dense)),
            "sample": (f"1 of {args.views} views: oracle rasterizer fwd+bwd {args.gaussians} Gaussians @{args.res}^2 "
                       f"(1 thread {t_raster[False]:.2f} s; OpenMP over tiles, {ncores} cores, {t_raster[True]:.2f} s) + "
                       f"fp32 torch-CPU VAE fwd/dgrad + UNet x2 fwd ({nthreads} threads, {t_dense:.2f} s); "
                       f"scaled x{args.views}")}


def vsd_main(args):
    """NeTF texture-stage iteration (Garment_Deformer_NeTF/netf/trainer.py:158-256) with the mesh render
    replaced by a synthetic 512^2 image leaf (nvdiffrast + tiny-cuda-nn are out of scope)."""
    from garmentdreamer_amd import dist as gdist
    from garmentdreamer_amd.guidance import sd21
    from garmentdreamer_amd.guidance.sd_vsd import LoraUnet, StableDiffusionVSD
    check_world(args, int(os.environ.get("WORLD_SIZE", "1")), need_gpus=True)
    rk, lr, ws = gdist.init_from_env()
    device = torch.device("cuda", lr % torch.cuda.device_count())   # (ranks may share a GPU only under GD_DIST_BACKEND=gloo)
    torch.cuda.set_device(device)
    gd = StableDiffusionVSD(device, fp16=True, use_hip_graphs=not args.no_graphs, fp8_unet=bool(args.fp8))
    with torch.device(device):
        lora = sd21.init_random_(sd21.LoraUNet2DConditionModel(), 2)
    lora = lora.to(torch.bfloat16).to(memory_format=torch.channels_last)
    lora.trainables_to_fp32()        # the reference trains adapters, camera MLP and shading embeddings in fp32 (sd_vsd_utils.py:35); base weights bf16
    train = lora.freeze_base()
    q = LoraUnet(lora)
    # trainer.py:137 steps torch.optim.Adam over the adapters + embeddings: here one launch over the flat fp32 adapter buffer
    # (garmentdreamer_amd/flat_adam.py; constructed BEFORE the training graphs are captured -- it re-seats the adapters)
    from garmentdreamer_amd.flat_adam import FlatAdam
    opt = FlatAdam.for_lora_unet(lora, train, lr=1e-4) if os.environ.get("GD_FLAT_ADAM", "1") != "0" else torch.optim.Adam(train, lr=1e-4)
    g = torch.Generator(device=device).manual_seed(7 + rk)
    gd.set_text_embeds(torch.randn(1, 77, 1024, device=device, generator=g),
                       torch.randn(1, 77, 1024, device=device, generator=g))
    # configs[4] "1024^2 renders": the reference's NeTF stage asserts a 512^2 VAE input (sd_vsd_utils.py:146), so a
    # 1024^2 render is reduced bilinearly first -- the Garment_3DGS convention (stable_diffusion_guidance.py:394-396)
    res = args.res if args.res in (512, 1024) else 512
    img = torch.rand(1, 3, res, res, device=device, generator=g, requires_grad=True)
    bucket = None
    if args.warmup < gd.fp8_calibration_steps + 3 and args.fp8:
        args.warmup = gd.fp8_calibration_steps + 3      # calibration forwards + graph capture stay outside the timed region

    import contextlib
    _LORA_STREAM = os.environ.get("GD_VSD_LORA_STREAM", "1") != "0"

    def step():
        nonlocal bucket
        pose = torch.randn(1, 16, device=device, generator=g)
        x = img if res == 512 else torch.nn.functional.interpolate(img, (512, 512), mode="bilinear", align_corners=False)
        loss, _, latents = gd.train_step(x, guidance_scale=7.5, q_unet=q, pose=pose, shading="albedo")
        img.grad = None
        loss.backward()
        lu = gd.lora_train_loss(q, latents, pose, shading="albedo", unet_bs=1)
        opt.zero_grad(set_to_none=True)
        # the adapters' backward pass + optimizer step on the guidance's LoRA stream (the two lines a maintainer
        # wraps in trainer.py:250-256); GD_VSD_LORA_STREAM=0: on the caller's stream like the reference
        with (gd.lora_stream() if _LORA_STREAM else contextlib.nullcontext()):
            lu.backward()
            if gdist.collectives_on():
                grads = [p.grad for p in train if p.grad is not None]
                if bucket is None:
                    bucket = gdist.GradBucket(grads)
                bucket.all_reduce_mean_(grads)
            opt.step()

    for _ in range(args.warmup):
        step()
    tele = Telemetry(device.index)
    gdist.barrier()
    torch.cuda.synchronize()
    tele.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    gdist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    telemetry = tele.stop()
    # health (untimed): a replayed graph that went wrong shows up as non-finite adapters / image gradient
    with torch.no_grad():
        health = {"lora_params_finite": all(bool(torch.isfinite(p).all()) for p in train),
                  "image_grad_finite": bool(img.grad is not None and torch.isfinite(img.grad).all()),
                  "image_grad_nonzero": bool(img.grad is not None and float(img.grad.abs().sum()) > 0)}
    if not all(health.values()):
        raise SystemExit(f"bench.py --vsd: unhealthy run {health}")
    host_syncs = observed_host_syncs(step)
    if rk == 0:
        tfl = 3 * UNET_TFLOP_PER_SAMPLE + 2 * VAE_TFLOP_PER_IMAGE + 3 * UNET_TFLOP_PER_SAMPLE
        emit({"metric": "NeTF VSD iters/sec (VAE + 3 UNet fwd + LoRA-UNet fwd/bwd), 512^2, 1 view/GPU",
                          "value": ws * args.steps / el, "unit": "view-iters/s", "n_gpus": ws, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None,
                          "dtype": "bf16" if not args.fp8 else "bf16 + fp8(e4m3) 3x3 convolutions of the three no-grad UNet forwards",
                          "data": "synthetic",
                          "config": {"workload": f"VSD step, SD-2.1 UNet + LoRA UNet (rank 4) random-init, batch 1, {res}^2 image leaf"
                                                 + (" reduced to 512^2 for the VAE" if res != 512 else ""),
                                     "hip_graphs": bool(gd.use_hip_graphs), "fp8_unet": bool(gd.fp8_unet),
                                     "streams": ("three (caller: VAE + frozen UNet; LoRA no-grad forward; LoRA training pass + optimizer "
                                                 "step inside guidance.lora_stream())" if _LORA_STREAM else
                                                 "three inside the guidance, the reference's unedited call sequence"),
                                     "torch_host_syncs_per_step_observed": host_syncs["count"],
                                     "torch_host_sync_sites": host_syncs["sites"],
                                     "fp8_sites_run": sum(getattr(n, "fp8").sites_run for n in (gd.unet, lora)
                                                          if getattr(n, "fp8", None) is not None)},
                          "health": health, "telemetry": telemetry,
                          "roofline_dense": {"bound": "mfma", "achieved": tfl / (el / args.steps),
                                             "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                             "frac": tfl / (el / args.steps) / PEAK_BF16_TFLOPS}})
    if gdist.is_dist():
        torch.distributed.destroy_process_group()


_RESULT_OUT = None


def _claim_stdout():
    """The driver reads ONE JSON line from stdout.  Native libraries write there too -- RCCL prints a five-line version
    banner to the C stdout of every process that creates a communicator, flushed at exit, i.e. AFTER the result -- so
    the process's fd 1 is pointed at stderr for everything except the result line, which goes to a private duplicate
    of the original stdout."""
    global _RESULT_OUT
    if _RESULT_OUT is None:
        sys.stdout.flush()
        _RESULT_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    return _RESULT_OUT


def emit(line: dict):
    out = _claim_stdout()
    out.write(json.dumps(line) + "\n")
    out.flush()


def launch_ranks_if_needed(args):
    """``python bench.py --gpus N`` with N > 1 and no torchrun environment: become the launcher.  The process re-executes
    itself under ``torch.distributed.run`` with one rank per GPU (the form the driver uses for N > 1), so that the same
    command line that measures N = 1 measures N GPUs -- never N-fold one GPU.  Under an external torchrun (WORLD_SIZE
    set) this is a no-op; in every mode ``check_world`` then refuses a world that is not ``--gpus``."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    # --standalone: the launcher's own c10d rendezvous on a port IT binds (port 0) -- no bind-then-close window in which
    # another process could take a pre-selected port; --local-addr: the container's hostname may not resolve
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    sys.stderr.flush()
    os.execve(sys.executable, cmd, env)


def check_world(args, ws: int, need_gpus: bool):
    """The number of ranks IS --gpus, and every rank has its own GPU; anything else is refused (exit code 2) instead of
    measured under a wrong label."""
    if ws != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has WORLD_SIZE={ws}; launch one rank per GPU "
                         f"(python bench.py --gpus {args.gpus} launches them itself)")
    if need_gpus:
        n = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n < args.gpus and os.environ.get("GD_DIST_BACKEND") != "gloo":
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {n} GPU(s) are visible")


def rank_accounting(device, local_rank: int) -> dict:
    """What the process group actually is, from the group itself: ``n_gpus`` = its size, ``rccl_ranks`` = an all-reduced
    counter (each rank adds 1 through the backend the step uses), ``distinct_devices`` = how many different local
    device ordinals the ranks run on."""
    from garmentdreamer_amd import dist as gdist
    if not gdist.is_dist():
        return {"n_gpus": 1, "rccl_ranks": 1, "distinct_devices": 1, "backend": None}
    import torch.distributed as td
    one = torch.ones(1, device=device, dtype=torch.int32)
    td.all_reduce(one, op=td.ReduceOp.SUM)
    mine = torch.tensor([local_rank if device.type == "cuda" else -1 - td.get_rank()], device=device, dtype=torch.int32)
    every = [torch.zeros_like(mine) for _ in range(td.get_world_size())]
    td.all_gather(every, mine)
    return {"n_gpus": td.get_world_size(), "rccl_ranks": int(one.item()),
            "distinct_devices": len({int(t.item()) for t in every}), "backend": td.get_backend()}


def stub_main(args):
    """--stub-step: the launch + accounting skeleton of main() around a step that is one 1 KiB all-reduce.  CPU / gloo."""
    from garmentdreamer_amd import dist as gdist
    rk, lr, ws = gdist.init_from_env("gloo" if not torch.cuda.is_available() else None)
    check_world(args, ws, need_gpus=False)
    device = torch.device("cpu")
    buf = torch.ones(256)
    for _ in range(args.warmup):
        gdist.all_reduce_mean_(buf)
    gdist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gdist.all_reduce_mean_(buf)
    gdist.barrier()
    el = time.perf_counter() - t0
    acct = rank_accounting(device, lr)
    if rk == 0:
        emit({"metric": "stub", "value": args.steps / el, "unit": "iters/s", "n_gpus": acct["n_gpus"],
              "rccl_ranks": acct["rccl_ranks"], "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
              "dtype": "none", "data": "none", "config": {"workload": "stub step (one 1 KiB all-reduce); NOT a measurement",
                                                          "backend": acct["backend"]}})
    if gdist.is_dist():
        torch.distributed.destroy_process_group()


def main():
    args = parse()
    launch_ranks_if_needed(args)
    _claim_stdout()
    if args.nn_lib:
        from garmentdreamer_amd import nn_ops as _nn
        _nn.use_library(args.nn_lib)
    if args.raster_lib:
        from garmentdreamer_amd import _native as _nat
        _nat.use_library(args.raster_lib)
    if args.stub_step:
        return stub_main(args)
    if args.cpu_baseline_only:
        emit(cpu_baseline(args))
        return
    if args.vsd:
        return vsd_main(args)
    from garmentdreamer_amd import _native, dist as gdist
    from garmentdreamer_amd.guidance.stable_diffusion_guidance import PromptEmbeddings, StableDiffusionGuidance
    from garmentdreamer_amd.scene import GaussianParams, synthetic_gaussians
    from garmentdreamer_amd.sds_loop import SDSLoop

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP rasterizer has no CPU path")
    check_world(args, int(os.environ.get("WORLD_SIZE", "1")), need_gpus=True)   # before any device / group is touched
    rk, lr, ws = gdist.init_from_env()
    device = torch.device("cuda", lr % torch.cuda.device_count())   # (ranks may share a GPU only under GD_DIST_BACKEND=gloo)
    torch.cuda.set_device(device)
    _native.lib()  # fail loudly right here if the HIP library is missing
    torch.backends.cuda.matmul.allow_tf32 = False
    from garmentdreamer_amd import _runtime_env
    if not args.no_graphs and not args.raster_only and not _runtime_env.graph_replay_safe():
        raise SystemExit("bench.py: hipGraph replay is not available in this process (" + _runtime_env.FLAG + "=0 was not in "
                         "place before the HIP runtime started) -- the headline configuration replays the UNet / VAE as "
                         "graphs; fix the environment or pass --no-graphs to measure the eager path explicitly")

    view_ids = gdist.shard_views(args.views, rk, ws)
    scene = synthetic_gaussians(args.gaussians, seed=0, sh_degree=0)
    if args.torch_adam:
        gaussians = GaussianParams(scene, sh_degree=0, device=device)
    else:   # flat parameter / gradient buffers, one HIP Adam launch, native densification statistics
        from garmentdreamer_amd.gaussian_model import GaussianModel
        gaussians = GaussianModel.from_activated(scene, sh_degree=0, device=device)
    bg = torch.ones(3, device=device)
    if args.raster_only:
        guidance, prompt = None, None
    else:
        guidance = StableDiffusionGuidance({"guidance_scale": 100.0, "grad_clip": [0, 1.5, 2.0, 1000],
                                            "use_hip_graphs": not args.no_graphs, "fp8_unet": bool(args.fp8)},
                                           device=device)
        prompt = PromptEmbeddings.random(device)
    loop = SDSLoop(gaussians, guidance, prompt, bg, batch_invariant=bool(args.batch_invariant),
                   sync_free=not args.host_sync_raster,
                   route_as=(args.simulate_world, 0) if args.simulate_world > 1 else None)
    if args.per_view_raster:
        from garmentdreamer_amd.gaussian_renderer import render

        def per_view(cb, pc, bgc):
            pk = [render(c, pc, None, bgc) for c in _cams_on(cb, device)]
            return {"render": torch.stack([p["render"] for p in pk]), "viewspace_points": _VS(pk),
                    "radii": torch.stack([p["radii"] for p in pk]),
                    "depth_3dgs": torch.stack([p["depth_3dgs"] for p in pk]),
                    "alpha": torch.stack([p["alpha"] for p in pk]), "visibility_filter": None}
        loop.render_batch_fn = per_view

    gen = torch.Generator(device=device)
    from garmentdreamer_amd import nn_ops as nn_ops_mod

    def one_step(step):
        batch = camera_batch(args, step, view_ids)
        V = len(view_ids)
        gen.manual_seed(1234 + 1000 * step + rk)
        if args.raster_only:
            out = loop.render_views(batch)
            w = torch.randn(out["comp_rgb"].shape, device=device, generator=gen)
            loss = (out["comp_rgb"] * w).sum() + nn_ops_mod.sparsity_loss(out["depth"], out["depth_max"])
            if loop.native_scene:
                gaussians.zero_grad()
            else:
                loop.optimizer.zero_grad(set_to_none=True)
            loss.backward()
            return
        noise = torch.randn(V, 4, 64, 64, device=device, generator=gen)
        vae_noise = torch.randn(V, 4, 64, 64, device=device, generator=gen)
        t = torch.randint(20, 981, (V,), device=device, generator=gen)
        loop.step(batch, noise=noise, timesteps=t, vae_noise=vae_noise)

    from garmentdreamer_amd import nn_ops
    nn_ops.library_fallbacks(reset=True)       # counted over warm-up (where the graphs are captured) AND the timed region
    for s in range(args.warmup):
        one_step(s)
    torch.cuda.synchronize()
    _native.profile_reset()
    _native.profile_enable(True)
    if not args.raster_only:
        nn_ops.conv_profile(enable=True, reset=True)
    tele = Telemetry(device.index)
    gdist.barrier()
    torch.cuda.synchronize()
    tele.start()
    loop.time_collectives = ws > 1 and not args.raster_only      # HIP events either side of the gradient all-reduce
    t0 = time.perf_counter()
    for s in range(args.steps):
        one_step(args.warmup + s)
    torch.cuda.synchronize()
    gdist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    telemetry = tele.stop()
    coll_n, coll_ms, coll_bytes = loop.collective_times()
    loop.time_collectives = False
    fallbacks = nn_ops.library_fallbacks()
    _native.profile_enable(False)
    prof = _native.profile_read()
    conv_ms, conv_n, conv_flops = nn_ops.conv_profile(enable=False) if not args.raster_only else (0.0, 0, 0.0)
    conv_bytes = nn_ops.conv_profile_bytes() if not args.raster_only else 0.0
    host_syncs = observed_host_syncs(lambda: one_step(args.warmup + args.steps + 1000))     # untimed, after the counters are read
    conv_steps = args.steps
    conv_note = "HIP events around every launch inside the timed region"
    graphs_active = bool(guidance is not None and guidance.cfg.use_hip_graphs)   # False if capture failed / was refused
    if not args.raster_only and conv_n == 0:
        # The timed region replays the UNet / VAE as hipGraphs; launches inside a graph cannot be bracketed by
        # events, so the same kernels are timed in an eager pass of the same workload right after the timed region.
        guidance.cfg.use_hip_graphs = False
        nn_ops.conv_profile(enable=True, reset=True)
        conv_steps = 2
        for s in range(conv_steps):
            one_step(args.warmup + args.steps + s)
        torch.cuda.synchronize()
        conv_ms, conv_n, conv_flops = nn_ops.conv_profile(enable=False)
        conv_bytes = nn_ops.conv_profile_bytes()
        guidance.cfg.use_hip_graphs = graphs_active
        conv_note = ("HIP events around every launch in an eager (non-graph) pass of the same workload run right after "
                     "the timed region, whose UNet/VAE launches are replayed from hipGraphs")
    if ws > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- health (untimed): a fast step that has silently gone non-finite renders nothing and times "well" ----
    with torch.no_grad():
        params_finite = all(bool(torch.isfinite(p).all()) for p in gaussians.parameters())
    if not params_finite:
        raise SystemExit("bench.py: Gaussian parameters are not finite after the timed steps -- the measurement is void")

    # ---- work accounting (untimed) ----
    counts = count_pairs(loop, camera_batch(args, args.warmup, view_ids), device)
    if counts["num_rendered"] <= 0:
        raise SystemExit("bench.py: nothing is rendered after the timed steps -- the measurement is void")
    bwd_ms, bwd_n = prof["render_bwd"]
    flops_launch = FLOPS_VISITED_PAIR * counts["pairs_visited_bwd"] + FLOPS_CONTRIB_PAIR * counts["pairs_contrib"]
    roofline = None
    roofline_conv = None
    if conv_n > 0:
        ach = conv_flops / (conv_ms * 1e-3) / 1e12
        roofline_conv = {"bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / PEAK_BF16_TFLOPS, "traffic": None,
                         "kernel": ("conv3x3 family: conv3x3_wino_kernel (Winograd F(2,3) along x) + conv3x3_wide_kernel (128 ch x "
                                    "16x32 px) + conv3x3_gn_patch_kernel + conv3x3_patch_stream_kernel + conv3x3_nhwc_bf16_kernel"),
                         "avg_launch_us": conv_ms / conv_n * 1e3, "launches": conv_n,
                         "flops_per_launch": conv_flops / conv_n, "ms_per_step": conv_ms / max(conv_steps, 1),
                         "algorithmic_bytes_per_launch": conv_bytes / conv_n,
                         "timing": conv_note,
                         "note": ("dominant kernel family of the step (largest share of GPU time): the bf16 MFMA 3x3 "
                                  "convolution of the VAE encoder / UNet -- routed per shape (nn_ops._conv_route) between the "
                                  "Winograd F(2,3)-along-x kernel (2/3 of the multiplications of the direct form), the wide-tile "
                                  "direct kernel, the patch-staged direct kernels (GroupNorm+SiLU in the loader) and the "
                                  "implicit-GEMM kernel (stride 2, dgrad classes, split-K); algorithmic FLOPs = the DIRECT form's "
                                  "2*M*Cout*taps*Cin per launch whatever the kernel multiplies, summed over the launches of "
                                  "the timed region.  The MFMA stream alone (no loads / LDS reads / barriers) measures "
                                  "1.2-1.45 PFLOP/s on this chip with random data (power-limited clock, "
                                  "profiles/r01_conv_ablation.txt; round 4: two structurally different kernels of the 128 -> 128 layer both "
                                  "deliver ~1.0 PFLOP/s at 1360-1400 W of the 1400 W cap, profiles/r04_regw_ablation.txt), i.e. the practical "
                                  "ceiling is ~0.4-0.55 of `peak` by shape")}
    if bwd_n > 0:
        avg_s = bwd_ms / bwd_n * 1e-3
        ach = flops_launch / avg_s / 1e12
        V_ = counts["views"]
        # per (strip, entry) with a contribution (about 0.83 per list position on this scene -> counted as 1): the 16-byte
        # compact-list record, 40 B of centre / conic / colour gathered, one 40-byte row stored; 20 B per pixel
        alg_bytes = (16.0 + 40.0 + 40.0) * counts["num_rendered"] + 20.0 * V_ * args.res * args.res
        roofline = {"bound": "valu", "achieved": ach, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / PEAK_FP32_TFLOPS, "traffic": None,
                    "measured_issue_roof_tflops": MEASURED_PACKED_FP32_TFLOPS,
                    "frac_vs_measured_issue_roof": ach / MEASURED_PACKED_FP32_TFLOPS,
                    "kernel": "render_backward_block_kernel", "avg_launch_us": avg_s * 1e6, "launches": bwd_n,
                    "flops_per_launch": flops_launch, "algorithmic_bytes_per_launch": alg_bytes,
                    "note": ("fp32 VALU roof: 157.3 TF/s = the datasheet figure, which needs PACKED fp32 (v_pk_fma_f32) in "
                             "every slot: measured on this chip a wave64 v_fma_f32 holds its SIMD ~4.5 cycles, a "
                             "v_pk_fma_f32 ~5.3 (tools/probes/valu_rate_probe.hip), so plain-fp32 code tops out at "
                             "~70 TF/s and fully packed code at ~120; no MFMA is issued.  FLOPs are the REFERENCE "
                             "algorithm's (14 per pair its backward visits + 87 per contributing pair, "
                             "backward.cu:517-598); this kernel evaluates only the (entry, 4x4 block) cells the "
                             "forward pass's ballots name, two pixels per packed instruction")}

    # HBM bytes per launch from rocprofv3 PMC passes of this same command (tools/pmc_all.sh -> profiles/):
    # bench.py cannot read hardware counters itself, so `traffic` quotes the committed counter file
    kernels_per_step = None
    pmc_file = latest_profile("pmc.json")
    pmc_note = None
    workload = {"gaussians": args.gaussians, "views": args.views, "res": args.res}
    if pmc_file is not None:
        try:
            pmc = json.load(open(pmc_file))
            rel = os.path.relpath(pmc_file, ROOT)
            if pmc.get("kernel_source_hash") != kernel_source_hash():
                pmc_note = f"{rel} was collected on other kernel sources (hash {pmc.get('kernel_source_hash')}): not quoted"
            elif pmc.get("workload") != workload:
                pmc_note = f"{rel} was collected on another workload ({pmc.get('workload')}): not quoted"
            else:
                ks = pmc["kernels"]
                if roofline_conv is not None:
                    fam = {k: v for k, v in ks.items() if k.startswith("conv3x3_") and "hbm_bytes_per_launch" in v
                           and "first" not in k and "flip" not in k and "weights" not in k}
                    n = sum(v["launches_sampled"] for v in fam.values())
                    if n:
                        roofline_conv["traffic"] = sum(v["hbm_bytes_per_launch"] * v["launches_sampled"] for v in fam.values()) / n
                        roofline_conv["traffic_per_kernel"] = {k: v["hbm_bytes_per_launch"] for k, v in fam.items()}
                        roofline_conv["traffic_source"] = rel + " (launch-weighted mean over the family)"
                        # matrix-pipe utilisation and HBM GB/s against chip peak (north_star), per kernel and time-weighted
                        # over the family: mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) --
                        # 0.976 on an MFMA-only stream (profiles/r05_mfma_util_check.txt) --, hbm_gbps = counter bytes /
                        # duration, clock = GUI cycles per XCD / duration; all three under the profiler (tools/pmc_all.sh)
                        w = {k: v["launches_sampled"] * v.get("avg_duration_us_profiled", 0.0) for k, v in fam.items()}
                        tw = sum(w.values())
                        if tw > 0 and all("mfma_util" in v and "hbm_gbps" in v for v in fam.values()):
                            roofline_conv["mfma_util"] = sum(v["mfma_util"] * w[k] for k, v in fam.items()) / tw
                            roofline_conv["hbm_gbps"] = sum(v["hbm_gbps"] * w[k] for k, v in fam.items()) / tw
                            roofline_conv["hbm_frac_of_8TBps"] = roofline_conv["hbm_gbps"] / 8000.0
                            roofline_conv["clock_ghz_profiled"] = sum(v.get("clock_ghz", 0.0) * w[k] for k, v in fam.items()) / tw
                            roofline_conv["util_per_kernel"] = {k: {"mfma_util": round(v["mfma_util"], 4),
                                                                    "hbm_gbps": round(v["hbm_gbps"], 1),
                                                                    "clock_ghz": round(v.get("clock_ghz", 0.0), 3)}
                                                                for k, v in fam.items()}
                if roofline is not None:
                    for k, v in ks.items():
                        if k.startswith("render_backward") and "hbm_bytes_per_launch" in v:
                            roofline["traffic"] = v["hbm_bytes_per_launch"]
                            roofline["traffic_source"] = rel
                            for key in ("hbm_gbps", "clock_ghz", "mfma_util"):
                                if key in v:
                                    roofline[key if key != "clock_ghz" else "clock_ghz_profiled"] = v[key]
                            if "hbm_gbps" in v:
                                roofline["hbm_frac_of_8TBps"] = v["hbm_gbps"] / 8000.0
                            if v.get("SQ_INSTS_VALU") and v.get("SQ_BUSY_CYCLES"):
                                # SIMD-cycles of the launch: SQ_BUSY_CYCLES is summed over the 32 shader engines; 1024 SIMDs
                                simd_cycles = v["SQ_BUSY_CYCLES"] / 32.0 * 1024.0
                                # VALU issue utilisation: a wave64 VALU instruction holds its SIMD for ~4.5 cycles (measured,
                                # tools/probes/valu_rate_probe.hip; packed / transcendental ones longer)
                                roofline["valu_issue_util_min"] = 4.5 * v["SQ_INSTS_VALU"] / simd_cycles
                                roofline["valu_insts_per_launch"] = v["SQ_INSTS_VALU"]
        except Exception as e:
            pmc_note = f"counter file unreadable: {type(e).__name__}: {e}"
    if pmc_note is not None:
        for r in (roofline, roofline_conv):
            if r is not None:
                r["traffic_source"] = pmc_note
    steady = latest_profile("bench_N1_kernel_stats_steady.csv")
    if (steady is not None and workload == {"gaussians": 100000, "views": 8, "res": 512} and ws == 1 and not args.fp8
            and graphs_active and pmc_note is None and pmc_file is not None):   # the workload / tree that CSV was taken on
        try:
            import csv
            kernels_per_step = sum(float(r["CallsPerStep"]) for r in csv.DictReader(open(steady)))
        except Exception:
            pass
    acct = rank_accounting(device, lr)
    if acct["n_gpus"] != args.gpus or acct["rccl_ranks"] != args.gpus:
        raise SystemExit(f"bench.py: rank accounting {acct} does not match --gpus {args.gpus}")
    if rk == 0:
        V = len(view_ids)
        line = {
            "metric": METRIC, "value": args.steps / elapsed, "unit": "iters/s", "n_gpus": acct["n_gpus"],
            "rccl_ranks": acct["rccl_ranks"], "distinct_devices": acct["distinct_devices"],
            "collective_backend": acct["backend"], "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16" if not args.fp8 else "bf16 + fp8(e4m3) 3x3 convolutions of the no-grad UNet forward",
            "data": "synthetic",
            "config": {"workload": (f"SDS loop: {args.gaussians} Gaussians x {args.views} views @{args.res}^2, "
                                    f"SD-2.1 UNet+VAE random-init, {V} view(s)/GPU"),
                       "gaussians": args.gaussians, "views": args.views, "resolution": args.res,
                       "views_per_gpu": V, "parallelism": f"view-sharded dp{ws}",
                       "grad_allreduce": None if not coll_n else {
                           "ms_per_step": coll_ms * coll_n / args.steps, "bytes": coll_bytes, "calls_per_step": coll_n / args.steps,
                           "note": "HIP events on the step's stream either side of the ONE data-path collective (flat fp32 Gaussian "
                                   "gradients incl. the view-space tail, in place, blocking after the backward pass); rank 0's view"},
                       "raster": "per-view loop" if args.per_view_raster else "batched",
                       "raster_only": bool(args.raster_only), "hip_graphs": graphs_active,
                       "fp8_unet_sites": (guidance.unet.fp8.sites_run if guidance is not None and
                                          getattr(guidance.unet, "fp8", None) is not None else 0),
                       "kernels_per_step": kernels_per_step,
                       "raster_binning": ("tile-bucketed (count / scan / scatter / per-tile LDS counting sort; radix fallback beyond 4096 "
                                          "instances per tile)" if os.environ.get("GD_RASTER_BUCKETS", "1") != "0"
                                          else "global radix sort (GD_RASTER_BUCKETS=0)"),
                       "own_gemm_geglu": os.environ.get("GD_OWN_GEMM", "1") != "0",
                       "batch_invariant": bool(loop.batch_invariant),
                       "batch_invariant_route_scale": (loop._route_kr()[0] if loop.batch_invariant else 1),
                       "raster_forward_host_syncs_in_timed_region": (
                           0 if loop.capacity is not None and loop.capacity.calls_sync_free >= args.warmup + args.steps - 1
                           else args.steps),
                       "raster_instance_capacity": None if loop.capacity is None else loop.capacity.value,
                       "torch_host_syncs_per_step_observed": host_syncs["count"], "torch_host_sync_sites": host_syncs["sites"],
                       "library_fallbacks": sum(fallbacks.values()),
                       "library_fallback_sites": fallbacks or None,
                       "library_override": {"nn": args.nn_lib, "raster": args.raster_lib}
                       if (args.nn_lib or args.raster_lib) else None},
            "roofline": roofline_conv if roofline_conv is not None else roofline,
            "roofline_raster_bwd": roofline,
            "raster_kernels_ms_per_step": {k: v[0] / max(args.steps, 1) for k, v in prof.items()},
            "pair_counts_rank0": counts,
            "health": {"params_finite": params_finite, "visible_after_timed_steps": counts["visible"]},
            "telemetry": telemetry,
        }
        for r in (line["roofline"], line["roofline_raster_bwd"]):
            if r is not None:
                r["clock_mhz_mean"], r["power_w_mean"] = telemetry.get("clock_mhz_mean"), telemetry.get("power_w_mean")
        if not args.raster_only:
            dense_tflop = V * (2 * UNET_TFLOP_PER_SAMPLE + 2 * VAE_TFLOP_PER_IMAGE)
            raster_ms = sum(v[0] for v in prof.values()) / max(args.steps, 1)
            dense_s = max(elapsed / args.steps - raster_ms * 1e-3, 1e-9)
            line["roofline_dense"] = {"bound": "mfma", "achieved": dense_tflop / dense_s, "peak": PEAK_BF16_TFLOPS,
                                      "unit": "TFLOP/s", "frac": dense_tflop / dense_s / PEAK_BF16_TFLOPS,
                                      "tflop_per_step_per_gpu": dense_tflop,
                                      "note": "step time minus rasterizer kernel time; includes Adam, losses, host gaps"}
        if ws == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(args)
            except Exception as e:  # the baseline is reporting only; never fail the bench line on it
                line["cpu_baseline"] = {"value": None, "unit": "iters/s", "cores": 0, "kind": "port",
                                        "sample": f"failed: {type(e).__name__}: {e}"}
        emit(line)
    if gdist.is_dist():
        torch.distributed.destroy_process_group()


def _cams_on(cb, device):
    from garmentdreamer_amd.cameras import Camera
    return [Camera(torch.cat([c.R, c.T[:, None]], 1).new_zeros(4, 4).copy_(
        torch.cat([torch.cat([c.R, c.T[:, None]], 1), torch.tensor([[0., 0., 0., 1.]])], 0)), c.FoVy, c.image_height,
        c.image_width, data_device=device) for c in cb.cameras]


class _VS:
    """Adapter: per-view viewspace gradient holders exposed as one object with a stacked ``.grad``."""

    def __init__(self, pk):
        self.pk = pk

    @property
    def grad(self):
        return torch.stack([p["viewspace_points"].grad for p in self.pk])


if __name__ == "__main__":
    main()
