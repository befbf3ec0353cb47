"""Stand-in for the reference's pybind module ``diff_gaussian_rasterization._C``.

Exports the same three callables with the same positional signatures and return tuples as
``DGR/ext.cpp:15-19`` / ``DGR/rasterize_points.{h,cu}`` (DGR = Garment_3DGS/gaussiansplatting/
submodules/diff-gaussian-rasterization), implemented over the C-ABI of ``libgd_raster.so``
(``include/gd_raster.h``) with torch tensors only as device memory + stream providers.

Differences from the reference glue, all deliberate:
  * scratch byte tensors are allocated on ``means3D.device`` (reference: ``torch::kCUDA``
    default device, rasterize_points.cu:73-77) and kernels run on torch's *current* stream
    (reference: legacy default stream);
  * backward outputs are ``torch.empty`` -- the library writes every element -- instead of ten
    ``torch::zeros`` memsets (rasterize_points.cu:155-164);
  * errors come back as ``RuntimeError`` with the library's message (no C++ exceptions).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from .. import _native

NUM_CHANNELS = 3


def _ptr(t):
    """Device pointer or NULL for an absent optional (the reference passes torch.Tensor([]))."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def _f32c(t: torch.Tensor, device) -> torch.Tensor:
    if t is None or t.numel() == 0:
        return t
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32 tensor, got {t.dtype}")
    if t.device != device:
        raise RuntimeError(f"tensor on {t.device}, expected {device}")
    return t.contiguous()


# GD_RASTER_POISON_SCRATCH=1 (the GPU tests set it): every scratch buffer and output the library is handed starts as
# 0xFF bytes -- NaN floats, -1 integers -- instead of whatever the caching allocator returns, so that a kernel relying on
# an element it never wrote fails its parity test.  Off in production: the library writes everything it reads.
_POISON = os.environ.get("GD_RASTER_POISON_SCRATCH") == "1"


def _empty(shape, dtype, device):
    t = torch.empty(shape, dtype=dtype, device=device)
    if _POISON and t.numel():
        t.view(torch.uint8).fill_(0xFF)
    return t


class _Scratch:
    """Python side of the C-ABI allocator callback (replaces resizeFunctional,
    rasterize_points.cu:27-33)."""

    def __init__(self, device):
        self.device = device
        self.tensor = torch.empty(0, dtype=torch.uint8, device=device)
        self.cb = _native.ALLOC_FN(self._alloc)

    def _alloc(self, _user, nbytes):
        self.tensor = _empty(int(nbytes), torch.uint8, self.device)
        return self.tensor.data_ptr()


def _require_gpu(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError(
            "garmentdreamer_amd rasterizer runs on MI355X only: tensors must live on a HIP device "
            f"(got {t.device}); there is no CPU path in the product.")


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug):
    """RasterizeGaussiansCUDA (rasterize_points.cu:35-119).  Returns
    ``(num_rendered, color, depth, alpha, radii, geomBuffer, binningBuffer, imgBuffer)``."""
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    _require_gpu(means3D)
    dev = means3D.device
    L = _native.lib()
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    out_color = _empty((NUM_CHANNELS, H, W), torch.float32, dev)
    out_depth = _empty((1, H, W), torch.float32, dev)
    out_alpha = _empty((1, H, W), torch.float32, dev)
    radii = _empty((P,), torch.int32, dev)
    geom, binning, img = _Scratch(dev), _Scratch(dev), _Scratch(dev)
    M = sh.size(1) if (sh is not None and sh.numel() != 0) else 0
    keep = [_f32c(t, dev) for t in (background, means3D, sh, colors, opacity, scales, rotations, cov3D_precomp,
                                    viewmatrix, projmatrix, campos)]
    bg, m3, shc, col, opa, scl, rot, cov, vm, pm, cp = keep
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        rendered = L.gd_raster_forward(
            stream, geom.cb, None, binning.cb, None, img.cb, None, P, int(degree), M, _ptr(bg), W, H, _ptr(m3),
            _ptr(shc), _ptr(col), _ptr(opa), _ptr(scl), float(scale_modifier), _ptr(rot), _ptr(cov), _ptr(vm),
            _ptr(pm), _ptr(cp), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)), out_color.data_ptr(),
            out_depth.data_ptr(), out_alpha.data_ptr(), radii.data_ptr() if P else None, int(bool(debug)))
    _native.check(rendered, "gd_raster_forward")
    return rendered, out_color, out_depth, out_alpha, radii, geom.tensor, binning.tensor, img.tensor


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth,
                                 dL_dout_alpha, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, alphas,
                                 debug):
    """RasterizeGaussiansBackwardCUDA (rasterize_points.cu:121-208).  Returns
    ``(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)``."""
    _require_gpu(means3D)
    dev = means3D.device
    L = _native.lib()
    P = means3D.size(0)
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    M = sh.size(1) if (sh is not None and sh.numel() != 0) else 0
    mk = lambda *shape: _empty(shape, torch.float32, dev)
    dL_dmeans3D, dL_dmeans2D, dL_dcolors = mk(P, 3), mk(P, 3), mk(P, NUM_CHANNELS)
    dL_dconic, dL_dopacity, dL_dcov3D = mk(P, 2, 2), mk(P, 1), mk(P, 6)
    dL_ddepths = mk(P, 1)
    dL_dsh, dL_dscales, dL_drotations = mk(P, M, 3), mk(P, 3), mk(P, 4)
    if P == 0:
        return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations
    has_scales = scales is not None and scales.numel() != 0
    if not has_scales:  # outputs the kernel does not produce on this path stay defined (zeros)
        dL_dscales.zero_()
        dL_drotations.zero_()
    scratch = _empty(L.gd_raster_backward_scratch_bytes(P, 1, int(R)), torch.uint8, dev)
    keep = [_f32c(t, dev) for t in (background, means3D, sh, colors, alphas, scales, rotations, cov3D_precomp,
                                    viewmatrix, projmatrix, campos, dL_dout_color, dL_dout_depth, dL_dout_alpha)]
    bg, m3, shc, col, alp, scl, rot, cov, vm, pm, cp, gcol, gdep, galp = keep
    radii_c = radii.contiguous()
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        ret = L.gd_raster_backward(
            stream, P, int(degree), M, int(R), _ptr(bg), W, H, _ptr(m3), _ptr(shc), _ptr(col), _ptr(alp), _ptr(scl),
            float(scale_modifier), _ptr(rot), _ptr(cov), _ptr(vm), _ptr(pm), _ptr(cp), float(tan_fovx),
            float(tan_fovy), radii_c.data_ptr(), geomBuffer.data_ptr(), _ptr(binningBuffer) or geomBuffer.data_ptr(),
            imageBuffer.data_ptr(), scratch.data_ptr(), _ptr(gcol), _ptr(gdep), _ptr(galp), dL_dmeans2D.data_ptr(),
            dL_dconic.data_ptr(), dL_dopacity.data_ptr(), dL_dcolors.data_ptr(), dL_ddepths.data_ptr(),
            dL_dmeans3D.data_ptr(), dL_dcov3D.data_ptr(), _ptr(dL_dsh), dL_dscales.data_ptr(),
            dL_drotations.data_ptr(), int(bool(debug)))
    _native.check(ret, "gd_raster_backward")
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible (rasterize_points.cu:210-229)."""
    _require_gpu(means3D)
    dev = means3D.device
    P = means3D.size(0)
    present = torch.zeros((P,), dtype=torch.bool, device=dev)
    if P != 0:
        m3, vm, pm = (_f32c(t, dev) for t in (means3D, viewmatrix, projmatrix))
        with torch.cuda.device(dev):
            ret = _native.lib().gd_raster_mark_visible(torch.cuda.current_stream(dev).cuda_stream, P, m3.data_ptr(),
                                                       vm.data_ptr(), pm.data_ptr(), present.data_ptr())
        _native.check(ret, "gd_raster_mark_visible")
    return present


# ---------------------------------------------------------------------------------------------
# Batched multi-view entry (no reference counterpart)
# ---------------------------------------------------------------------------------------------

def _farr(vals):
    arr = (C.c_float * len(vals))(*[float(v) for v in vals])
    return arr


class InstanceCapacity:
    """Caller-side state of the SYNC-FREE batched forward pass (``gd_raster_forward_batched_capacity``, include/gd_raster.h).

    The reference reads ``num_rendered`` back to the host in the middle of every forward pass to size the binning buffer
    (rasterizer_impl.cu:282) -- the only stream synchronisation of the iteration.  With an ``InstanceCapacity`` handed to
    ``rasterize_gaussians_batched`` the binning buffer is sized for ``capacity`` instances instead, the kernels read the live
    count on the device, and the count comes back through a DEFERRED asynchronous copy into pinned memory that the NEXT call
    looks at (by then it landed long ago): it re-sizes the capacity (``margin`` x the LARGEST count of the last ``window`` calls,
    rounded up to ``quantum`` -- a training loop draws a new random camera batch every iteration, so the count moves from call
    to call) and RAISES if the previous call overflowed -- that call binned nothing (every view shows the background), so its results,
    and whatever an optimizer did with them, are void; a loop that cannot tolerate that uses a larger margin or
    ``reset()`` (the next call then takes the synchronising path once and re-seeds the capacity) whenever the scene changes
    abruptly (densification).  The first call, or any call after ``reset()``, synchronises like the reference."""

    def __init__(self, margin: float = 1.5, quantum: int = 1 << 16, window: int = 32):
        self.margin, self.quantum, self.window = float(margin), int(quantum), int(window)
        self._recent = []            # counts of the last `window` calls
        self.value = None            # capacity of the next call; None -> synchronising path
        self.last_count = None       # num_rendered of the most recent call whose count has been read
        self.calls_sync_free = 0
        self.overflows = 0           # sync-free calls that overflowed and were noticed through overflowed() / collect()
        self.radix = False           # a tile list outgrew the tile-bucketed binning: sync-free calls use the radix sort
        self.last_capacity = None
        self._dev = self._host = self._event = None
        self._pending = False

    @property
    def pending(self) -> bool:
        """The most recent call ran sync-free and its count has not been looked at yet."""
        return self._pending

    def reset(self):
        self.collect()
        self.value = None
        self._recent = []
        self.radix = False

    def _round(self, n: int) -> int:
        self._recent = (self._recent + [int(n)])[-self.window:]
        q = self.quantum
        return max(q, (int(max(self._recent) * self.margin) + q - 1) // q * q)

    def seed(self, num_rendered: int):
        self.last_count = int(num_rendered)
        self.value = self._round(num_rendered)

    def buffers(self, device):
        if self._dev is None or self._dev.device != device:
            self._dev = torch.zeros(4, dtype=torch.int32, device=device)
            self._host = torch.zeros(4, dtype=torch.int32).pin_memory()
            self._event = torch.cuda.Event()
        return self._dev

    def observe(self, device):
        """Right after a sync-free call: start the count's trip to the host (no wait)."""
        self._host.copy_(self._dev, non_blocking=True)
        self._event.record(torch.cuda.current_stream(device))
        self._pending = True
        self.calls_sync_free += 1

    def overflowed(self) -> bool:
        """Look at the most recent sync-free call's count NOW (a wait only if the GPU has not reached that copy yet) and say
        whether that call overflowed -- without raising.  A training loop calls this BEFORE it lets the call's results reach
        an optimizer (SDSLoop.step: after the guidance forward, when the copy landed long ago) and, on True, repeats the render:
        the capacity has been dropped, so the repeat takes the synchronising path and re-seeds it."""
        if not self._pending:
            return False
        self._event.synchronize()
        self._pending = False
        total, _live, over, cap = (int(v) & 0xffffffff for v in self._host.tolist())
        self.last_count, self.last_capacity = total, cap
        if over:
            # 1: more instances than the capacity; 2: a tile list longer than the tile-bucketed binning sorts in LDS (4096) --
            # from now on (until reset()) the sync-free calls ask for the radix-sort binning (a negative capacity, gd_raster.h)
            if over == 2:
                self.radix = True
            self.value = None
            self._recent = []
            self.overflows += 1
            return True
        self.value = self._round(total)
        return False

    def collect(self):
        """Look at the previous sync-free call's count (a wait only if the GPU has not reached that copy yet); RAISES if that
        call overflowed and nobody asked ``overflowed()`` in between (its results were used unchecked)."""
        if self.overflowed():
            raise RuntimeError(f"rasterizer: the previous sync-free forward pass overflowed its instance capacity "
                               f"({self.last_count} instances > capacity {self.last_capacity}): it rendered nothing and its "
                               f"results are void")


def rasterize_gaussians_batched(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                                viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree,
                                campos, prefiltered, debug, capacity: "InstanceCapacity | None" = None):
    """V views in one launch set.  viewmatrix/projmatrix [V,4,4], campos [V,3], tan_fov*: sequences
    of V floats.  Returns ``(num_rendered, color[V,3,H,W], depth[V,1,H,W], alpha[V,1,H,W],
    radii[V,P], geom, binning, img)``.  ``capacity``: an ``InstanceCapacity`` -- the call then runs WITHOUT the host read-back
    of the instance count whenever the object holds a capacity, and the first returned value is that capacity (what the
    backward call wants as R), not the instance count."""
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    _require_gpu(means3D)
    dev = means3D.device
    L = _native.lib()
    V = int(viewmatrix.shape[0])
    if not (1 <= V <= _native.GD_MAX_VIEWS) or len(tan_fovx) != V or len(tan_fovy) != V:
        raise RuntimeError(f"batched rasterizer needs 1..{_native.GD_MAX_VIEWS} views with matching tan_fov lists")
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    out_color = _empty((V, NUM_CHANNELS, H, W), torch.float32, dev)
    out_depth = _empty((V, 1, H, W), torch.float32, dev)
    out_alpha = _empty((V, 1, H, W), torch.float32, dev)
    radii = _empty((V, P), torch.int32, dev)
    geom, binning, img = _Scratch(dev), _Scratch(dev), _Scratch(dev)
    M = sh.size(1) if (sh is not None and sh.numel() != 0) else 0
    keep = [_f32c(t, dev) for t in (background, means3D, sh, colors, opacity, scales, rotations, cov3D_precomp,
                                    viewmatrix, projmatrix, campos)]
    bg, m3, shc, col, opa, scl, rot, cov, vm, pm, cp = keep
    tx, ty = _farr(tan_fovx), _farr(tan_fovy)
    if capacity is not None:
        capacity.collect()           # the previous call's count: re-sizes the capacity, raises after an overflow
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        head = (stream, V, geom.cb, None, binning.cb, None, img.cb, None, P, int(degree), M, _ptr(bg), W, H, _ptr(m3),
                _ptr(shc), _ptr(col), _ptr(opa), _ptr(scl), float(scale_modifier), _ptr(rot), _ptr(cov), _ptr(vm),
                _ptr(pm), _ptr(cp), tx, ty, int(bool(prefiltered)), out_color.data_ptr(), out_depth.data_ptr(),
                out_alpha.data_ptr(), radii.data_ptr() if P else None, int(bool(debug)))
        if capacity is not None and capacity.value is not None and P > 0:
            rendered = L.gd_raster_forward_batched_capacity(*head, -int(capacity.value) if capacity.radix else int(capacity.value),
                                                            capacity.buffers(dev).data_ptr())
            _native.check(rendered, "gd_raster_forward_batched_capacity")
            capacity.observe(dev)
        else:
            rendered = L.gd_raster_forward_batched(*head)
            _native.check(rendered, "gd_raster_forward_batched")
            if capacity is not None:
                capacity.seed(rendered)
    return rendered, out_color, out_depth, out_alpha, radii, geom.tensor, binning.tensor, img.tensor


def rasterize_gaussians_backward_batched(background, means3D, radii, colors, scales, rotations, scale_modifier,
                                         cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                         dL_dout_depth, dL_dout_alpha, sh, degree, campos, geomBuffer, R,
                                         binningBuffer, imageBuffer, alphas, debug):
    """Returns ``(dL_dmeans2D[V,P,3], dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh,
    dL_dscales, dL_drotations)`` -- all but the first summed over views."""
    _require_gpu(means3D)
    dev = means3D.device
    L = _native.lib()
    P = means3D.size(0)
    V, H, W = dL_dout_color.size(0), dL_dout_color.size(2), dL_dout_color.size(3)
    M = sh.size(1) if (sh is not None and sh.numel() != 0) else 0
    mk = lambda *shape: _empty(shape, torch.float32, dev)
    dL_dmeans3D, dL_dmeans2D, dL_dcolors = mk(P, 3), mk(V, P, 3), mk(P, NUM_CHANNELS)
    dL_dopacity, dL_dcov3D = mk(P, 1), mk(P, 6)
    dL_dsh, dL_dscales, dL_drotations = mk(P, M, 3), mk(P, 3), mk(P, 4)
    if P == 0:
        return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations
    if scales is None or scales.numel() == 0:
        dL_dscales.zero_()
        dL_drotations.zero_()
    scratch = _empty(L.gd_raster_backward_scratch_bytes(P, V, int(R)), torch.uint8, dev)
    keep = [_f32c(t, dev) for t in (background, means3D, sh, colors, alphas, scales, rotations, cov3D_precomp,
                                    viewmatrix, projmatrix, campos, dL_dout_color, dL_dout_depth, dL_dout_alpha)]
    bg, m3, shc, col, alp, scl, rot, cov, vm, pm, cp, gcol, gdep, galp = keep
    radii_c = radii.contiguous()
    tx, ty = _farr(tan_fovx), _farr(tan_fovy)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        ret = L.gd_raster_backward_batched(
            stream, V, P, int(degree), M, int(R), _ptr(bg), W, H, _ptr(m3), _ptr(shc), _ptr(col), _ptr(alp),
            _ptr(scl), float(scale_modifier), _ptr(rot), _ptr(cov), _ptr(vm), _ptr(pm), _ptr(cp), tx, ty,
            radii_c.data_ptr(), geomBuffer.data_ptr(), _ptr(binningBuffer) or geomBuffer.data_ptr(),
            imageBuffer.data_ptr(), scratch.data_ptr(), _ptr(gcol), _ptr(gdep), _ptr(galp), dL_dmeans2D.data_ptr(),
            dL_dopacity.data_ptr(), dL_dcolors.data_ptr(), dL_dmeans3D.data_ptr(), dL_dcov3D.data_ptr(),
            _ptr(dL_dsh), dL_dscales.data_ptr(), dL_drotations.data_ptr(), int(bool(debug)))
    _native.check(ret, "gd_raster_backward_batched")
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations
