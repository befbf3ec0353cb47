"""Shared builders for the parity tests: seeded scenes + cameras as plain numpy, the oracle call,
and readers for the HIP library's scratch buffers."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from garmentdreamer_amd import cameras as gcam
from garmentdreamer_amd.scene import synthetic_gaussians
from oracle import gd_oracle


def make_camera(azimuth=30.0, elevation=15.0, distance=2.75, fovy_deg=55.0, H=64, W=64):
    c2w = gcam.c2w_3dgs(azimuth, elevation, distance)
    return gcam.Camera(c2w, math.radians(fovy_deg), H, W, data_device="cpu")


def raster_inputs(P=500, H=64, W=64, seed=0, sh_degree=0, azimuth=30.0, elevation=15.0, distance=2.75,
                  fovy_deg=55.0, scale_mul=1.0, bg=(1.0, 1.0, 1.0)):
    sc = synthetic_gaussians(P, seed=seed, sh_degree=sh_degree)
    cam = make_camera(azimuth, elevation, distance, fovy_deg, H, W)
    return dict(
        bg=np.asarray(bg, np.float32), means3D=sc["means3D"], colors_precomp=None, opacities=sc["opacities"],
        scales=(sc["scales"] * scale_mul).astype(np.float32), rotations=sc["rotations"], scale_modifier=1.0,
        cov3D_precomp=None, viewmatrix=cam.world_view_transform.numpy().copy(),
        projmatrix=cam.full_proj_transform.numpy().copy(), tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
        image_height=H, image_width=W, sh=sc["shs"], degree=sh_degree, campos=cam.camera_center.numpy().copy())


def oracle_forward(inp):
    return gd_oracle.forward(inp["bg"], inp["means3D"], inp["colors_precomp"], inp["opacities"], inp["scales"],
                             inp["rotations"], inp["scale_modifier"], inp["cov3D_precomp"], inp["viewmatrix"],
                             inp["projmatrix"], inp["tanfovx"], inp["tanfovy"], inp["image_height"],
                             inp["image_width"], inp["sh"], inp["degree"], inp["campos"])


def to_torch(inp, device):
    """numpy dict -> positional args of _C.rasterize_gaussians (19 of them)."""
    t = lambda a: torch.Tensor([]) if a is None else torch.as_tensor(np.ascontiguousarray(a), device=device)
    return (t(inp["bg"]), t(inp["means3D"]), t(inp["colors_precomp"]), t(inp["opacities"]), t(inp["scales"]),
            t(inp["rotations"]), float(inp["scale_modifier"]), t(inp["cov3D_precomp"]), t(inp["viewmatrix"]),
            t(inp["projmatrix"]), float(inp["tanfovx"]), float(inp["tanfovy"]), int(inp["image_height"]),
            int(inp["image_width"]), t(inp["sh"]), int(inp["degree"]), t(inp["campos"]), False, False)


def read_scratch(geom, binning, img, P, V, W, H, R):
    """Decode the library's three byte buffers (torch uint8, device) into named numpy arrays."""
    from garmentdreamer_amd import _native
    L = _native.lib()
    lay = _native.Layout()
    L.gd_raster_get_layout(geom.data_ptr(), img.data_ptr(), binning.data_ptr(), P, V, W, H, R, C.byref(lay))
    g, b, im = geom.cpu().numpy(), binning.cpu().numpy(), img.cpu().numpy()
    VP = V * P
    tiles = V * ((W + 15) // 16) * ((H + 15) // 16)

    def view(buf, off, count, dtype):
        n = count * np.dtype(dtype).itemsize
        return np.frombuffer(buf[off:off + n].tobytes(), dtype=dtype)

    rgbd = view(g, lay.rgb, VP * 4, np.float32).reshape(VP, 4)
    out = dict(
        rgb=rgbd[:, :3].copy(), depths=rgbd[:, 3].copy(),
        clamped=view(g, lay.clamped, VP * 3, np.uint8).reshape(VP, 3),
        means2D=view(g, lay.means2D, VP * 2, np.float32).reshape(VP, 2),
        cov3D=view(g, lay.cov3D, VP * 6, np.float32).reshape(VP, 6),
        conic_opacity=view(g, lay.conic_opacity, VP * 4, np.float32).reshape(VP, 4),
        tiles_touched=view(g, lay.tiles_touched, VP, np.uint32),
        point_offsets=view(g, lay.point_offsets, VP, np.uint32),
        ranges=view(im, lay.ranges, tiles * 2, np.uint32).reshape(tiles, 2),
        n_contrib=view(im, lay.n_contrib, V * H * W, np.uint32).reshape(V, H, W),
        pair_counts=view(im, lay.pair_counts, V * H * W * 2, np.uint32).reshape(V, H, W, 2),
        point_list=view(b, lay.point_list, R, np.uint32),
        keys=view(b, lay.keys, R, np.uint64),
    )
    return out


def random_image_grads(H, W, seed=1):
    rng = np.random.default_rng(seed)
    return (rng.normal(size=(3, H, W)).astype(np.float32), rng.normal(size=(1, H, W)).astype(np.float32),
            rng.normal(size=(1, H, W)).astype(np.float32))


def poison_lds():
    """NaN bit patterns into every CU's LDS (gd_raster_poison_lds): a kernel that reads a shared-memory cell it never
    wrote then fails its parity test instead of passing on whatever the previous workgroup left there."""
    from garmentdreamer_amd import _native
    L = _native.lib()
    with torch.cuda.device(0):
        assert L.gd_raster_poison_lds(torch.cuda.current_stream().cuda_stream) == 0


def needle_inputs(P, HW, seed):
    """Strongly anisotropic splats (axis ratio up to ~60:1, random orientation) with opacities down to the
    1/255 threshold: the hard case for the render kernels' per-strip reachability test (a needle that
    crosses a 16x4 strip diagonally, or clips its corner, must not be dropped)."""
    inp = raster_inputs(P=P, H=HW, W=HW, seed=seed, scale_mul=1.0)
    rng = np.random.default_rng(seed)
    sc = inp["scales"].copy()
    sc[:, 0] *= rng.uniform(5.0, 30.0, size=P).astype(np.float32)
    sc[:, 1] *= rng.uniform(0.5, 2.0, size=P).astype(np.float32)
    sc[:, 2] *= rng.uniform(0.3, 1.0, size=P).astype(np.float32)
    inp["scales"] = sc
    op = inp["opacities"].copy()
    op[: P // 4] = rng.uniform(0.002, 0.02, size=(P // 4, 1)).astype(np.float32)   # around and below 1/255
    inp["opacities"] = op
    return inp
