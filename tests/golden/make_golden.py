"""Generate the committed golden vectors by IMPORTING the reference's pure-Python helpers.

Run in the build container only (``/root/reference`` does not exist on the GPU box; tests read
just the ``.npz`` / ``.json`` this script writes):

    python tests/golden/make_golden.py

What is captured:
  cameras.npz     (c2w, FoVy, H, W) -> world_view_transform, full_proj_transform, camera_center,
                  FoVx from ``gaussiansplatting/scene/cameras.py::Camera`` (CUDA moves patched to
                  identity), plus ``pose_spherical``-derived c2w_3dgs matrices restated from
                  ``threestudio/data/uncond.py:371-390`` using the reference's own
                  ``getWorld2View2_tensor`` / ``getProjectionMatrix``.
  sh_eval.npz     ``eval_sh(deg, sh, dirs)`` for deg 0..3 from ``gaussiansplatting/utils/sh_utils.py``
                  (+0.5, clamp) -- pins computeColorFromSH's constants and basis order.
  marshalling.json  positional-argument order / kinds the reference's Python op hands to its native
                  module (forward: 19 args, backward: 24 args) and the native-grad-tuple -> input map,
                  recorded with a stub ``_C`` pre-seeded in ``sys.modules``.
No reference source text is stored -- only inputs and outputs.
"""
import json
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/Garment_3DGS"
DGR = REF + "/gaussiansplatting/submodules/diff-gaussian-rasterization"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def make_cameras():
    sys.path.insert(0, REF)
    _stub("plyfile", PlyData=object, PlyElement=object)
    _stub("simple_knn")
    _stub("simple_knn._C", distCUDA2=None)
    torch.Tensor.cuda = lambda self, *a, **k: self  # no GPU here: keep everything on the host
    from gaussiansplatting.utils import graphics_utils as gu  # noqa
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_cameras", REF + "/gaussiansplatting/scene/cameras.py")
    cams = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cams)

    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from garmentdreamer_amd import cameras as mine  # pose inputs come from our restated c2w_3dgs

    rows = []
    grid = [(az, el, dist, fovy, H, W)
            for az in (-170.0, -45.0, 0.0, 30.0, 120.0)
            for (el, dist, fovy, H, W) in ((15.0, 2.75, 55.0, 512, 512), (-20.0, 1.5, 40.0, 256, 256),
                                           (65.0, 4.0, 70.0, 1024, 1024), (0.0, 3.0, 50.0, 75, 117))]
    for (az, el, dist, fovy, H, W) in grid:
        c2w = mine.c2w_3dgs(az, el, dist)
        cam = cams.Camera(c2w=c2w, FoVy=math.radians(fovy), height=H, width=W, data_device="cpu")
        rows.append(dict(az=az, el=el, dist=dist, fovy=math.radians(fovy), H=H, W=W, c2w=c2w.numpy(),
                         wvt=cam.world_view_transform.numpy(), full=cam.full_proj_transform.numpy(),
                         center=cam.camera_center.numpy(), FoVx=cam.FoVx))
    np.savez(os.path.join(OUT, "cameras.npz"),
             params=np.array([[r["az"], r["el"], r["dist"], r["fovy"], r["H"], r["W"], r["FoVx"]] for r in rows]),
             c2w=np.stack([r["c2w"] for r in rows]), wvt=np.stack([r["wvt"] for r in rows]),
             full=np.stack([r["full"] for r in rows]), center=np.stack([r["center"] for r in rows]))
    print("cameras.npz:", len(rows), "cameras")


def make_sh():
    sys.path.insert(0, REF)
    from gaussiansplatting.utils.sh_utils import eval_sh
    rng = np.random.default_rng(7)
    P = 64
    dirs = rng.normal(size=(P, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    out = {"dirs": dirs.astype(np.float32)}
    for deg in range(4):
        M = (deg + 1) ** 2
        sh = rng.normal(scale=0.5, size=(P, M, 3)).astype(np.float32)
        # reference layout for eval_sh: [..., C, (deg+1)^2]
        res = eval_sh(deg, torch.tensor(sh).transpose(1, 2), torch.tensor(out["dirs"]))
        out[f"sh{deg}"] = sh
        out[f"rgb{deg}"] = torch.clamp_min(res + 0.5, 0.0).numpy()
        out[f"raw{deg}"] = (res + 0.5).numpy()
    np.savez(os.path.join(OUT, "sh_eval.npz"), **out)
    print("sh_eval.npz written")


def make_marshalling():
    rec = {}

    def kind(a):
        if isinstance(a, torch.Tensor):
            return ["tensor", list(a.shape), str(a.dtype).replace("torch.", "")]
        return [type(a).__name__, a if isinstance(a, (int, float, bool)) else None]

    P, H, W, M = 5, 8, 8, 1

    def fwd(*args):
        rec["forward_args"] = [kind(a) for a in args]
        return (7, torch.zeros(3, H, W), torch.zeros(1, H, W), torch.zeros(1, H, W), torch.zeros(P, dtype=torch.int32),
                torch.zeros(3, dtype=torch.uint8), torch.zeros(4, dtype=torch.uint8), torch.zeros(5, dtype=torch.uint8))

    def bwd(*args):
        rec["backward_args"] = [kind(a) for a in args]
        # sentinel-filled grads: value = native tuple index + 1
        shapes = [(P, 3), (P, 3), (P, 1), (P, 3), (P, 6), (P, M, 3), (P, 3), (P, 4)]
        return tuple(torch.full(s, float(i + 1)) for i, s in enumerate(shapes))

    pkg = _stub("diff_gaussian_rasterization")
    pkg.__path__ = [DGR + "/diff_gaussian_rasterization"]
    _stub("diff_gaussian_rasterization._C", rasterize_gaussians=fwd, rasterize_gaussians_backward=bwd,
          mark_visible=lambda *a: torch.ones(P, dtype=torch.bool))
    import importlib.util
    spec = importlib.util.spec_from_file_location("diff_gaussian_rasterization",
                                                  DGR + "/diff_gaussian_rasterization/__init__.py",
                                                  submodule_search_locations=[DGR + "/diff_gaussian_rasterization"])
    ref = importlib.util.module_from_spec(spec)
    sys.modules["diff_gaussian_rasterization"] = ref
    spec.loader.exec_module(ref)

    rs = ref.GaussianRasterizationSettings(H, W, 0.5, 0.6, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0,
                                           torch.zeros(3), False, False)
    names = ["means3D", "means2D", "opacities", "shs", "scales", "rotations"]
    leaves = dict(means3D=torch.zeros(P, 3), means2D=torch.zeros(P, 3), opacities=torch.zeros(P, 1),
                  shs=torch.zeros(P, M, 3), scales=torch.zeros(P, 3), rotations=torch.zeros(P, 4))
    for v in leaves.values():
        v.requires_grad_(True)
    color, radii, depth, alpha = ref.GaussianRasterizer(rs)(**leaves)
    (color.sum() + depth.sum() + alpha.sum()).backward()
    rec["settings_fields"] = list(ref.GaussianRasterizationSettings._fields)
    rec["forward_return"] = ["color", "radii", "depth", "alpha"]
    rec["grad_sentinel_by_input"] = {n: float(leaves[n].grad.flatten()[0]) for n in names}
    with open(os.path.join(OUT, "marshalling.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print("marshalling.json:", len(rec["forward_args"]), "fwd args,", len(rec["backward_args"]), "bwd args,",
          rec["grad_sentinel_by_input"])


if __name__ == "__main__":
    make_cameras()
    make_sh()
    make_marshalling()
