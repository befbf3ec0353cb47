"""Checks against golden vectors captured from the reference's importable Python
(tests/golden/make_golden.py; the reference itself is not needed at test time)."""
import json
import os

import numpy as np
import torch

from garmentdreamer_amd import cameras as gcam
from tests import helpers as h

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_camera_matrices_match_reference_camera_bit_for_bit():
    d = np.load(os.path.join(G, "cameras.npz"))
    for i, (az, el, dist, fovy, H, W, fovx) in enumerate(d["params"]):
        c2w = gcam.c2w_3dgs(float(az), float(el), float(dist))
        np.testing.assert_array_equal(c2w.numpy(), d["c2w"][i])
        cam = gcam.Camera(torch.tensor(d["c2w"][i]), float(fovy), int(H), int(W), data_device="cpu")
        np.testing.assert_array_equal(cam.world_view_transform.numpy(), d["wvt"][i])
        np.testing.assert_array_equal(cam.full_proj_transform.numpy(), d["full"][i])
        np.testing.assert_array_equal(cam.camera_center.numpy(), d["center"][i])
        assert cam.FoVx == fovx
        # the row-vector convention the kernels index: last column of the view matrix is (0,0,0,1)
        np.testing.assert_allclose(cam.world_view_transform.numpy()[:, 3], [0, 0, 0, 1], atol=1e-6)


def test_camera_batch_packs_same_matrices():
    batch = gcam.orbit_batch(4, height=64, width=64)
    cams = [gcam.Camera(batch["c2w_3dgs"][i], batch["fovy"][i], 64, 64, data_device="cpu") for i in range(4)]
    cb = gcam.CameraBatch(cams, "cpu")
    for i, c in enumerate(cams):
        np.testing.assert_array_equal(cb.viewmatrix[i].numpy(), c.world_view_transform.numpy())
        np.testing.assert_array_equal(cb.projmatrix[i].numpy(), c.full_proj_transform.numpy())
        np.testing.assert_array_equal(cb.campos[i].numpy(), c.camera_center.numpy())
    assert cb.tanfovx == [c.tanfovx for c in cams]


def test_oracle_sh_colours_match_reference_eval_sh():
    """Oracle's computeColorFromSH restatement vs the reference's eval_sh (+0.5, clamp) for degrees 0-3."""
    d = np.load(os.path.join(G, "sh_eval.npz"))
    dirs = d["dirs"]
    P = dirs.shape[0]
    for deg in range(4):
        # put Gaussian g at position dirs[g]*2 with the camera at the origin -> direction = dirs[g]
        cam = h.make_camera(H=32, W=32)
        inp = h.raster_inputs(P=P, H=32, W=32)
        inp["means3D"] = (dirs * 2.0).astype(np.float32)
        inp["sh"] = d[f"sh{deg}"]
        inp["degree"] = deg
        inp["campos"] = np.zeros(3, np.float32)
        # make every Gaussian pass culling irrespective of the camera: use an orthonormal view that
        # looks down +z from far behind, huge image so rects are non-empty
        view = np.eye(4, dtype=np.float32)
        view[3, 2] = 10.0  # row-vector convention: translation in the last ROW
        inp["viewmatrix"] = view
        proj = cam.projection_matrix.numpy()
        inp["projmatrix"] = (view @ proj).astype(np.float32)
        st = h.oracle_forward(inp)
        vis = st.radii > 0
        assert vis.sum() > P // 2
        np.testing.assert_allclose(st.rgb[vis], d[f"rgb{deg}"][vis], rtol=2e-5, atol=2e-6)
        np.testing.assert_array_equal(st.clamped[vis].astype(bool), (d[f"raw{deg}"] < 0)[vis])


def test_python_op_marshals_like_the_reference(monkeypatch):
    """Our autograd.Function hands the native layer the same 19 / 24 positional arguments (kinds,
    shapes, order) and maps the native gradient tuple to the same inputs as the reference's op."""
    with open(os.path.join(G, "marshalling.json")) as f:
        gold = json.load(f)
    import garmentdreamer_amd.diff_gaussian_rasterization as dgr
    rec = {}
    P, H, W, M = 5, 8, 8, 1

    def kind(a):
        if isinstance(a, torch.Tensor):
            return ["tensor", list(a.shape), str(a.dtype).replace("torch.", "")]
        return [type(a).__name__, a if isinstance(a, (int, float, bool)) else None]

    def fwd(*args):
        rec["forward_args"] = [kind(a) for a in args]
        return (7, torch.zeros(3, H, W), torch.zeros(1, H, W), torch.zeros(1, H, W), torch.zeros(P, dtype=torch.int32),
                torch.zeros(3, dtype=torch.uint8), torch.zeros(4, dtype=torch.uint8), torch.zeros(5, dtype=torch.uint8))

    def bwd(*args):
        rec["backward_args"] = [kind(a) for a in args]
        shapes = [(P, 3), (P, 3), (P, 1), (P, 3), (P, 6), (P, M, 3), (P, 3), (P, 4)]
        return tuple(torch.full(s, float(i + 1)) for i, s in enumerate(shapes))

    monkeypatch.setattr(dgr._C, "rasterize_gaussians", fwd)
    monkeypatch.setattr(dgr._C, "rasterize_gaussians_backward", bwd)
    rs = dgr.GaussianRasterizationSettings(H, W, 0.5, 0.6, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0,
                                           torch.zeros(3), False, False)
    leaves = dict(means3D=torch.zeros(P, 3), means2D=torch.zeros(P, 3), opacities=torch.zeros(P, 1),
                  shs=torch.zeros(P, M, 3), scales=torch.zeros(P, 3), rotations=torch.zeros(P, 4))
    for v in leaves.values():
        v.requires_grad_(True)
    color, radii, depth, alpha = dgr.GaussianRasterizer(rs)(**leaves)
    (color.sum() + depth.sum() + alpha.sum()).backward()
    assert list(dgr.GaussianRasterizationSettings._fields) == gold["settings_fields"]
    assert rec["forward_args"] == gold["forward_args"]
    assert rec["backward_args"] == gold["backward_args"]
    got = {n: float(leaves[n].grad.flatten()[0]) for n in leaves}
    assert got == gold["grad_sentinel_by_input"]
    assert color.shape == (3, H, W) and radii.dtype == torch.int32 and depth.shape == (1, H, W)
