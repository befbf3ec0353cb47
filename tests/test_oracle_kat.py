"""Known-answer tests that pin the CPU oracle (oracle/gd_oracle.c).

The reference ships no tests or golden vectors for the rasterizer and its CUDA op cannot be
built here, so the oracle is pinned by closed-form cases, by an independent fp64
autograd statement of the math (tests/dense_reference.py) and by golden camera/SH vectors
generated from the reference's importable Python (tests/test_golden_fixtures.py).
"""
import math

import numpy as np
import pytest
import torch

from oracle import gd_oracle
from tests import dense_reference as dr
from tests import helpers as h


def _single(P_xyz, scales, opac, H=32, W=32, rot=None, shs=None, distance=3.0, fovy=50.0, bg=(0.0, 0.0, 0.0)):
    cam = h.make_camera(azimuth=0.0, elevation=0.0, distance=distance, fovy_deg=fovy, H=H, W=W)
    P = len(P_xyz)
    rot = np.tile(np.array([[1, 0, 0, 0]], np.float32), (P, 1)) if rot is None else rot
    shs = np.zeros((P, 1, 3), np.float32) if shs is None else shs
    inp = dict(bg=np.asarray(bg, np.float32), means3D=np.asarray(P_xyz, np.float32), colors_precomp=None,
               opacities=np.asarray(opac, np.float32).reshape(P, 1), scales=np.asarray(scales, np.float32),
               rotations=rot.astype(np.float32), scale_modifier=1.0, cov3D_precomp=None,
               viewmatrix=cam.world_view_transform.numpy().copy(), projmatrix=cam.full_proj_transform.numpy().copy(),
               tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, image_height=H, image_width=W, sh=shs, degree=0,
               campos=cam.camera_center.numpy().copy())
    return inp, cam


def test_kat1_single_isotropic_gaussian_closed_form():
    """One isotropic Gaussian at the origin seen from distance d: sigma^2 = (f s / d)^2 + 0.3,
    radius = ceil(3 sqrt(sigma^2 + sqrt(0.1))), alpha(pixel) = o exp(-r^2 / (2 sigma^2))."""
    H = W = 32
    s, o, d = 0.1, 0.8, 3.0
    inp, cam = _single([[0, 0, 0]], [[s, s, s]], [o], H, W, distance=d,
                       shs=np.full((1, 1, 3), (0.7 - 0.5) / 0.28209479177387814, np.float32))
    st = h.oracle_forward(inp)
    f = W / (2 * cam.tanfovx)
    var = (f * s / d) ** 2 + 0.3
    # isotropic: mid^2 - det = 0, so the eigenvalue clamp max(0.1, .) adds sqrt(0.1) (forward.cu:230-232)
    assert st.radii[0] == math.ceil(3 * math.sqrt(var + math.sqrt(0.1)))
    cx, cy = (W - 1) / 2, (H - 1) / 2
    np.testing.assert_allclose(st.means2D[0], [cx, cy], atol=1e-4)
    np.testing.assert_allclose(st.depths[0], d, rtol=1e-6)
    ys, xs = np.mgrid[0:H, 0:W]
    r2 = (xs - cx) ** 2 + (ys - cy) ** 2
    a = np.minimum(0.99, o * np.exp(-0.5 * r2 / var))
    a[a < 1 / 255] = 0
    # the Gaussian's tile rectangle covers the 4 centre tiles = whole 32x32 image here
    np.testing.assert_allclose(st.alpha[0], a, atol=2e-5)
    np.testing.assert_allclose(st.color[0], 0.7 * a, atol=2e-5)  # bg = 0
    np.testing.assert_allclose(st.depth[0], d * a, atol=1e-4)
    assert st.tiles_touched[0] == 4 and st.num_rendered == 4
    assert (st.n_contrib[a > 0] == 1).all() and (st.n_contrib[a == 0] == 0).all()


def test_kat2_depth_order_flips_with_swapped_depth():
    H = W = 16
    for near_first in (True, False):
        z = [0.5, -0.5] if near_first else [-0.5, 0.5]  # camera sits on +... axis; swap who is nearer
        inp, cam = _single([[0, 0, 0], [0, 0, 0]], [[0.2] * 3, [0.2] * 3], [0.6, 0.6], H, W)
        # move along the viewing axis (camera centre -> origin)
        axis = inp["campos"] / np.linalg.norm(inp["campos"])
        inp["means3D"] = np.stack([axis * z[0], axis * z[1]]).astype(np.float32)
        inp["sh"] = np.array([[[1.0, 0, 0]], [[0, 0, 1.0]]], np.float32) / 0.28209479177387814
        st = h.oracle_forward(inp)
        nearer = int(np.argmin(st.depths))
        assert list(st.point_list) == [nearer, 1 - nearer]
        assert st.n_contrib[8, 8] == 2
        # front Gaussian dominates: its channel is larger at the centre
        c = st.color[:, 8, 8]
        front_ch = 0 if nearer == 0 else 2
        assert c[front_ch] > c[2 - front_ch]


def test_kat3_saturation_stops_blending():
    """A stack of opaque Gaussians: blending stops once T(1-alpha) < 1e-4; alpha out = 1 - T."""
    H = W = 16
    n = 12
    inp, cam = _single([[0, 0, 0]] * n, [[0.5] * 3] * n, [0.99] * n, H, W)
    axis = inp["campos"] / np.linalg.norm(inp["campos"])
    inp["means3D"] = np.stack([axis * (0.05 * i) for i in range(n)]).astype(np.float32)
    st = h.oracle_forward(inp)
    # replay pixel (8,8) in float64 from the per-Gaussian conics (pinned by KAT1)
    T, k_expect, wsum = 1.0, 0, 0.0
    for j, g in enumerate(st.point_list[:n]):
        dx, dy = st.means2D[g, 0] - 8.0, st.means2D[g, 1] - 8.0
        ca, cb, cc, op = [float(v) for v in st.conic_opacity[g]]
        alpha = min(0.99, op * math.exp(-0.5 * (ca * dx * dx + cc * dy * dy) - cb * dx * dy))
        if T * (1 - alpha) < 1e-4:
            break
        wsum += alpha * T
        T *= 1 - alpha
        k_expect = j + 1
    assert 2 <= k_expect < n            # it does saturate well before the end of the stack
    assert st.n_contrib[8, 8] == k_expect
    np.testing.assert_allclose(st.alpha[0, 8, 8], wsum, rtol=1e-5)
    np.testing.assert_allclose(st.alpha[0, 8, 8], 1 - T, rtol=1e-5)   # alpha out == 1 - T_final
    assert st.pairs_visited_fwd < n * H * W  # early exit happened


def test_kat4_culling_cases():
    H = W = 32
    inp, cam = _single([[0, 0, 0], [0, 0, 0], [0, 0, 0]], [[0.05] * 3] * 3, [0.5] * 3, H, W)
    axis = inp["campos"] / np.linalg.norm(inp["campos"])
    cam_dist = np.linalg.norm(inp["campos"])
    behind = axis * (cam_dist + 1.0)             # behind the camera
    near = axis * (cam_dist - 0.1)               # z = 0.1 <= 0.2 -> near-culled
    # far off to the side: projects outside every tile
    side = np.cross(axis, [0, 0, 1.0])
    side = side / np.linalg.norm(side) * 50.0
    inp["means3D"] = np.stack([behind, near, side]).astype(np.float32)
    st = h.oracle_forward(inp)
    assert list(st.radii) == [0, 0, 0] and st.num_rendered == 0
    np.testing.assert_array_equal(st.color, np.zeros_like(st.color))
    vis = gd_oracle.mark_visible(inp["means3D"], inp["viewmatrix"], inp["projmatrix"])
    assert list(vis) == [False, False, True]     # markVisible only tests z > 0.2


def test_kat5_identical_depth_keeps_index_order():
    H = W = 32
    n = 6
    inp, cam = _single([[0, 0, 0]] * n, [[0.1] * 3] * n, [0.3] * n, H, W)
    st = h.oracle_forward(inp)
    assert len(set(st.depths.view(np.uint32))) == 1
    per_tile = st.point_list.reshape(-1, n)
    for row in per_tile:
        assert list(row) == list(range(n))
    assert (np.diff(st.keys.astype(np.int64)) >= 0).all()


def test_kat6_higher_msb_table():
    for n in (1, 2, 3, 255, 256, 257, 1024, 4096, 8192, 65535):
        assert gd_oracle.higher_msb(n) == n.bit_length(), n
    # sort width = 32 + msb(tiles): 256 tiles -> 41, 1024 -> 43, 4096 -> 45
    assert 32 + gd_oracle.higher_msb(256) == 41 and 32 + gd_oracle.higher_msb(1024) == 43
    assert 32 + gd_oracle.higher_msb(4096) == 45


def _dense_case(P=24, H=16, W=32, seed=3, scale_mul=6.0):
    inp = h.raster_inputs(P=P, H=H, W=W, seed=seed, scale_mul=scale_mul, bg=(0.2, 0.5, 0.9), fovy_deg=40.0)
    st = h.oracle_forward(inp)
    f64 = lambda a: torch.tensor(np.asarray(a, np.float64))
    leaves = dict(means3D=f64(inp["means3D"]).requires_grad_(True), scales=f64(inp["scales"]).requires_grad_(True),
                  rotations=f64(inp["rotations"]).requires_grad_(True),
                  opacities=f64(inp["opacities"]).requires_grad_(True),
                  shs=f64(inp["sh"][:, 0, :]).requires_grad_(True))
    mask = dr.tile_mask_from_oracle(st, H, W)
    out = dr.render_dense(leaves["means3D"], leaves["scales"], leaves["rotations"], leaves["opacities"],
                          leaves["shs"], f64(inp["viewmatrix"]), f64(inp["projmatrix"]), f64(inp["campos"]),
                          float(inp["tanfovx"]), float(inp["tanfovy"]), H, W, f64(inp["bg"]), tile_mask=mask)
    return inp, st, leaves, out


def test_kat7_forward_matches_independent_fp64_math():
    inp, st, leaves, (color, depth, alpha) = _dense_case()
    assert st.num_rendered > 0 and st.pairs_blended_fwd > 100
    np.testing.assert_allclose(color.detach().numpy(), st.color, atol=3e-5, rtol=1e-4)
    np.testing.assert_allclose(depth.detach().numpy(), st.depth, atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(alpha.detach().numpy(), st.alpha, atol=3e-5, rtol=1e-4)


def test_kat7_backward_matches_autograd_of_independent_math():
    """The oracle's analytic gradients (backward.cu restated) == fp64 autograd of the math, for
    every differentiable input and all three output heads (colour, depth, alpha)."""
    inp, st, leaves, (color, depth, alpha) = _dense_case()
    H, W = inp["image_height"], inp["image_width"]
    gc, gd, ga = h.random_image_grads(H, W, seed=5)
    loss = (color * torch.tensor(gc, dtype=torch.float64)).sum() + \
        (depth * torch.tensor(gd, dtype=torch.float64)).sum() + (alpha * torch.tensor(ga, dtype=torch.float64)).sum()
    loss.backward()
    ref = gd_oracle.backward(st, gc, gd, ga)

    def close(name, got, want, rtol=2e-3):
        want = want.numpy().reshape(got.shape)
        scale = np.abs(want).max() + 1e-30
        err = np.abs(got - want)
        assert (err <= rtol * np.abs(want) + 2e-4 * scale).all(), \
            f"{name}: max err {err.max():.3e} (scale {scale:.3e})"

    close("means3D", ref["dL_dmeans3D"], leaves["means3D"].grad)
    close("scales", ref["dL_dscales"], leaves["scales"].grad)
    close("rotations", ref["dL_drotations"], leaves["rotations"].grad)
    close("opacities", ref["dL_dopacity"], leaves["opacities"].grad)
    close("sh", ref["dL_dsh"][:, 0, :], leaves["shs"].grad)


def test_kat8_absent_optional_paths_agree_with_equivalent_inputs():
    """colors_precomp == SH-evaluated colours and cov3D_precomp == computed cov3D give the same
    image; the gradient w.r.t. the precomputed colour equals the oracle's dL_dcolors."""
    inp = h.raster_inputs(P=300, H=48, W=48, seed=8)
    st = h.oracle_forward(inp)
    inp2 = dict(inp, colors_precomp=st.rgb.copy(), sh=None, cov3D_precomp=st.cov3D.copy(), scales=None,
                rotations=None)
    st2 = h.oracle_forward(inp2)
    np.testing.assert_array_equal(st.color, st2.color)
    np.testing.assert_array_equal(st.point_list, st2.point_list)
    gc, gd, ga = h.random_image_grads(48, 48)
    g1 = gd_oracle.backward(st, gc, gd, ga)
    g2 = gd_oracle.backward(st2, gc, gd, ga)
    np.testing.assert_array_equal(g1["dL_dcolors"], g2["dL_dcolors"])
    np.testing.assert_array_equal(g1["dL_dcov3D"], g2["dL_dcov3D"])
    assert np.abs(g2["dL_dscales"]).sum() == 0 and g2["dL_dsh"].size == 0
    # SH path adds the view-direction term only for degree > 0: at degree 0 mean grads agree
    np.testing.assert_allclose(g1["dL_dmeans3D"], g2["dL_dmeans3D"], rtol=0, atol=0)


def test_sh_degree3_gradient_by_finite_differences():
    """Degree-3 SH colour + its view-direction gradient into means3D, against central differences
    of the oracle's own forward (colours_precomp path keeps the rest fixed)."""
    rng = np.random.default_rng(4)
    inp = h.raster_inputs(P=40, H=16, W=16, seed=12, sh_degree=3, scale_mul=5.0)
    inp["sh"] = rng.normal(scale=0.3, size=inp["sh"].shape).astype(np.float32)
    inp["sh"][:, 0, :] += 1.0
    st = h.oracle_forward(inp)
    assert not st.clamped.all() and st.rgb.max() > 0
    gc, gd, ga = h.random_image_grads(16, 16, seed=2)
    g = gd_oracle.backward(st, gc, gd * 0, ga * 0)
    # d(loss)/d(sh[k]) by finite differences
    def loss_of(sh):
        s2 = h.oracle_forward(dict(inp, sh=sh))
        return float((s2.color.astype(np.float64) * gc).sum())
    picks = [(3, 0, 1), (7, 5, 0), (11, 9, 2), (20, 15, 1), (33, 12, 2)]
    for (gi, k, c) in picks:
        if st.radii[gi] <= 0 or st.clamped[gi, c]:
            continue
        e = 1e-2
        shp, shm = inp["sh"].copy(), inp["sh"].copy()
        shp[gi, k, c] += e
        shm[gi, k, c] -= e
        fd = (loss_of(shp) - loss_of(shm)) / (2 * e)
        assert abs(fd - g["dL_dsh"][gi, k, c]) <= 2e-2 * abs(fd) + 2e-3, (gi, k, c, fd, g["dL_dsh"][gi, k, c])


def test_gradient_shapes_and_zero_rows_for_culled():
    inp = h.raster_inputs(P=200, H=32, W=32, seed=1, distance=0.9)
    st = h.oracle_forward(inp)
    culled = st.radii <= 0
    assert culled.any() and (~culled).any()
    g = gd_oracle.backward(st, *h.random_image_grads(32, 32))
    for name in ("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dopacity", "dL_dmeans2D"):
        assert np.abs(g[name][culled]).sum() == 0, name
    assert g["dL_dmeans2D"].shape == (200, 3) and np.abs(g["dL_dmeans2D"][:, 2]).sum() == 0


def test_openmp_build_of_the_oracle_matches_the_serial_build():
    """bench.py's multi-core CPU baseline is the same C file built with -fopenmp (tiles over cores): forward outputs
    bit-identical, backward sums equal up to the fp64 atomic order."""
    inp = h.raster_inputs(P=3000, H=96, W=112, seed=5)
    a = h.oracle_forward(inp)
    b = gd_oracle.forward(inp["bg"], inp["means3D"], inp["colors_precomp"], inp["opacities"], inp["scales"],
                          inp["rotations"], inp["scale_modifier"], inp["cov3D_precomp"], inp["viewmatrix"],
                          inp["projmatrix"], inp["tanfovx"], inp["tanfovy"], inp["image_height"], inp["image_width"],
                          inp["sh"], inp["degree"], inp["campos"], omp=True)
    for k in ("color", "depth", "alpha", "n_contrib", "point_list", "ranges", "radii"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k
    assert a.pairs_visited_fwd == b.pairs_visited_fwd and a.pairs_blended_fwd == b.pairs_blended_fwd
    gc, gd, ga = h.random_image_grads(96, 112)
    ra, rb = gd_oracle.backward(a, gc, gd, ga), gd_oracle.backward(b, gc, gd, ga)
    assert a.pairs_visited_bwd == b.pairs_visited_bwd
    for k in ra:
        np.testing.assert_allclose(rb[k], ra[k], rtol=1e-5, atol=1e-7 * (np.abs(ra[k]).max() + 1e-30))


def test_blend_exp_is_within_one_ulp_of_the_exact_exponential():
    """gd_expf (the exp both the oracle and the HIP kernels use in the alpha blend) against float64 exp on the range the
    blend feeds it (power <= 0; contributions need power >= ln(1/255) ~ -5.5) and beyond."""
    rng = np.random.default_rng(0)
    xs = np.concatenate([-rng.uniform(0, 8, 20000), -rng.uniform(8, 87, 2000), [-0.0, -1e-8, -86.9, -5.541263545158426]])
    xs = xs.astype(np.float32)
    got = np.array([gd_oracle.expf(float(x)) for x in xs], np.float32)
    ref = np.exp(xs.astype(np.float64))
    ulp = np.spacing(ref.astype(np.float32)).astype(np.float64)
    assert np.abs(got.astype(np.float64) - ref).max() / 1.0 >= 0.0
    assert (np.abs(got.astype(np.float64) - ref) <= 1.0 * ulp).all()
    assert gd_oracle.expf(-100.0) == 0.0 and gd_oracle.expf(0.0) == 1.0


def test_defined_blend_exp_agrees_with_the_c_library_exp_end_to_end():
    """The blend's exponential is a DEFINED function (gd_expf) shared by the oracle and the HIP kernels, so their
    bit-equality says nothing about the definition itself.  Here the whole forward + backward of the oracle runs once
    on gd_expf and once on the C library's expf (-DGD_ORACLE_LIBM_EXP): images within the pixel tolerance of SURVEY
    8d, the same blended pairs up to a few threshold flips, gradients within the gradient tolerance."""
    inp = h.raster_inputs(P=3000, H=96, W=96, seed=4)
    a = h.oracle_forward(inp)
    b = gd_oracle.forward(inp["bg"], inp["means3D"], inp["colors_precomp"], inp["opacities"], inp["scales"],
                          inp["rotations"], inp["scale_modifier"], inp["cov3D_precomp"], inp["viewmatrix"],
                          inp["projmatrix"], inp["tanfovx"], inp["tanfovy"], inp["image_height"], inp["image_width"],
                          inp["sh"], inp["degree"], inp["campos"], omp="libm")
    assert a.num_rendered == b.num_rendered and np.array_equal(a.point_list, b.point_list)
    for x, y in ((a.color, b.color), (a.depth, b.depth), (a.alpha, b.alpha)):
        assert np.all(np.abs(x - y) <= 1e-5 + 1e-4 * np.abs(y))
    assert np.mean(a.n_contrib != b.n_contrib) < 2e-3          # a 1-ulp exp may flip a pair sitting on 1/255 or 1e-4
    gc, gd, ga = h.random_image_grads(96, 96)
    ga_, gb_ = gd_oracle.backward(a, gc, gd, ga), gd_oracle.backward(b, gc, gd, ga)
    for k in ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dscales", "dL_drotations"):
        s = np.abs(gb_[k]).max()
        assert np.abs(ga_[k] - gb_[k]).max() <= 2e-3 * s, k


def test_needle_gradients_are_conditioned_on_the_exponential():
    """the GPU backward deviates from the oracle by up to 6e-3 * max|ref| on dL_dscales /
    dL_drotations of 60:1 needles -- kernel defect or conditioning?  Settled without any GPU code: the ORACLE, run on
    the same needle inputs with two valid exponentials < 1 ulp apart (its defined gd_expf and the C library's expf),
    moves those tensors by the same amount, while the blend's own sums (colour, opacity, 2-D mean) and dL_dcov3D move
    100x less.  The GPU test's bar for the amplified chain (tests/test_raster_gpu.py) is twice what is measured here."""
    inp = h.needle_inputs(20000, 256, 12)
    a = h.oracle_forward(inp)
    b = gd_oracle.forward(inp["bg"], inp["means3D"], inp["colors_precomp"], inp["opacities"], inp["scales"],
                          inp["rotations"], inp["scale_modifier"], inp["cov3D_precomp"], inp["viewmatrix"],
                          inp["projmatrix"], inp["tanfovx"], inp["tanfovy"], inp["image_height"], inp["image_width"],
                          inp["sh"], inp["degree"], inp["campos"], omp="libm")
    assert np.mean(a.n_contrib != b.n_contrib) < 2e-3
    gc, gd, ga = h.random_image_grads(256, 256, seed=12)
    ga_, gb_ = gd_oracle.backward(a, gc, gd, ga), gd_oracle.backward(b, gc, gd, ga)
    d = {k: float(np.abs(ga_[k] - gb_[k]).max() / np.abs(gb_[k]).max()) for k in ga_ if k in gb_ and gb_[k] is not None
         and np.size(gb_[k])}
    # amplified chain: moved by the exponential alone at the 1e-3 .. 1e-2 level (measured 5.6e-3 / 6.7e-3 / 2.2e-3)
    assert 1e-3 < d["dL_dscales"] < 7e-3 and 1e-3 < d["dL_drotations"] < 7e-3 and 5e-4 < d["dL_dmeans3D"] < 7e-3, d
    # the blend's own sums and the 3-D covariance gradient: 10 .. 100x less
    for k in ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dcov3D"):
        assert d[k] < 6e-4, (k, d[k])
