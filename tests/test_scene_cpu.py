"""CPU checks of the scene-side logic: the simple-knn oracle against a brute-force
3-NN, the host helpers against golden vectors generated from the reference's Python
(tests/golden/make_golden_scene.py), densification / pruning bookkeeping, and the PLY codec."""
import os

import numpy as np
import pytest
import torch

from garmentdreamer_amd import gaussian_model as gm

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "scene_helpers.npz"))


def _brute_dist2(pts):
    dx = pts[None, :, 0] - pts[:, None, 0]
    dy = pts[None, :, 1] - pts[:, None, 1]
    dz = pts[None, :, 2] - pts[:, None, 2]
    D = (dx * dx + dy * dy) + dz * dz          # same fp32 expression order as simple_knn.cu:139-140
    np.fill_diagonal(D, np.inf)
    b = np.sort(D, axis=1)[:, :3].astype(np.float32)
    return ((b[:, 0] + b[:, 1]) + b[:, 2]) / np.float32(3.0)


@pytest.mark.parametrize("P,seed", [(4, 0), (37, 1), (1024, 2), (1025, 3), (5000, 4)])
def test_knn_oracle_equals_brute_force_bit_for_bit(P, seed):
    from oracle import gd_oracle
    rng = np.random.default_rng(seed)
    pts = (rng.normal(size=(P, 3)) * np.array([1.0, 0.4, 2.5]) + np.array([0.3, -0.2, 1.5])).astype(np.float32)
    pts[: P // 8] = pts[P // 8: 2 * (P // 8)]           # exact duplicates: distance 0 ties
    d, codes, order = gd_oracle.dist2(pts, return_order=True)
    ref = _brute_dist2(pts)
    assert np.array_equal(d.view(np.uint32), ref.view(np.uint32))
    # Morton codes are sorted and stable, and the bounding box includes the origin (reduce init = 0)
    assert (np.diff(codes[order].astype(np.int64)) >= 0).all()
    same = codes[order][1:] == codes[order][:-1]
    assert (order[1:][same] > order[:-1][same]).all()
    assert codes.max() < (1 << 30)


def test_knn_oracle_bounding_box_contains_origin_quirk():
    """All points far from the origin: the reference's reduction still starts from {0,0,0}
    (simple_knn.cu:190-197), so the normalised coordinates never reach 0 and the low Morton cells stay empty."""
    from oracle import gd_oracle
    rng = np.random.default_rng(9)
    pts = (rng.uniform(size=(500, 3)) + 10.0).astype(np.float32)
    _, codes, _ = gd_oracle.dist2(pts, return_order=True)
    assert codes.min() > 0x30000000 >> 1     # every axis lands in the top ~10 % of its range


def test_expon_lr_func_matches_reference_golden():
    f1 = gm.get_expon_lr_func(lr_init=0.00016 * 5.0, lr_final=0.0000016 * 5.0, lr_delay_mult=0.01, max_steps=30000)
    f2 = gm.get_expon_lr_func(lr_init=0.01, lr_final=0.0001, lr_delay_steps=200, lr_delay_mult=0.1, max_steps=1000)
    steps = GOLD["lr_steps"]
    np.testing.assert_array_equal(np.array([f1(int(s)) for s in steps]), GOLD["lr_xyz"])
    np.testing.assert_array_equal(np.array([f2(int(s)) for s in steps]), GOLD["lr_delay"])
    a = gm.OptimizationParams
    f3 = gm.get_expon_lr_func(lr_init=a.position_lr_init, lr_final=a.position_lr_final,
                              lr_delay_mult=a.position_lr_delay_mult, max_steps=a.position_lr_max_steps)
    np.testing.assert_array_equal(np.array([f3(int(s)) for s in steps]), GOLD["lr_fork"])


def test_optimization_params_match_reference_golden():
    """arguments/__init__.py:63-90 of THIS fork (not vanilla 3DGS), captured by tests/golden/make_golden_scene.py."""
    import json
    import os
    from garmentdreamer_amd.scene import GaussianParams
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "optimization_params.json")))["optimization"]
    for k in ("position_lr_init", "position_lr_final", "position_lr_delay_mult", "position_lr_max_steps", "feature_lr",
              "opacity_lr", "scaling_lr", "rotation_lr", "percent_dense", "densification_interval",
              "opacity_reset_interval", "densify_from_iter", "densify_until_iter", "densify_grad_threshold"):
        assert getattr(gm.OptimizationParams, k) == ref[k], k
    g = GaussianParams({"means3D": np.zeros((2, 3), np.float32), "shs": np.zeros((2, 1, 3), np.float32),
                        "scales": np.ones((2, 3), np.float32), "rotations": np.ones((2, 4), np.float32),
                        "opacities": np.full((2, 1), 0.5, np.float32)}, device="cpu")
    lrs = {d["name"]: d["lr"] for d in g.param_groups()}
    assert lrs == {"xyz": ref["position_lr_init"], "f_dc": ref["feature_lr"], "f_rest": ref["feature_lr"] / 20.0,
                   "opacity": ref["opacity_lr"], "scaling": ref["scaling_lr"], "rotation": ref["rotation_lr"]}


def test_build_rotation_and_rgb2sh_match_reference_golden():
    R = gm.build_rotation(torch.from_numpy(GOLD["rot_q"]))
    np.testing.assert_allclose(R.numpy(), GOLD["rot_R"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(gm.RGB2SH(torch.from_numpy(GOLD["rgb"])).numpy(), GOLD["sh"])


def _toy_model(P=50, deg=1, seed=0):
    g = torch.Generator().manual_seed(seed)
    m = gm.GaussianModel(sh_degree=deg, device="cpu")
    M = (deg + 1) ** 2
    m._pack({"xyz": torch.randn(P, 3, generator=g), "f_dc": torch.randn(P, 1, 3, generator=g),
             "f_rest": torch.randn(P, M - 1, 3, generator=g), "opacity": torch.randn(P, 1, generator=g),
             "scaling": torch.randn(P, 3, generator=g) * 0.3 - 2.0, "rotation": torch.randn(P, 4, generator=g)})
    m.spatial_lr_scale = 1.0
    m.training_setup()
    return m


def test_flat_layout_views_and_gradient_accumulation():
    m = _toy_model()
    P = 50
    assert m._flat.numel() == P * (3 + 3 + 9 + 1 + 3 + 4)
    # every parameter and its .grad alias the flat buffers
    for p in (m._xyz, m._features_dc, m._features_rest, m._opacity, m._scaling, m._rotation):
        assert p.data_ptr() >= m._flat.data_ptr() and p.grad.data_ptr() >= m._grad.data_ptr()
    loss = (m.get_xyz ** 2).sum() + m.get_opacity.sum() + (m.get_scaling * 2).sum() + m.get_features.sum()
    loss.backward()
    assert torch.allclose(m.flat_grad[: 3 * P].view(P, 3), 2 * m._xyz.data)      # accumulated IN the flat buffer
    assert m.flat_grad.abs().sum() > 0
    m.zero_grad()
    assert m.flat_grad.abs().sum() == 0 and m._xyz.grad.abs().sum() == 0
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.step()


def test_densify_clone_split_prune_bookkeeping():
    m = _toy_model(P=40, deg=0, seed=3)
    m._exp_avg.uniform_(0.1, 1.0)
    m._exp_avg_sq.uniform_(0.1, 1.0)
    m.percent_dense = 0.01
    extent = 10.0
    # points 0..9: large gradient + small scale -> clone; 10..14: large gradient + large scale -> split
    with torch.no_grad():
        m._scaling.data[:] = -5.0
        m._scaling.data[10:15] = 1.0
        m._opacity.data[:] = 2.0
        m._opacity.data[30:33] = -9.0        # sigmoid < 0.005 -> pruned
    m.xyz_gradient_accum[:15] = 1.0
    m.denom[:] = 1.0
    m.denom[35:] = 0.0                        # 0/0 -> NaN -> 0 (gaussian_model.py:396)
    xyz_before = m._xyz.data.clone()
    ea_before = m._group_views(m._exp_avg)["xyz"].clone()
    g = torch.Generator().manual_seed(0)
    # CPU: the op-for-op statement of the reference's sequence (the product path is two HIP passes and has no CPU form;
    # tests/test_scene_gpu.py holds it against this statement on the GPU)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.densify_and_prune(max_grad=0.5, min_opacity=0.005, extent=extent, max_screen_size=None, generator=g)
    m.densify_and_prune_torch(max_grad=0.5, min_opacity=0.005, extent=extent, max_screen_size=None, generator=g)
    # 40 + 10 clones + (5 split -> 10 new - 5 removed) - 3 pruned = 52
    assert m._xyz.shape[0] == 52
    for t in (m._features_dc, m._features_rest, m._opacity, m._scaling, m._rotation):
        assert t.shape[0] == 52
    assert m.xyz_gradient_accum.shape == (52, 1) and m.denom.shape == (52, 1) and m.max_radii2D.shape == (52,)
    assert m._flat.numel() == 52 * 14 and m._exp_avg.numel() == 52 * 14
    # survivors keep their values and Adam moments, new points start with zero moments
    keep = [i for i in range(40) if not (10 <= i < 15) and not (30 <= i < 33)]
    assert torch.equal(m._xyz.data[: len(keep)], xyz_before[keep])
    ea = m._group_views(m._exp_avg)["xyz"]
    assert torch.equal(ea[: len(keep)], ea_before[keep])
    assert ea[len(keep):].abs().sum() == 0
    # clones are exact copies of points 0..9; split children have scale / (0.8 * 2)
    assert torch.equal(m._xyz.data[len(keep): len(keep) + 10], xyz_before[:10])
    assert torch.allclose(m._scaling.data[-10:], torch.log(torch.exp(torch.tensor(1.0)) / 1.6).expand(10, 3))
    # gradients alias the new flat buffer
    assert m._xyz.grad.data_ptr() == m._grad.data_ptr()


def test_reset_opacity_and_lr_schedule():
    m = _toy_model(P=20, deg=0)
    m._exp_avg.fill_(1.0)
    m.reset_opacity()
    assert (m.get_opacity <= 0.01 + 1e-7).all()
    v = m._group_views(m._exp_avg)
    assert v["opacity"].abs().sum() == 0 and v["xyz"].abs().sum() > 0
    lr0 = m.update_learning_rate(0)
    lr1 = m.update_learning_rate(30000)
    # lr_delay_steps = 0: the delay multiplier is inactive (general_utils.py:51-57)
    assert abs(lr0 - 0.00005) < 1e-12 and abs(lr1 - 0.000025) < 1e-12 and m.lrs["xyz"] == lr1


def test_ply_round_trip_and_header_layout(tmp_path):
    m = _toy_model(P=17, deg=2, seed=5)
    path = str(tmp_path / "sub" / "last_3dgs.ply")
    m.save_ply(path)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode("ascii").split("\n")
    assert lines[0] == "ply" and lines[1] == "format binary_little_endian 1.0" and lines[2] == "element vertex 17"
    props = [ln.split()[-1] for ln in lines[3:] if ln.startswith("property float ")]
    # construct_list_of_attributes order (gaussian_model.py:187-203): x y z nx ny nz f_dc_* f_rest_* opacity scale_* rot_*
    assert props == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(24)] + \
        ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    assert len(body) == 17 * len(props) * 4
    row0 = np.frombuffer(body[: len(props) * 4], dtype="<f4")
    np.testing.assert_array_equal(row0[:3], m._xyz.data[0].numpy())
    np.testing.assert_array_equal(row0[3:6], 0)
    # f_rest is stored channel-major (transpose(1,2).flatten): f_rest_0.. = all coefficients of channel 0
    np.testing.assert_array_equal(row0[9:9 + 8], m._features_rest.data[0, :, 0].numpy())
    m2 = gm.GaussianModel(sh_degree=2, device="cpu")
    m2.load_ply(path)
    for a, b in ((m._xyz, m2._xyz), (m._features_dc, m2._features_dc), (m._features_rest, m2._features_rest),
                 (m._opacity, m2._opacity), (m._scaling, m2._scaling), (m._rotation, m2._rotation)):
        assert torch.equal(a.data, b.data)
    assert m2.active_sh_degree == 2
    # ascii files (plyfile's text=True) load too
    names, data = gm.read_ply(path)
    apath = str(tmp_path / "a.ply")
    with open(apath, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment test\nelement vertex 17\n" + "".join(f"property float {n}\n" for n in names)
                + "end_header\n")
        for r in data:
            f.write(" ".join(repr(float(x)) for x in r) + "\n")
    n2, d2 = gm.read_ply(apath)
    assert n2 == names and np.array_equal(d2, data)


def test_export_helpers_match_reference_golden_and_camera_record():
    import json
    from garmentdreamer_amd import export as ex
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "export_helpers.json")))
    for i, f in enumerate(gold["fov"]):
        for j, p in enumerate(gold["pixels"]):
            assert ex.fov2focal(f, p) == gold["fov2focal"][i][j]
            assert ex.focal2fov(ex.fov2focal(f, p), 777) == gold["focal2fov"][i][j]
    for k, v in gold["hash"].items():
        model, prompt = k.split("-a wedding", 1)[0], "a wedding" + k.split("-a wedding", 1)[1]
        assert ex.hash_prompt(model, prompt) == v
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, :3] = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], np.float32)
    c2w[:3, 3] = [1.0, 2.0, 3.0]
    keep = c2w.copy()
    rec = ex.camera_info_entry(c2w, 7, 640, 480, 0.9)
    assert np.array_equal(c2w, keep)                                   # the caller's matrix is not negated in place
    assert rec["id"] == 7 and rec["img_name"] == "7" and rec["width"] == 640 and rec["height"] == 480
    assert rec["position"] == [1.0, 2.0, 3.0]
    assert np.array_equal(np.array(rec["rotation"]), -keep[:3, :3])    # GaussianDreamer.py:356 ``rot[:, :] *= -1``
    assert rec["fy"] == ex.fov2focal(0.9, 480) and abs(rec["fx"] - rec["fy"]) < 1e-9 * rec["fy"]


def test_rgba_png_dump_and_cameras_json(tmp_path):
    import json
    import struct
    import zlib
    from garmentdreamer_amd import export as ex
    rng = np.random.default_rng(0)
    H, W = 37, 53
    out = {"comp_rgb": torch.from_numpy(rng.uniform(-0.1, 1.1, size=(1, H, W, 3)).astype(np.float32)),
           "alphas": torch.from_numpy(rng.uniform(size=(1, H, W, 1)).astype(np.float32))}
    batch = {"index": torch.tensor([12]), "fovy": torch.tensor([0.96]), "c2w": torch.eye(4)[None], "width": W, "height": H}
    cams = []
    path = ex.dump_test_view(str(tmp_path), out, batch, cams, alpha_threshold=0.5)
    assert path.endswith(os.path.join("gs_rendered_rgba", "12.png")) and len(cams) == 1 and cams[0]["id"] == 12
    raw = open(path, "rb").read()
    assert raw[:8] == bytes([0x89, 0x50, 0x4E, 0x47, 0x0D, 0x0A, 0x1A, 0x0A])
    # independent decode: walk the chunks, check CRCs, inflate, strip filter bytes
    pos, chunks = 8, {}
    while pos < len(raw):
        n, tag = struct.unpack(">I4s", raw[pos:pos + 8])
        data = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + data) & 0xFFFFFFFF
        chunks[tag] = chunks.get(tag, b"") + data
        pos += 12 + n
    assert struct.unpack(">IIBBBBB", chunks[b"IHDR"]) == (W, H, 8, 6, 0, 0, 0)
    px = np.frombuffer(zlib.decompress(chunks[b"IDAT"]), np.uint8).reshape(H, 1 + 4 * W)
    assert (px[:, 0] == 0).all()
    px = px[:, 1:].reshape(H, W, 4)
    exp_rgb = np.rint(out["comp_rgb"][0].numpy().clip(0, 1) * 255.0).astype(np.uint8)
    exp_a = np.where(out["alphas"][0, :, :, 0].numpy() >= 0.5, 255, 0).astype(np.uint8)
    assert np.array_equal(px[..., :3], exp_rgb) and np.array_equal(px[..., 3], exp_a)
    try:
        from PIL import Image
        im = np.array(Image.open(path))
        assert im.shape == (H, W, 4) and np.array_equal(im, px)
    except ImportError:
        pass
    jpath = str(tmp_path / "cameras.json")
    ex.save_cameras_json(jpath, cams)
    assert json.load(open(jpath)) == cams
