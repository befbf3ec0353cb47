import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
from tests import helpers as h
from tests.test_raster_gpu import _run_gpu_forward, DEV
from garmentdreamer_amd.diff_gaussian_rasterization import _C
from oracle import gd_oracle

HW = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
inp = h.raster_inputs(P=30000, H=HW, W=HW, sh_degree=deg, seed=21, scale_mul=1.5)
st = h.oracle_forward(inp)
args, out = _run_gpu_forward(inp)
gc, gd, ga = h.random_image_grads(HW, HW, seed=3)
ref = gd_oracle.backward(st, gc, gd, ga)
R, color, depth, alpha, radii, geom, binning, img = out
sc = h.read_scratch(geom, binning, img, 30000, 1, HW, HW, R)
print("n_contrib mismatches", int((sc["n_contrib"][0] != st.n_contrib).sum()), "of", st.n_contrib.size)
print("blended pairs gpu", int(sc["pair_counts"][0, :, :, 1].sum()), "oracle", st.pairs_blended_fwd)
t = lambda a: torch.as_tensor(a, device=DEV)
(bg, means3D, colors, opac, scales, rots, smod, cov, vm, pm, tx, ty, H, W, sh, degree, campos, _, _) = args
grads = _C.rasterize_gaussians_backward(bg, means3D, radii, colors, scales, rots, smod, cov, vm, pm, tx, ty, t(gc), t(gd),
                                        t(ga), sh, degree, campos, geom, R, binning, img, alpha, False)
torch.cuda.synchronize()
names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity")
for n, g in zip(names, grads):
    o = ref[n]
    g = g.detach().cpu().numpy().reshape(o.shape)
    err = np.abs(g - o)
    scale = np.abs(o).max()
    tol = 1e-3 * np.abs(o) + 2e-5 * scale
    bad = (err > tol).reshape(o.shape[0], -1).any(1)
    print(n, "scale", scale, "max err", err.max(), "bad", int(bad.sum()))
    idx = np.argsort(-err.reshape(o.shape[0], -1).max(1))[:5]
    for i in idx:
        print("   id", i, "gpu", g[i].ravel()[:3], "ref", o[i].ravel()[:3], "opac", inp["opacities"][i], "radius", st.radii[i],
              "depth", st.depths[i], "conic", st.conic_opacity[i][:3], "xy", st.means2D[i])
