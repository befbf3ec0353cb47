import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
from tests import helpers as h
from tests.test_raster_gpu import _needle_inputs, _run_gpu_forward, DEV
from garmentdreamer_amd.diff_gaussian_rasterization import _C
from oracle import gd_oracle

P, HW, seed = 3000, 128, 11
inp = _needle_inputs(P, HW, seed)
st = h.oracle_forward(inp)
args, out = _run_gpu_forward(inp)
gc, gd, ga = h.random_image_grads(HW, HW, seed=seed)
ref = gd_oracle.backward(st, gc, gd, ga)
R, color, depth, alpha, radii, geom, binning, img = out
t = lambda a: torch.as_tensor(a, device=DEV)
(bg, means3D, colors, opac, scales, rots, smod, cov, vm, pm, tx, ty, H, W, sh, degree, campos, _, _) = args
grads = _C.rasterize_gaussians_backward(bg, means3D, radii, colors, scales, rots, smod, cov, vm, pm, tx, ty, t(gc), t(gd),
                                        t(ga), sh, degree, campos, geom, R, binning, img, alpha, False)
torch.cuda.synchronize()
names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")
for n, g in zip(names, grads):
    o = ref[n]
    g = g.detach().cpu().numpy().reshape(o.shape)
    err = np.abs(g - o)
    scale = np.abs(o).max()
    print(n, "scale", scale, "max err", err.max(), "max rel(err/scale)", err.max() / scale)
    if n in ("dL_dmeans2D", "dL_dopacity", "dL_dcolors"):
        idx = np.argsort(-err.reshape(o.shape[0], -1).max(1))[:6]
        for i in idx:
            print("   id", i, "gpu", g[i].ravel()[:3], "ref", o[i].ravel()[:3], "opac", inp["opacities"][i], "scales", inp["scales"][i],
                  "radius", st.radii[i], "conic_o", st.conic_opacity[i])
