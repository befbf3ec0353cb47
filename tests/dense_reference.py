"""Independent float64 torch-autograd statement of the splatting MATH (EWA projection + ordered
alpha compositing), written from the equations rather than from the reference's kernels.

Purpose: pin the oracle's hand-derived backward (oracle/gd_oracle.c, following
DGR/cuda_rasterizer/backward.cu) against automatic differentiation.  Dense O(pixels x P), so
only for tiny cases.  The reference's discrete decisions (near cull z<=0.2, tile rectangle,
power>0, alpha<1/255, 0.99 clamp, stop when T(1-alpha)<1e-4) are applied as constant masks
derived inside this function -- they are piecewise constant, so autograd through the rest is exact
wherever the oracle's analytic gradient is defined.
"""
from __future__ import annotations

import math

import torch

SH_C0 = 0.28209479177387814


def render_dense(means3D, scales, rotations, opacities, shs_dc, view, proj, campos, tanfovx, tanfovy, H, W, bg,
                 tile_mask=None):
    """All tensor args float64 (requires_grad as needed).  view/proj: the reference's transposed
    4x4 matrices.  shs_dc: [P,3] degree-0 coefficients.  tile_mask: optional bool [H*W, P] saying
    which (pixel, Gaussian) pairs the tile binning lets meet.  Returns color[3,H,W], depth[1,H,W],
    alpha[1,H,W]."""
    P = means3D.shape[0]
    dt = means3D.dtype
    fx = W / (2.0 * tanfovx)
    fy = H / (2.0 * tanfovy)
    ones = torch.ones(P, 1, dtype=dt)
    hom = torch.cat([means3D, ones], 1)
    p_view = hom @ view  # row-vector convention
    p_hom = hom @ proj
    p_w = 1.0 / (p_hom[:, 3] + 0.0000001)
    ndc = p_hom[:, :2] * p_w[:, None]
    mx = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    my = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    tz = p_view[:, 2]
    visible = tz > 0.2

    # covariance in world space: R diag(s^2) R^T, quaternion (r,x,y,z) NOT normalised
    r, x, y, z = rotations.unbind(1)
    Rm = torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], 1),
        torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], 1),
        torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1)], 1)
    Sigma = Rm @ torch.diag_embed(scales * scales) @ Rm.transpose(1, 2)

    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz = p_view[:, 0] / tz
    tytz = p_view[:, 1] / tz
    tx = torch.clamp(txtz, -limx, limx) * tz
    ty = torch.clamp(tytz, -limy, limy) * tz
    zero = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz)], 1),
                     torch.stack([zero, fy / tz, -(fy * ty) / (tz * tz)], 1)], 1)  # [P,2,3]
    Rwv = view[:3, :3].transpose(0, 1)  # rows (v0,v4,v8),...
    A = J @ Rwv
    cov = A @ Sigma @ A.transpose(1, 2)
    a = cov[:, 0, 0] + 0.3
    b = cov[:, 0, 1]
    c = cov[:, 1, 1] + 0.3
    det = a * c - b * b
    con_a, con_b, con_c = c / det, -b / det, a / det

    rgb = torch.clamp_min(SH_C0 * shs_dc + 0.5, 0.0)

    order = torch.argsort(tz.detach(), stable=True)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    pxf, pyf = xs.reshape(-1, 1), ys.reshape(-1, 1)
    dx = mx[order][None] - pxf
    dy = my[order][None] - pyf
    power = -0.5 * (con_a[order][None] * dx * dx + con_c[order][None] * dy * dy) - con_b[order][None] * dx * dy
    alpha = torch.clamp_max(opacities[order, 0][None] * torch.exp(power), 0.99)
    keep = (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0) & visible[order][None]
    if tile_mask is not None:
        keep = keep & tile_mask[:, order]
    alpha_eff = torch.where(keep, alpha, torch.zeros_like(alpha))
    # sequential early stop: a pair is blended only if T*(1-alpha) >= 1e-4 at its turn and no earlier stop
    with torch.no_grad():
        T = torch.ones(H * W, dtype=dt)
        stopped = torch.zeros(H * W, dtype=torch.bool)
        blend = torch.zeros_like(keep)
        for j in range(P):
            k = keep[:, j] & ~stopped
            test_T = T * (1 - alpha_eff[:, j])
            stop_now = k & (test_T < 0.0001)
            stopped |= stop_now
            ok = k & ~stop_now
            blend[:, j] = ok
            T = torch.where(ok, test_T, T)
    a_b = torch.where(blend, alpha_eff, torch.zeros_like(alpha_eff))
    one_minus = 1 - a_b
    T_excl = torch.cumprod(torch.cat([torch.ones(H * W, 1, dtype=dt), one_minus[:, :-1]], 1), 1)
    w = a_b * T_excl
    T_final = torch.prod(one_minus, 1)
    color = w @ rgb[order] + T_final[:, None] * bg[None]
    depth = w @ tz[order]
    asum = w.sum(1)
    return (color.transpose(0, 1).reshape(3, H, W), depth.reshape(1, H, W), asum.reshape(1, H, W))


def tile_mask_from_oracle(st, H, W):
    """[H*W, P] bool: pixel's 16x16 tile lies inside the Gaussian's tile rectangle."""
    import numpy as np
    P = st.P
    gx, gy = (W + 15) // 16, (H + 15) // 16
    mask = np.zeros((H * W, P), bool)
    for g in range(P):
        r = int(st.radii[g])
        if r <= 0:
            continue
        px, py = float(st.means2D[g, 0]), float(st.means2D[g, 1])
        x0 = min(gx, max(0, int((np.float32(px) - r) / 16)))
        y0 = min(gy, max(0, int((np.float32(py) - r) / 16)))
        x1 = min(gx, max(0, int((np.float32(px) + r + 16 - 1) / 16)))
        y1 = min(gy, max(0, int((np.float32(py) + r + 16 - 1) / 16)))
        m = np.zeros((H, W), bool)
        m[y0 * 16:y1 * 16, x0 * 16:x1 * 16] = True
        mask[:, g] = m.reshape(-1)
    return torch.from_numpy(mask)
