"""Gaussian scene parameters for the hot path: the synthetic benchmark scene and a
parameter holder with the reference's activations.

``GaussianParams`` keeps what ``GaussianModel`` (Garment_3DGS/gaussiansplatting/scene/
gaussian_model.py:39-115,149-169) exposes to ``render()`` and to the optimiser: raw leaves
``_xyz, _features_dc, _features_rest, _scaling, _rotation, _opacity`` and the activated views
``get_xyz, get_features, get_scaling (exp), get_rotation (normalize), get_opacity (sigmoid)``.
Densification, PLY I/O and simple-knn initialisation are out of scope.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

SH_C0 = 0.28209479177387814


def synthetic_gaussians(P: int, seed: int = 0, sh_degree: int = 0):
    """Seeded host-side scene (numpy ``default_rng`` so the CPU oracle and the GPU see identical
    bits).  Already-activated values, as the rasterizer consumes them:
      xyz uniform in the unit ball; scales U(0.005, 0.02) per axis; rotations normalised N(0,1)^4;
      opacities U(0.05, 0.95); shs[P,(deg+1)^2,3]: DC = (U(0,1)-0.5)/C0, higher bands N(0,0.1).
    """
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(P, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = rng.uniform(size=(P, 1)) ** (1.0 / 3.0)
    xyz = (d * r).astype(np.float32)
    scales = rng.uniform(0.005, 0.02, size=(P, 3)).astype(np.float32)
    q = rng.normal(size=(P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    rotations = q.astype(np.float32)
    opacities = rng.uniform(0.05, 0.95, size=(P, 1)).astype(np.float32)
    M = (sh_degree + 1) ** 2
    shs = np.zeros((P, M, 3), np.float32)
    shs[:, 0, :] = ((rng.uniform(size=(P, 3)) - 0.5) / SH_C0).astype(np.float32)
    if M > 1:
        shs[:, 1:, :] = rng.normal(scale=0.1, size=(P, M - 1, 3)).astype(np.float32)
    return dict(means3D=xyz, scales=scales, rotations=rotations, opacities=opacities, shs=shs)


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


class GaussianParams(nn.Module):
    """Raw (pre-activation) leaves + the reference's activated accessors."""

    def __init__(self, scene: dict, sh_degree: int = 0, device="cuda"):
        super().__init__()
        t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=device)
        self.max_sh_degree = sh_degree
        self.active_sh_degree = sh_degree
        shs = t(scene["shs"])
        self._xyz = nn.Parameter(t(scene["means3D"]).clone())
        self._features_dc = nn.Parameter(shs[:, 0:1, :].clone().contiguous())
        self._features_rest = nn.Parameter(shs[:, 1:, :].clone().contiguous())
        self._scaling = nn.Parameter(torch.log(t(scene["scales"])))
        self._rotation = nn.Parameter(t(scene["rotations"]).clone())
        self._opacity = nn.Parameter(inverse_sigmoid(t(scene["opacities"])))

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    def param_groups(self, lr_xyz=0.00005, lr_f_dc=0.0125, lr_f_rest=0.0125 / 20.0, lr_opacity=0.01,
                     lr_scaling=0.005, lr_rotation=0.001):
        """The six Adam groups of ``training_setup`` (gaussian_model.py:156-165; default rates from
        Garment_3DGS/gaussiansplatting/arguments/__init__.py:73-80)."""
        return [
            {"params": [self._xyz], "lr": lr_xyz, "name": "xyz"},
            {"params": [self._features_dc], "lr": lr_f_dc, "name": "f_dc"},
            {"params": [self._features_rest], "lr": lr_f_rest, "name": "f_rest"},
            {"params": [self._opacity], "lr": lr_opacity, "name": "opacity"},
            {"params": [self._scaling], "lr": lr_scaling, "name": "scaling"},
            {"params": [self._rotation], "lr": lr_rotation, "name": "rotation"},
        ]
