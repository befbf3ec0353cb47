"""Camera matrices for the rasterizer, in the reference's (transposed, row-vector) convention.

Restates ``Camera`` (Garment_3DGS/gaussiansplatting/scene/cameras.py:17-53), the matrix helpers of
Garment_3DGS/gaussiansplatting/utils/graphics_utils.py:59-99 and the pose construction of
Garment_3DGS/threestudio/data/uncond.py:30-54,371-390.  Pinned bit-for-bit by the golden vectors in
tests/golden/cameras.npz, which were produced by importing those reference files.

MI355X-side change: the reference builds each camera with two CPU 4x4 inverses, two H2D copies, a
GPU bmm and a GPU inverse, plus implicit syncs (cameras.py:50-53).  Here all of it is computed on
the host in float32 with the same torch operations (so results match the reference's CPU values)
and uploaded with ONE pinned, non-blocking copy per camera batch.
"""
from __future__ import annotations

import math
from typing import Sequence

import numpy as np
import torch


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


def getWorld2View2_tensor(R, t, translate=torch.tensor([.0, .0, .0]), scale=1.0):
    """graphics_utils.py:59-70."""
    Rt = torch.zeros((4, 4))
    Rt[:3, :3] = R.transpose(0, 1)
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = torch.linalg.inv(Rt)
    cam_center = C2W[:3, 3]
    cam_center = (cam_center + translate) * scale
    C2W[:3, 3] = cam_center
    Rt = torch.linalg.inv(C2W)
    return Rt.float()


def getProjectionMatrix(znear, zfar, fovX, fovY):
    """graphics_utils.py:73-93."""
    tanHalfFovY = math.tan((fovY / 2))
    tanHalfFovX = math.tan((fovX / 2))
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = torch.zeros(4, 4)
    z_sign = 1.0
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


class Camera:
    """Same attributes the renderer reads from the reference's ``Camera`` (cameras.py:17-53):
    ``FoVx, FoVy, image_height, image_width, world_view_transform, full_proj_transform,
    camera_center`` (+ ``projection_matrix``, ``R``, ``T``, ``znear``, ``zfar``)."""

    def __init__(self, c2w, FoVy, height, width, trans=torch.tensor([0.0, 0.0, 0.0]), scale=1.0,
                 data_device="cuda"):
        c2w = torch.as_tensor(c2w).detach().cpu()
        FoVy = float(FoVy)
        FoVx = focal2fov(fov2focal(FoVy, height), width)
        R = c2w[:3, :3]
        T = c2w[:3, 3]
        self.R = R.float()
        self.T = T.float()
        self.FoVx = FoVx
        self.FoVy = FoVy
        self.image_height = height
        self.image_width = width
        self.zfar = 100.0
        self.znear = 0.01
        self.trans = trans.float()
        self.scale = scale
        wvt = getWorld2View2_tensor(R, T).transpose(0, 1).float()
        proj = getProjectionMatrix(znear=self.znear, zfar=self.zfar, fovX=self.FoVx, fovY=self.FoVy) \
            .transpose(0, 1).float()
        full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).float()
        center = wvt.inverse()[3, :3].float()
        self._host = (wvt, proj, full, center)
        self.data_device = torch.device(data_device) if (data_device != "cuda" or torch.cuda.is_available()) \
            else torch.device("cpu")
        dev = self.data_device
        if dev.type == "cuda":
            packed = torch.cat([wvt.reshape(-1), proj.reshape(-1), full.reshape(-1), center.reshape(-1)]).pin_memory()
            d = packed.to(dev, non_blocking=True)
            self.world_view_transform = d[0:16].view(4, 4)
            self.projection_matrix = d[16:32].view(4, 4)
            self.full_proj_transform = d[32:48].view(4, 4)
            self.camera_center = d[48:51]
        else:
            self.world_view_transform, self.projection_matrix, self.full_proj_transform, self.camera_center = \
                wvt, proj, full, center

    @property
    def tanfovx(self):
        return math.tan(self.FoVx * 0.5)

    @property
    def tanfovy(self):
        return math.tan(self.FoVy * 0.5)


# ---- pose construction, uncond.py:30-54 ----
def _trans_t(t):
    return torch.Tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, t], [0, 0, 0, 1]]).float()


def _rot_phi(phi):
    return torch.Tensor([[1, 0, 0, 0], [0, np.cos(phi), -np.sin(phi), 0], [0, np.sin(phi), np.cos(phi), 0],
                         [0, 0, 0, 1]]).float()


def _rot_theta(th):
    return torch.Tensor([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0],
                         [0, 0, 0, 1]]).float()


def pose_spherical(theta, phi, radius):
    c2w = _trans_t(radius)
    c2w = _rot_phi(phi / 180. * np.pi) @ c2w
    c2w = _rot_theta(theta / 180. * np.pi) @ c2w
    c2w = torch.Tensor(np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]])) @ c2w
    return c2w


def c2w_3dgs(azimuth_deg: float, elevation_deg: float, camera_distance: float) -> torch.Tensor:
    """One row of ``batch['c2w_3dgs']`` (uncond.py:371-390)."""
    render_pose = pose_spherical(azimuth_deg + 180.0 - 90, -elevation_deg, camera_distance)
    matrix = torch.linalg.inv(render_pose)
    R = -torch.transpose(matrix[:3, :3], 0, 1)
    R[:, 0] = -R[:, 0]
    T = -matrix[:3, 3]
    c2w_single = torch.cat([R, T[:, None]], 1)
    c2w_single = torch.cat([c2w_single, torch.tensor([[0, 0, 0, 1]])], 0)
    return c2w_single


def orbit_batch(n_views: int, elevation_deg: float = 15.0, camera_distance: float = 2.75, fovy_deg: float = 55.0,
                height: int = 512, width: int = 512, azimuth_offset_deg: float = 0.0, view_ids=None):
    """The benchmark's synthetic camera batch: ``n_views`` azimuths evenly spaced over
    (-180, 180), fixed elevation / distance / fovy.  Returns the dict the reference's data module
    yields (uncond.py:395-408), restricted to the keys the loop consumes.  ``view_ids`` selects a
    subset (view sharding: rank r takes ``range(r, n_views, world)``)."""
    ids = list(range(n_views)) if view_ids is None else list(view_ids)
    az = [-180.0 + 360.0 * (i + 0.5) / n_views + azimuth_offset_deg for i in ids]
    c2w = torch.stack([c2w_3dgs(a, elevation_deg, camera_distance) for a in az], 0)
    B = len(ids)
    return {
        "c2w_3dgs": c2w,
        "fovy": torch.full((B,), math.radians(fovy_deg)),
        "height": height,
        "width": width,
        "elevation": torch.full((B,), float(elevation_deg)),
        "azimuth": torch.tensor(az),
        "camera_distances": torch.full((B,), float(camera_distance)),
    }


def random_batch(n_views: int, generator=None, camera_distance_range=(1.5, 4.0), fovy_range_deg=(40.0, 70.0),
                 elevation_range_deg=(-10.0, 60.0), height: int = 512, width: int = 512, view_ids=None):
    """A camera batch drawn like the reference's training data module draws one: per view an independent
    ``camera_distance`` in [1.5, 4.0], ``fovy`` in [40, 70] degrees (configs/gaussiandreamer-sd.yaml:11-12), elevation and
    azimuth uniform in their ranges (threestudio/data/uncond.py, RandomCameraIterableDataset.collate: default elevation range
    (-10, 60), azimuth (-180, 180)).  The projected area of the scene varies up to ~25x from view to view, which is what the
    rasterizer's sync-free instance capacity has to survive (tests/test_scene_gpu.py).  Same dict as ``orbit_batch``;
    ``view_ids`` selects this rank's views AFTER the draw so that every rank draws the same batch from a same-seeded
    generator."""
    u = torch.rand(n_views, 4, generator=generator)
    dist = camera_distance_range[0] + (camera_distance_range[1] - camera_distance_range[0]) * u[:, 0]
    fovy = fovy_range_deg[0] + (fovy_range_deg[1] - fovy_range_deg[0]) * u[:, 1]
    elev = elevation_range_deg[0] + (elevation_range_deg[1] - elevation_range_deg[0]) * u[:, 2]
    az = -180.0 + 360.0 * u[:, 3]
    ids = list(range(n_views)) if view_ids is None else list(view_ids)
    c2w = torch.stack([c2w_3dgs(float(az[i]), float(elev[i]), float(dist[i])) for i in ids], 0)
    sel = torch.tensor(ids, dtype=torch.long)
    return {
        "c2w_3dgs": c2w,
        "fovy": torch.deg2rad(fovy[sel]),
        "height": height,
        "width": width,
        "elevation": elev[sel].clone(),
        "azimuth": az[sel].clone(),
        "camera_distances": dist[sel].clone(),
    }


class CameraBatch:
    """V cameras packed for the batched rasterizer: one upload for all matrices."""

    def __init__(self, cameras: Sequence[Camera], device):
        self.cameras = list(cameras)
        V = len(self.cameras)
        host = torch.empty(V, 35)
        for i, c in enumerate(self.cameras):
            wvt, _proj, full, center = c._host
            host[i, 0:16] = wvt.reshape(-1)
            host[i, 16:32] = full.reshape(-1)
            host[i, 32:35] = center
        device = torch.device(device)
        if device.type == "cuda":
            d = host.pin_memory().to(device, non_blocking=True)
        else:
            d = host
        self.viewmatrix = d[:, 0:16].reshape(V, 4, 4).contiguous()
        self.projmatrix = d[:, 16:32].reshape(V, 4, 4).contiguous()
        self.campos = d[:, 32:35].contiguous()
        self.tanfovx = [c.tanfovx for c in self.cameras]
        self.tanfovy = [c.tanfovy for c in self.cameras]
        self.image_height = int(self.cameras[0].image_height)
        self.image_width = int(self.cameras[0].image_width)
