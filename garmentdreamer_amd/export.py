"""On-disk products either side of the path: the per-view RGBA PNG + ``cameras.json`` dump that
stages 3-4 consume (Garment_3DGS/threestudio/systems/GaussianDreamer.py:330-417, utils/saving.py:331-354) and the
prompt-embedding cache key (models/prompt_processors/base.py:19-23).  ``last_3dgs.ply`` lives in
``gaussian_model.GaussianModel.save_ply``.  Pure Python / numpy (the reference uses cv2, absent here)."""
from __future__ import annotations

import hashlib
import json
import math
import os
import struct
import zlib
from typing import Dict, List

import numpy as np


def fov2focal(fov, pixels):
    """gaussiansplatting/utils/graphics_utils.py:95-96"""
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal, pixels):
    """gaussiansplatting/utils/graphics_utils.py:98-99"""
    return 2 * math.atan(pixels / (2 * focal))


def hash_prompt(model: str, prompt: str) -> str:
    """Cache key of a prompt's text embeddings: ``<cache_dir>/<md5(model-prompt)>.pt`` (base.py:19-23,413-422)."""
    return hashlib.md5(f"{model}-{prompt}".encode()).hexdigest()


def camera_info_entry(c2w, index: int, width: int, height: int, fovy: float) -> Dict:
    """One ``cameras.json`` record (GaussianDreamer.py:351-362): position = c2w translation, rotation = the
    NEGATED c2w rotation (``rot[:, :] *= -1``), fy from fovy, fx through the fov round trip the reference does."""
    C2W = np.array(c2w, dtype=np.float32, copy=True)
    pos = C2W[:3, 3]
    rot = C2W[:3, :3] * -1
    fy = fov2focal(float(fovy), height)
    return {"id": int(index), "img_name": str(int(index)), "width": width, "height": height,
            "position": pos.tolist(), "rotation": [x.tolist() for x in rot], "fy": fy,
            "fx": fov2focal(focal2fov(fy, width), width)}


def save_cameras_json(path: str, camera_info_list: List[Dict]):
    """GaussianDreamer.py:330-332"""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        json.dump(camera_info_list, f)


def _png_chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def save_image_rgba(path: str, rgb, mask) -> str:
    """``save_image_rgba`` (saving.py:331-354): rgb [H,W,3] in [0,1], mask [H,W] in {0,1} -> 8-bit RGBA PNG
    (clip, x255, round to nearest even like cv2's saturate_cast).  ``rgb`` / ``mask``: numpy or torch."""
    to_np = lambda a: a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)   # noqa: E731
    rgb, mask = to_np(rgb).astype(np.float32), to_np(mask).astype(np.float32)
    assert mask.max() <= 1.0
    img = np.concatenate((rgb, mask[..., None]), axis=-1).clip(0, 1) * np.float32(255.0)
    img8 = np.rint(img).astype(np.uint8)
    H, W = img8.shape[:2]
    raw = np.concatenate((np.zeros((H, 1), np.uint8), img8.reshape(H, W * 4)), axis=1).tobytes()   # filter 0 per row
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(_png_chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 6, 0, 0, 0)))   # 8-bit, colour type 6 = RGBA
        f.write(_png_chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(_png_chunk(b"IEND", b""))
    return path


def dump_test_view(save_dir: str, out: Dict, batch: Dict, camera_info_list: List[Dict], alpha_threshold: float = 0.5,
                   view: int = 0):
    """``test_step`` (GaussianDreamer.py:334-410) for one rendered view: thresholded alpha as the mask, RGBA PNG
    under ``gs_rendered_rgba/<index>.png``, one record appended to ``camera_info_list``."""
    alpha = out["alphas"][view].squeeze()
    rgb = out["comp_rgb"][view].squeeze()
    mask = (alpha >= alpha_threshold).to(rgb.dtype) if hasattr(alpha, "to") else (alpha >= alpha_threshold).astype(np.float32)
    idx = int(batch["index"][view])
    fovy = float(batch["fovy"][view])
    c2w = batch["c2w"][view]
    c2w = c2w.detach().cpu().numpy() if hasattr(c2w, "detach") else c2w
    camera_info_list.append(camera_info_entry(c2w, idx, batch["width"], batch["height"], fovy))
    return save_image_rgba(os.path.join(save_dir, "gs_rendered_rgba", f"{idx}.png"), rgb, mask)
