"""In-tree build of the HIP libraries (hipcc cross-compiles gfx950 without a GPU).

``python -m garmentdreamer_amd._build`` or ``__graft_entry__.build()``.  Objects and the
resulting ``.so`` files stay next to the sources (git-ignored, but they travel with the
working tree to the GPU box).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ARCH = "gfx950"

COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]

# (source, extra flags).  raster_preprocess: no FMA contraction -- radii / tile rects / depth-bit
# keys must be bit-identical to oracle/gd_oracle.c (also built with -ffp-contract=off).
RASTER_SOURCES = [
    ("raster_preprocess.hip", ["-ffp-contract=off"]),
    ("raster_binning.hip", []),
    ("raster_render.hip", []),
    # the SLP vectoriser pairs scalar fp32 adds into v_pk_add_f32, which keeps the DPP moves of the row scans from
    # being fused into their adds; the packed math of this file is written out as float2
    ("raster_render_bwd.hip", ["-fno-slp-vectorize"]),
    ("raster_api.hip", []),
    ("raster_scene.hip", ["-ffp-contract=off"]),
    ("raster_densify.hip", ["-ffp-contract=off"]),
]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the HIP rasterizer cannot be built")
    return exe


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    inc = os.path.join(HERE, "..", "include")
    hs += [os.path.join(inc, f) for f in sorted(os.listdir(inc)) if f.endswith(".h")]   # gd_raster.h, gd_nn.h, gd_scene.h
    return hs


def build_library(name: str, sources, force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    objs = []
    hdrs = _headers()
    for src, extra in sources:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        if force or _newer(o, [s] + hdrs):
            cmd = [hipcc] + COMMON + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    so = os.path.join(HERE, name)
    if force or _newer(so, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", so] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return so


def build_all(force: bool = False, verbose: bool = False):
    out = [build_library("libgd_raster.so", RASTER_SOURCES, force, verbose)]
    try:
        from . import _build_nn  # optional second library (dense kernels), added when present
        out += _build_nn.build(force, verbose)
    except ImportError:
        pass
    return out


if __name__ == "__main__":
    for p in build_all(force="--force" in sys.argv, verbose=True):
        print("built", p)
