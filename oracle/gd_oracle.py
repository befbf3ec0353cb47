"""ctypes front-end of the CPU oracle (oracle/gd_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; the product package ``garmentdreamer_amd`` never
imports this module.  All arrays are numpy, float32 / int32 / uint32 / uint64, C-contiguous.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIB_OMP = None
_LIB_LIBM = None

_f32p = C.POINTER(C.c_float)


def build(force: bool = False, omp=False) -> str:
    """``omp``: False = the serial build, True = -fopenmp, "libm" = the blend on the C library's expf (tolerance
    check of gd_expf, tests only)."""
    name = {False: "libgd_oracle.so", True: "libgd_oracle_omp.so", "libm": "libgd_oracle_libm.so"}[omp]
    so = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "gd_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", name])
    return so


def lib(omp: bool = False):
    """``omp=True``: the same C file built with -fopenmp (tiles over the host cores) -- bench.py's multi-core CPU
    baseline; the parity tests use the serial, deterministic build."""
    global _LIB, _LIB_OMP, _LIB_LIBM
    if omp == "libm":
        if _LIB_LIBM is None:
            _LIB_LIBM = _bind(C.CDLL(build(omp="libm")))
        return _LIB_LIBM
    if omp:
        if _LIB_OMP is None:
            _LIB_OMP = _bind(C.CDLL(build(omp=True)))
        return _LIB_OMP
    if _LIB is None:
        _LIB = _bind(C.CDLL(build()))
    return _LIB


def _bind(L):
    L.gdo_forward.restype = C.c_void_p
    L.gdo_forward.argtypes = [C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_int] + [_f32p] * 5 + [
        C.c_float, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_float, C.c_float]
    L.gdo_backward.restype = None
    L.gdo_backward.argtypes = [C.c_void_p] + [_f32p] * 5 + [C.c_float] + [_f32p] * 5 + [C.c_float, C.c_float] + [
        _f32p] * 13
    L.gdo_free.argtypes = [C.c_void_p]
    L.gdo_mark_visible.argtypes = [C.c_int, _f32p, _f32p, _f32p, C.POINTER(C.c_uint8)]
    L.gdo_higher_msb.restype = C.c_uint32
    L.gdo_higher_msb.argtypes = [C.c_uint32]
    L.gdo_expf.restype = C.c_float
    L.gdo_expf.argtypes = [C.c_float]
    for n in ("num_rendered", "pairs_visited_fwd", "pairs_blended_fwd", "pairs_visited_bwd"):
        getattr(L, "gdo_" + n).restype = C.c_int64
        getattr(L, "gdo_" + n).argtypes = [C.c_void_p]
    for n in ("depths", "clamped", "radii", "means2D", "cov3D", "conic_opacity", "rgb", "tiles_touched",
              "point_offsets", "keys_unsorted", "vals_unsorted", "keys", "point_list", "ranges", "n_contrib",
              "out_color", "out_depth", "out_alpha"):
        getattr(L, "gdo_" + n).restype = C.c_void_p
        getattr(L, "gdo_" + n).argtypes = [C.c_void_p]
    return L


def _f(a):
    """float32 contiguous array or None (absent optional == NULL, like torch.Tensor([]))."""
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.size == 0:
        return None, None
    return a, a.ctypes.data_as(_f32p)


def _view(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0 or not ptr:
        return np.zeros(shape, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()


class OracleState:
    """Forward result + every intermediate the reference keeps in its geom/binning/img buffers."""

    def __init__(self, handle, P, M, W, H, keep, omp=False):
        L = lib(omp)
        self._omp = omp
        self._h = handle
        self._keep = keep  # inputs kept alive for backward
        self.P, self.M, self.W, self.H = P, M, W, H
        self.tiles_x, self.tiles_y = (W + 15) // 16, (H + 15) // 16
        tiles = self.tiles_x * self.tiles_y
        R = self.num_rendered = int(L.gdo_num_rendered(handle))
        g = lambda n: getattr(L, "gdo_" + n)(handle)
        self.depths = _view(g("depths"), (P,), np.float32)
        self.clamped = _view(g("clamped"), (P, 3), np.uint8)
        self.radii = _view(g("radii"), (P,), np.int32)
        self.means2D = _view(g("means2D"), (P, 2), np.float32)
        self.cov3D = _view(g("cov3D"), (P, 6), np.float32)
        self.conic_opacity = _view(g("conic_opacity"), (P, 4), np.float32)
        self.rgb = _view(g("rgb"), (P, 3), np.float32)
        self.tiles_touched = _view(g("tiles_touched"), (P,), np.uint32)
        self.point_offsets = _view(g("point_offsets"), (P,), np.uint32)
        self.keys_unsorted = _view(g("keys_unsorted"), (R,), np.uint64)
        self.vals_unsorted = _view(g("vals_unsorted"), (R,), np.uint32)
        self.keys = _view(g("keys"), (R,), np.uint64)
        self.point_list = _view(g("point_list"), (R,), np.uint32)
        self.ranges = _view(g("ranges"), (tiles, 2), np.uint32)
        self.n_contrib = _view(g("n_contrib"), (H, W), np.uint32)
        self.color = _view(g("out_color"), (3, H, W), np.float32)
        self.depth = _view(g("out_depth"), (1, H, W), np.float32)
        self.alpha = _view(g("out_alpha"), (1, H, W), np.float32)
        self.pairs_visited_fwd = int(L.gdo_pairs_visited_fwd(handle))
        self.pairs_blended_fwd = int(L.gdo_pairs_blended_fwd(handle))
        self.pairs_visited_bwd = 0

    def __del__(self):
        try:
            if self._h:
                lib(self._omp).gdo_free(self._h)
                self._h = None
        except Exception:
            pass


def forward(bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
            projmatrix, tanfovx, tanfovy, image_height, image_width, sh, degree, campos, omp: bool = False) -> OracleState:
    """Same argument meaning as ``_C.rasterize_gaussians`` (DGR/rasterize_points.cu:35-56)."""
    means3D = np.ascontiguousarray(means3D, dtype=np.float32)
    P = means3D.shape[0]
    sh_a = None if sh is None else np.ascontiguousarray(sh, dtype=np.float32)
    M = 0 if sh_a is None or sh_a.size == 0 else sh_a.shape[1]
    keep = {}
    ptr = {}
    for name, arr in dict(bg=bg, means3D=means3D, sh=sh_a, colors=colors_precomp, opac=opacities, scales=scales,
                          rot=rotations, cov=cov3D_precomp, view=viewmatrix, proj=projmatrix, campos=campos).items():
        keep[name], ptr[name] = _f(arr)
    h = lib(omp).gdo_forward(P, int(degree), M, ptr["bg"], int(image_width), int(image_height), ptr["means3D"],
                          ptr["sh"], ptr["colors"], ptr["opac"], ptr["scales"], float(scale_modifier), ptr["rot"],
                          ptr["cov"], ptr["view"], ptr["proj"], ptr["campos"], float(tanfovx), float(tanfovy))
    keep["ptr"] = ptr
    keep["args"] = dict(scale_modifier=float(scale_modifier), tanfovx=float(tanfovx), tanfovy=float(tanfovy))
    return OracleState(h, P, M, int(image_width), int(image_height), keep, omp)


def backward(st: OracleState, dL_dcolor, dL_ddepth, dL_dalpha) -> dict:
    """Gradients in the native tuple's naming (DGR/rasterize_points.cu:155-207)."""
    P, M = st.P, st.M
    k, p, a = st._keep, st._keep["ptr"], st._keep["args"]
    gc, gcp = _f(np.asarray(dL_dcolor, dtype=np.float32).reshape(3, st.H, st.W))
    gd, gdp = _f(np.asarray(dL_ddepth, dtype=np.float32).reshape(st.H, st.W))
    ga, gap = _f(np.asarray(dL_dalpha, dtype=np.float32).reshape(st.H, st.W))
    out = dict(dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dconic=np.zeros((P, 2, 2), np.float32),
               dL_dopacity=np.zeros((P, 1), np.float32), dL_dcolors=np.zeros((P, 3), np.float32),
               dL_ddepths=np.zeros((P, 1), np.float32), dL_dmeans3D=np.zeros((P, 3), np.float32),
               dL_dcov3D=np.zeros((P, 6), np.float32), dL_dsh=np.zeros((P, max(M, 0), 3), np.float32),
               dL_dscales=np.zeros((P, 3), np.float32), dL_drotations=np.zeros((P, 4), np.float32))
    op = {n: v.ctypes.data_as(_f32p) for n, v in out.items()}
    lib(st._omp).gdo_backward(st._h, p["bg"], p["means3D"], p["sh"], p["colors"], p["scales"], a["scale_modifier"],
                       p["rot"], p["cov"], p["view"], p["proj"], p["campos"], a["tanfovx"], a["tanfovy"], gcp, gdp,
                       gap, op["dL_dmeans2D"], op["dL_dconic"], op["dL_dopacity"], op["dL_dcolors"],
                       op["dL_ddepths"], op["dL_dmeans3D"], op["dL_dcov3D"], op["dL_dsh"], op["dL_dscales"],
                       op["dL_drotations"])
    st.pairs_visited_bwd = int(lib(st._omp).gdo_pairs_visited_bwd(st._h))
    return out


def mark_visible(means3D, viewmatrix, projmatrix) -> np.ndarray:
    m, mp = _f(means3D)
    v, vp = _f(viewmatrix)
    pr, pp = _f(projmatrix)
    P = 0 if m is None else m.shape[0]
    out = np.zeros(P, np.uint8)
    if P:
        lib().gdo_mark_visible(P, mp, vp, pp, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.astype(bool)


def expf(x: float) -> float:
    """The blend's exp as the oracle (and the HIP kernels) define it: gd_expf in gd_oracle.c."""
    return float(lib().gdo_expf(float(x)))


def higher_msb(n: int) -> int:
    return int(lib().gdo_higher_msb(n))


# ---------------------------------------------------------------------------------------------
# scene-side oracle (oracle/gd_scene_oracle.c): simple-knn distCUDA2
# ---------------------------------------------------------------------------------------------
_SLIB = None


def build_scene(force: bool = False) -> str:
    so = os.path.join(_HERE, "libgd_scene_oracle.so")
    src = os.path.join(_HERE, "gd_scene_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libgd_scene_oracle.so"])
    return so


def dist2(points, return_order: bool = False):
    """Mean squared distance to the 3 nearest neighbours, the reference's boxed Morton search
    (simple_knn.cu:153-220).  points: [P,3] float32."""
    global _SLIB
    if _SLIB is None:
        _SLIB = C.CDLL(build_scene())
        _SLIB.gdso_dist2.restype = None
        _SLIB.gdso_dist2.argtypes = [C.c_int, _f32p, _f32p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    pts = np.ascontiguousarray(points, dtype=np.float32)
    P = pts.shape[0]
    out = np.zeros(P, np.float32)
    codes = np.zeros(P, np.uint32)
    order = np.zeros(P, np.uint32)
    _SLIB.gdso_dist2(P, pts.ctypes.data_as(_f32p), out.ctypes.data_as(_f32p), codes.ctypes.data_as(C.POINTER(C.c_uint32)),
                     order.ctypes.data_as(C.POINTER(C.c_uint32)))
    return (out, codes, order) if return_order else out
