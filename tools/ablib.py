"""Imported by the scripts under tools/ right after ``sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)``: same-box A/B of experimental library
builds.  ``GD_NN_LIB=ablate/libgd_nn_x.so python tools/some_bench.py`` / ``GD_RASTER_LIB=...`` point the package at another
build through its explicit ``use_library`` API -- the package itself does not read these variables."""
import os

if os.environ.get("GD_NN_LIB"):
    from garmentdreamer_amd import nn_ops
    nn_ops.use_library(os.environ["GD_NN_LIB"])
if os.environ.get("GD_RASTER_LIB"):
    from garmentdreamer_amd import _native
    _native.use_library(os.environ["GD_RASTER_LIB"])
