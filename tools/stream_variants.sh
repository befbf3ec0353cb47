#!/bin/bash
# Timing-only build of libgd_nn.so WITH the weight-streaming small-map convolution experiment (tools/experimental/nn_conv_stream.h,
# not in the product library) -> ablate/libgd_nn_stream.so.   usage: tools/stream_variants.sh ; then
#   GD_NN_LIB=ablate/libgd_nn_stream.so python tools/stream_conv_bench.py
cd $(dirname $0)/..
mkdir -p ablate
python - <<'PY'
import sys, os, subprocess
sys.path.insert(0, ".")
from garmentdreamer_amd import _build, _build_nn
objs = []
for src, extra in _build_nn.NN_SOURCES:
    if src == "nn_conv3x3.hip":
        o = "ablate/stream_nn_conv3x3.o"
        subprocess.check_call([_build._hipcc()] + _build.COMMON + extra + ["-DGD_NN_EXPERIMENTAL_STREAM", "-Itools/experimental", "-c", os.path.join(_build.CSRC, src), "-o", o])
    else:
        o = os.path.join(_build.CSRC, os.path.splitext(src)[0] + ".o")
    objs.append(o)
subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", "ablate/libgd_nn_stream.so"] + objs)
print("built ablate/libgd_nn_stream.so")
PY
