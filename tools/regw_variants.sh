#!/bin/bash
# Timing-only builds of libgd_nn.so WITH the filter-bank-in-registers experiment (tools/experimental/nn_conv_regw.h, not in
# the product library since round 5) that differ in nn_conv3x3.hip only:
#   tools/regw_variants.sh "name:-DGD_REGW_ABLATE=1" ...  -> ablate/libgd_nn_<name>.so  (run with GD_NN_LIB=..., tools/ablib.py)
cd $(dirname $0)/..
mkdir -p ablate
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  python - "$name" $flags <<'PY' &
import sys, os, subprocess
sys.path.insert(0, ".")
from garmentdreamer_amd import _build, _build_nn
name, flags = sys.argv[1], sys.argv[2:]
objs = []
for src, extra in _build_nn.NN_SOURCES:
    if src == "nn_conv3x3.hip":
        o = f"ablate/{name}_nn_conv3x3.o"
        subprocess.check_call([_build._hipcc()] + _build.COMMON + extra + flags + ["-DGD_NN_EXPERIMENTAL_REGW", "-Itools/experimental", "-Wno-unused-variable", "-c", os.path.join(_build.CSRC, src), "-o", o])
    else:
        o = os.path.join(_build.CSRC, os.path.splitext(src)[0] + ".o")
    objs.append(o)
subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", f"ablate/libgd_nn_{name}.so"] + objs)
print("built", name)
PY
done
wait
