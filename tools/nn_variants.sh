#!/bin/bash
# Timing-only builds of libgd_nn.so with -D switches (csrc/nn_*.hip):  tools/nn_variants.sh "name:-DFLAG=1" ...
# -> ablate/libgd_nn_<name>.so; run with GD_NN_LIB=$PWD/ablate/libgd_nn_<name>.so (tools/nn_ab.sh).
cd $(dirname $0)/..
mkdir -p ablate
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  python - "$name" $flags <<'PY'
import sys, os, subprocess
sys.path.insert(0, ".")
from garmentdreamer_amd import _build, _build_nn
name, flags = sys.argv[1], sys.argv[2:]
objs = []
for src, extra in _build_nn.NN_SOURCES:
    o = f"ablate/{name}_{os.path.splitext(src)[0]}.o"
    subprocess.check_call([_build._hipcc()] + _build.COMMON + extra + flags + ["-Wno-unused-variable", "-c", os.path.join(_build.CSRC, src), "-o", o])
    objs.append(o)
subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", f"ablate/libgd_nn_{name}.so"] + objs)
print("built", name)
PY
done
