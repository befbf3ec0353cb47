#!/bin/bash
# Timing-only builds of csrc/nn_gemm.hip alone:  tools/gemm_variants.sh "name:-DGD_GEMM_ABLATE=1" ...
# -> tools/variants/libgd_gemm_<name>.so (git-ignored, travels to the GPU box); tools/gemm_own_bench.py reads GD_GEMM_LIB.
cd $(dirname $0)/..
mkdir -p tools/variants
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable $flags -shared \
    garmentdreamer_amd/csrc/nn_gemm.hip -o tools/variants/libgd_gemm_$name.so && echo built $name
done
