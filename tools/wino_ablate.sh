#!/bin/bash
# Ablation table of the Winograd convolution kernel (csrc/nn_conv_wino.h, -DGD_WINO_ABLATE=n builds; timing only).
cd $(dirname $0)/..
V=""
for n in 1 2 3 4 5 6 7; do V="$V w$n:-DGD_WINO_ABLATE=$n"; done
tools/nn_variants.sh $V > /dev/null
echo "product"; python tools/wino_conv_bench.py abl
for n in 1 2 3 4 5 6 7; do echo "ablate $n"; GD_NN_LIB=$PWD/ablate/libgd_nn_w$n.so python tools/wino_conv_bench.py abl; done
