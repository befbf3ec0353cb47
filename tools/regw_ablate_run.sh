#!/bin/bash
# same-box timings of the register-resident-filter convolution under the timing-only builds of tools/regw_variants.sh
python -c "from garmentdreamer_amd import _build_nn; _build_nn.build()" > /dev/null 2>&1
echo "product"; python tools/regw_conv_bench.py 2>&1 | grep "regw" | tail -2
for v in ablate/libgd_nn_*.so; do echo "$v"; GD_NN_LIB=$PWD/$v python tools/regw_conv_bench.py 2>&1 | grep "regw" | tail -2; done
