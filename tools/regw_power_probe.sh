#!/bin/bash
# Clock / power while the register-resident-filter convolution (product form and its MFMA-only timing build) and the wide
# tile run back to back on 8 x 128 -> 128 @ 512^2:  tools/regw_power_probe.sh   (builds ablate/libgd_nn_m53.so itself)
cd $GRAFT_REPO_ROOT
python -c "from garmentdreamer_amd import _build_nn; _build_nn.build()" > /dev/null 2>&1
bash tools/regw_variants.sh "m53:-DGD_REGW_ABLATE=53" > /dev/null 2>&1
cat > /tmp/regw_loop.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
import tools.ablib  # noqa
from garmentdreamer_amd import nn_ops
which, n = sys.argv[1], int(sys.argv[2])
C = 128
x = torch.randn(8, C, 512, 512, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn(C, C, 3, 3, device="cuda") / 34).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
b = torch.randn(C, device="cuda").to(torch.bfloat16)
f = nn_ops._regw_launch if which == "regw" else nn_ops._wide_launch
with torch.no_grad():
    for i in range(n):
        f(x, w, b, None, C)
        if i % 200 == 199:
            torch.cuda.synchronize()
torch.cuda.synchronize()
PY
probe() {
  python -u /tmp/regw_loop.py $1 3 > /dev/null 2>&1
  python -u /tmp/regw_loop.py $1 30000 > /dev/null 2>&1 &
  pid=$!
  sleep 8
  for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket" | sed 's/.*: (\([0-9]*Mhz\)).*/sclk \1/; s/.*Power (W): /power W /' | tr '\n' ' '; echo; sleep 0.5; done
  kill $pid 2>/dev/null; wait $pid 2>/dev/null
}
echo "wide tile (product)"; probe wide
echo "register-resident filter (product form)"; probe regw
echo "register-resident filter, MFMAs + exchange + barrier only (timing build)"; GD_NN_LIB=$PWD/ablate/libgd_nn_m53.so probe regw
