#!/bin/bash
# Same-box A/B of rasterizer builds on the benchmark rasterizer workload (8 views x 512^2, 100k Gaussians).
#   tools/raster_ab.sh build   (in the build container: ablate/libgd_raster_<name>.so for each variant below)
#   tools/raster_ab.sh         (on the GPU box: times every ablate/*.so and the product library)
# Variants are -D switches of csrc/*.hip that exist for timing only (wrong results).
cd $(dirname $0)/..
if [ "$1" == "build" ]; then
  mkdir -p ablate
  for v in ${VARIANTS:-"cut0:-DGD_BWD_CUT=0" "cut8:-DGD_BWD_CUT=8"}; do
    name=${v%%:*}; flags=${v#*:}
    python - "$name" $flags <<'PY'
import sys, os, subprocess
sys.path.insert(0, ".")
from garmentdreamer_amd import _build
name, flags = sys.argv[1], sys.argv[2:]
objs = []
for src, extra in _build.RASTER_SOURCES:
    o = f"ablate/{name}_{os.path.splitext(src)[0]}.o"
    subprocess.check_call([_build._hipcc()] + _build.COMMON + extra + flags + ["-c", os.path.join(_build.CSRC, src), "-o", o])
    objs.append(o)
subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", f"ablate/libgd_raster_{name}.so"] + objs)
print("built", name)
PY
  done
  exit 0
fi
for so in "" ablate/libgd_raster_*.so; do
  timeout 300 python bench.py ${so:+--raster-lib $PWD/$so} --raster-only --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d['raster_kernels_ms_per_step']; print('${so:-product}', ' '.join('%s %.4f' % (k, v) for k, v in r.items()), 'frac %.4f' % d['roofline_raster_bwd']['frac'])
except Exception as e: print('${so:-product} failed', e)"
done
