#!/bin/bash
# A/B of the backward-blend implementations / ablations on the benchmark rasterizer workload (8 views x 512^2, 100k).
IFS=","; for cfg in ${CFGS:-lists 0,lists 1,lists 2,lists 3,tree 0}; do IFS=" "
  set -- $cfg
  GD_RASTER_BWD_IMPL=$1 GD_RASTER_BWD_ABLATE=$2 timeout 300 python bench.py --raster-only --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d['raster_kernels_ms_per_step']; print('$1 ablate=$2', 'bwd %.4f ms  fwd %.4f  frac %.4f' % (r['render_bwd'], r['render_fwd'], d['roofline_raster_bwd']['frac']))
except Exception as e: print('$1 $2 failed', e)"
done
