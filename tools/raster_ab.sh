#!/bin/bash
# Timing of the backward blend (and ablations: 1 = no pass B, 2 = no pass C) on the benchmark rasterizer workload
# (8 views x 512^2, 100k Gaussians).   CFGS="0,1,2,3" tools/raster_ab.sh
IFS=","; for ab in ${CFGS:-0,1,2,3}; do IFS=" "
  GD_RASTER_BWD_ABLATE=$ab timeout 300 python bench.py --raster-only --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d['raster_kernels_ms_per_step']; print('ablate=$ab', ' '.join('%s %.4f' % (k, v) for k, v in r.items()), 'frac %.4f' % d['roofline_raster_bwd']['frac'])
except Exception as e: print('ablate=$ab failed', e)"
done
