#!/bin/bash
# Secondary measurements quoted in DESIGN.md / BASELINE.md (round 4) -> gpurun_out/r04x/
o=gpurun_out/r04x; mkdir -p $o
python bench.py --views 1 --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 > $o/bench_V1_line.json
python bench.py --views 2 --no-cpu-baseline --steps 20 --warmup 4 2>/dev/null | tail -1 > $o/bench_V2_line.json
python bench.py --views 4 --no-cpu-baseline --steps 20 --warmup 4 2>/dev/null | tail -1 > $o/bench_V4_line.json
python bench.py --fp8 --no-cpu-baseline --steps 10 --warmup 6 2>/dev/null | tail -1 > $o/bench_N1_fp8_line.json
python bench.py --res 1024 --views 4 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > $o/bench_ref_stage1_line.json
python bench.py --vsd --steps 10 --warmup 4 2>/dev/null | tail -1 > $o/bench_vsd_line.json
python bench.py --vsd --fp8 --res 1024 --steps 10 --warmup 8 2>/dev/null | tail -1 > $o/bench_vsd_fp8_1024_line.json
python bench.py --raster-only --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 > $o/bench_raster_only_line.json
python bench.py --raster-only --views 1 --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 > $o/bench_raster_only_V1_line.json
python tools/densify_time.py > $o/densify_time.txt 2>&1
bash tools/steady_profile.sh $o/bench_V1_kernel_stats_steady.csv --views 1 > $o/bench_V1_steady.txt 2>&1
bash tools/pmc_raster.sh $o/pmc_raster.txt > /dev/null 2>&1
for f in $o/*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['ms_per_step'],3), 'ms/step', d.get('raster_kernels_ms_per_step'))"; done
cat $o/densify_time.txt
