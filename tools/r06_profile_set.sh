#!/bin/bash
# The round-6 profile set -> gpurun_out/r06p/ (copied into profiles/ by hand afterwards).  Run on the GPU box, final tree.
o=gpurun_out/r06p; mkdir -p $o
bash tools/profile_round.sh $o r06 > $o/profile_round.log 2>&1        # pmc (-> profiles/r06_pmc.json), N1 line, N1 stats (whole / steady)
cp profiles/r06_pmc.json $o/pmc.json 2>/dev/null
python tools/util_table.py profiles/r06_pmc.json > $o/util_table.md 2>&1
bash tools/mfma_util_check.sh $o/mfma_util_check.txt > /dev/null 2>&1
python bench.py --views 1 --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 > $o/bench_V1_line.json
python bench.py --views 2 --no-cpu-baseline --steps 20 --warmup 4 2>/dev/null | tail -1 > $o/bench_V2_line.json
python bench.py --views 4 --no-cpu-baseline --steps 20 --warmup 4 2>/dev/null | tail -1 > $o/bench_V4_line.json
python bench.py --views 1 --batch-invariant --simulate-world 8 --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 > $o/bench_V1_batch_invariant_k8_line.json
python bench.py --views 4 --batch-invariant --simulate-world 2 --no-cpu-baseline --steps 20 --warmup 4 2>/dev/null | tail -1 > $o/bench_V4_batch_invariant_k2_line.json
python bench.py --fp8 --no-cpu-baseline --steps 10 --warmup 6 2>/dev/null | tail -1 > $o/bench_N1_fp8_line.json
python bench.py --res 1024 --views 4 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > $o/bench_ref_stage1_line.json
python bench.py --vsd --steps 20 --warmup 4 2>/dev/null | tail -1 > $o/bench_vsd_line.json
python bench.py --vsd --fp8 --res 1024 --steps 10 --warmup 8 2>/dev/null | tail -1 > $o/bench_vsd_fp8_1024_line.json
python bench.py --raster-only --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 > $o/bench_raster_only_line.json
python bench.py --raster-only --views 1 --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 > $o/bench_raster_only_V1_line.json
bash tools/steady_profile.sh $o/bench_V1_kernel_stats_steady.csv --views 1 > $o/bench_V1_steady.txt 2>&1
bash tools/vsd_steady.sh $o/bench_vsd_kernel_stats_steady.csv > $o/bench_vsd_steady.txt 2>&1
bash tools/pmc_raster.sh $o/pmc_raster.txt > /dev/null 2>&1
# this round's routed changes against the round-5 behaviour of the SAME tree (radix binning, GEGLU projections on the library), interleaved
for rep in 1 2 3; do
  GD_RASTER_BUCKETS=0 GD_OWN_GEMM=0 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | tail -1 > $o/ab_old_v8_$rep.json
  python bench.py --no-cpu-baseline --steps 20 2>/dev/null | tail -1 > $o/ab_new_v8_$rep.json
  GD_RASTER_BUCKETS=0 GD_OWN_GEMM=0 python bench.py --views 1 --no-cpu-baseline --steps 30 2>/dev/null | tail -1 > $o/ab_old_v1_$rep.json
  python bench.py --views 1 --no-cpu-baseline --steps 30 2>/dev/null | tail -1 > $o/ab_new_v1_$rep.json
done
bash tools/bench_soak.sh > $o/bench_soak.txt 2>&1
for f in $o/*.json; do python -c "
import json,sys
try:
    d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['ms_per_step'],3), 'ms/step', (d.get('telemetry') or {}).get('clock_mhz_mean'))
except Exception as e: print('$f', 'ERR', e)"; done
tail -2 $o/bench_soak.txt
