#!/bin/bash
# Round 6: tile-bucketed binning against the radix path: parity suite, per-kernel times (HIP events, bench.py --raster-only),
# kernel trace of both
out=gpurun_out/r06_sort; mkdir -p $out
echo skip-pytest
i=0
for b in 1 0 1 0; do i=$((i+1)); GD_RASTER_BUCKETS=$b python bench.py --raster-only --no-cpu-baseline --steps 50 > $out/raster_b${b}_$i.json 2>/dev/null; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/raster_b*_?.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d["ms_per_step"],3), {k:round(v,4) for k,v in d["raster_kernels_ms_per_step"].items()})
PY
cd /tmp && export TMPDIR=/tmp
for b in 1 0; do
  rm -rf /tmp/prof_b$b
  GD_RASTER_BUCKETS=$b rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b$b -o t -- python $GRAFT_REPO_ROOT/bench.py --raster-only --no-cpu-baseline --steps 20 > /dev/null 2>&1
  f=$(find /tmp/prof_b$b -name "*kernel_stats.csv" | head -1)
  cp "$f" $GRAFT_REPO_ROOT/$out/kernel_stats_b$b.csv
  echo "== buckets=$b"; head -16 "$f" | cut -d, -f1-6
done
