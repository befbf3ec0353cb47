#!/bin/bash
out=gpurun_out/r06_smalln; mkdir -p $out
for rep in 1 2; do for c in 0 1; do
  GD_GN_STATS_SMALLN=$c python bench.py --views 1 --no-cpu-baseline --steps 30 > $out/v1_s${c}_$rep.json 2>/dev/null
  GD_GN_STATS_SMALLN=$c python bench.py --vsd --steps 20 > $out/vsd_s${c}_$rep.json 2>/dev/null
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d["ms_per_step"],3))
PY
