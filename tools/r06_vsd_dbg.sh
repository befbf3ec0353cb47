#!/bin/bash
out=gpurun_out/r06_vsd_dbg2; mkdir -p $out
for rep in 1 2; do for g in 1 0; do GD_LORA_GROUP=$g python bench.py --vsd --steps 20 --warmup 4 > $out/vsd_g${g}_$rep.json 2>/dev/null; done; done
GD_VSD_DRAIN=0 python bench.py --vsd --steps 20 --warmup 4 > $out/vsd_nodrain.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d["ms_per_step"],3))
    except Exception as e: print(f, "ERR", e)
PY
