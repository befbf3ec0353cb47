#!/bin/bash
out=gpurun_out/r06_vsd; mkdir -p $out
timeout 900 python -m pytest tests/test_nn_gpu.py -m gpu -x -q -k "lora or flat_adam" > $out/pytest_lora.log 2>&1; tail -3 $out/pytest_lora.log
timeout 1200 python -m pytest tests/test_configs_gpu.py -m gpu -x -q -k "config4" > $out/pytest_vsd.log 2>&1; tail -3 $out/pytest_vsd.log
for rep in 1 2; do
  for g in 0 1; do GD_LORA_GROUP=$g python bench.py --vsd --steps 20 --warmup 4 > $out/vsd_g${g}_$rep.json 2>/dev/null; done
done
GPU_MAX_HW_QUEUES=8 python bench.py --vsd --steps 20 --warmup 4 > $out/vsd_q8.json 2>/dev/null
GD_VSD_DRAIN=0 python bench.py --vsd --steps 20 --warmup 4 > $out/vsd_nodrain.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d["ms_per_step"],3), d["config"].get("kernels_per_step"))
    except Exception as e: print(f, "ERR", e)
PY
