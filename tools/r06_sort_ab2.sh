#!/bin/bash
out=gpurun_out/r06_sort4; mkdir -p $out
timeout 600 python -m pytest tests/test_raster_gpu.py tests/test_scene_gpu.py -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
i=0
for b in 1 0 1 0; do i=$((i+1)); GD_RASTER_BUCKETS=$b python bench.py --raster-only --no-cpu-baseline --steps 50 > $out/raster_b${b}_$i.json 2>/dev/null; done
python -m pytest tests -m gpu -x -q > $out/pytest_all.log 2>&1; tail -3 $out/pytest_all.log
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/raster_b*_?.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d["ms_per_step"],3), {k:round(v,4) for k,v in d["raster_kernels_ms_per_step"].items()})
for f in sorted(glob.glob("$out/v8_bkt*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d["ms_per_step"],3))
PY
