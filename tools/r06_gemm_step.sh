#!/bin/bash
# Round 6: own GEMM in the step -- suite, per-shape table, same-box A/B of the routed GEGLU projections (GD_OWN_GEMM=0/1)
out=gpurun_out/r06_gemm_step; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
python tools/gemm_own_bench.py 8 > $out/gemm_shapes.txt 2>&1
for rep in 1 2; do
  for on in 0 1; do
    GD_OWN_GEMM=$on python bench.py --no-cpu-baseline --steps 20 > $out/v8_gemm${on}_$rep.json 2>/dev/null
    GD_OWN_GEMM=$on python bench.py --views 1 --no-cpu-baseline --steps 30 > $out/v1_gemm${on}_$rep.json 2>/dev/null
  done
done
python bench.py --views 1 --batch-invariant --simulate-world 8 --no-cpu-baseline --steps 30 > $out/v1_bi8.json 2>/dev/null
python - $out <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f"{os.path.basename(f):24s} {d['ms_per_step']:8.3f} ms")
    except Exception as e:
        print(f, "unreadable", e)
PY
cat $out/gemm_shapes.txt
