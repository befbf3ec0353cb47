#!/bin/bash
# Round 6, first GPU call: suite, then the zero-cost half-batch measurement the round-5 review asked for
# (two ranks x 4 views sharing ONE GPU under gloo against one rank x 8 views, same box, interleaved), and what
# batch-invariant routing really costs a rank with 1/K of the views (bench.py --simulate-world K).
out=gpurun_out/r06_first; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log; tail -3 $out/pytest.log
for rep in 1 2; do
  python bench.py --no-cpu-baseline --steps 20 > $out/v8_$rep.json 2>$out/v8_$rep.err
  GD_DIST_BACKEND=gloo python bench.py --gpus 2 --no-cpu-baseline --steps 20 > $out/half2x4_$rep.json 2>$out/half2x4_$rep.err
done
python bench.py --views 1 --no-cpu-baseline --steps 30 > $out/v1.json 2>$out/v1.err
python bench.py --views 1 --batch-invariant --simulate-world 8 --no-cpu-baseline --steps 30 > $out/v1_bi8.json 2>$out/v1_bi8.err
python bench.py --views 4 --no-cpu-baseline --steps 20 > $out/v4.json 2>$out/v4.err
python bench.py --views 4 --batch-invariant --simulate-world 2 --no-cpu-baseline --steps 20 > $out/v4_bi2.json 2>$out/v4_bi2.err
python - $out <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f"{os.path.basename(f):20s} {d['ms_per_step']:8.3f} ms  {d['value']:7.3f} it/s  ranks={d.get('rccl_ranks')} bi={d['config'].get('batch_invariant')} k={d['config'].get('batch_invariant_route_scale')}")
    except Exception as e:
        print(f, "unreadable", e)
PY
