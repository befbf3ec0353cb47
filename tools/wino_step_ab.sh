#!/bin/bash
# Same-box A/B of the whole step with / without the Winograd routing (GD_NN_WINO=0/1), interleaved.
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for w in 0 1; do
    GD_NN_WINO=$w python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('GD_NN_WINO=$w', round(d['ms_per_step'],2), 'ms  conv frac', round(d['roofline']['frac'],3), 'conv ms', round(d['roofline']['ms_per_step'],2))"
  done
done
