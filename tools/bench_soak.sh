#!/bin/bash
# Sixteen default-size bench runs back to back: step times and the count of runs the health check refused (non-finite parameters).
cd $(dirname $0)/..
ok=0; bad=0
for i in $(seq 1 16); do
  out=$(python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1)
  if echo "$out" | grep -q '"ms_per_step"'; then ok=$((ok+1)); echo "$out" | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'], end=' ')"; else bad=$((bad+1)); echo; echo "FAIL: $out" | cut -c1-200; fi
done
echo; echo "ok=$ok bad=$bad"
