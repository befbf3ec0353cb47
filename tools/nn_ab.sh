#!/bin/bash
# Same-box A/B of two libgd_nn builds on the default bench:  tools/nn_ab.sh <other.so> [bench args]   (interleaved runs)
cd $(dirname $0)/..
other=$(realpath $1); shift
for i in 1 2; do
  for so in "$other" ""; do
    python bench.py ${so:+--nn-lib $so} --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('${so:-product}'.split('/')[-1], 'ms/step %.3f' % d['ms_per_step'])"
  done
done
