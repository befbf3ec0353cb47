#!/bin/bash
# Same-box A/B of the GroupNorm statistics from the convolution epilogues: default bench with GD_NN_EPILOGUE_STATS=0 / 1, interleaved.
cd $(dirname $0)/..
for i in 1 2 3; do
  for f in 0 1; do
    GD_NN_EPILOGUE_STATS=$f python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('epilogue_stats=$f', 'ms/step %.3f' % d['ms_per_step'])"
  done
done
