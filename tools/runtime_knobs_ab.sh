#!/bin/bash
# Same-box A/B of ROCclr runtime knobs on the launch-bound steps (VSD iteration, 1-view SDS step): ms per step per setting, twice.
#   usage (GPU box): tools/runtime_knobs_ab.sh > gpurun_out/knobs.txt
cd $GRAFT_REPO_ROOT
run() {  # label, env assignments...
  local label=$1; shift
  for rep in 1 2; do
    v=$(env "$@" python bench.py --vsd --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))")
    s=$(env "$@" python bench.py --views 1 --steps 60 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))")
    echo "$label: vsd $v ms, 1-view $s ms"
  done
}
run "default (no torch syncs)" X=1
run "GD_PAGEABLE_COPIES=1 (3 / 1 syncs)" GD_PAGEABLE_COPIES=1
run "HSA_KERNARG_POOL_SIZE=16M" HSA_KERNARG_POOL_SIZE=16777216
run "ROC_SIGNAL_POOL_SIZE=4096" ROC_SIGNAL_POOL_SIZE=4096
run "ROC_AQL_QUEUE_SIZE=65536" ROC_AQL_QUEUE_SIZE=65536
run "ROC_ACTIVE_WAIT_TIMEOUT=1000" ROC_ACTIVE_WAIT_TIMEOUT=1000
run "kernarg 16M + signals 4096 + aql 65536" HSA_KERNARG_POOL_SIZE=16777216 ROC_SIGNAL_POOL_SIZE=4096 ROC_AQL_QUEUE_SIZE=65536
run "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
run "default again" X=1
