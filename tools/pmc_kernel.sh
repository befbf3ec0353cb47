#!/bin/bash
# usage: tools/pmc_kernel.sh <kernel-substring> <cmd...>   -> prints per-launch averages of PMC counters
pat=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rm -rf /tmp/pmck_$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmck_$i -o pmc -- "$@" > /tmp/pmck_$i.log 2>&1
  python - "$i" "$pat" <<'PY'
import csv, sys, glob, collections
i, pat = sys.argv[1], sys.argv[2]
f = glob.glob(f"/tmp/pmck_{i}/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file; log tail:"); print(open(f"/tmp/pmck_{i}.log").read()[-600:]); sys.exit(0)
agg = collections.defaultdict(float); cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
    if pat not in row["Kernel_Name"]: continue
    agg[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
for c, v in agg.items():
    print(f"{c} per_launch {v/cnt[c]:.4g}  (launches {cnt[c]})")
PY
done
