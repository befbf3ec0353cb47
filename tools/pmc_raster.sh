#!/bin/bash
# SQ counters of the rasterizer's render kernels on the benchmark workload (raster-only, 8 views x 512^2, 100k):
#   tools/pmc_raster.sh <out.txt>   [GD_RASTER_LIB=<other build> is passed through]
out=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > $out
i=0
for g in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmcr_$i
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/pmcr_$i -o pmc -- python bench.py ${GD_RASTER_LIB:+--raster-lib $GD_RASTER_LIB} --raster-only --no-cpu-baseline --steps 3 --warmup 1 > /tmp/pmcr_$i.log 2>&1
  python - /tmp/pmcr_$i >> $out <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        key = "render_backward" if "render_backward" in k else ("render_forward" if "render_forward" in k else None)
        if key is None: continue
        agg[(key, row["Counter_Name"])] += float(row["Counter_Value"]); cnt[(key, row["Counter_Name"])] += 1
for (k, c), v in sorted(agg.items()):
    print(f"{k:18s} {c:24s} {v / cnt[(k, c)]:16.1f} per launch ({cnt[(k, c)]} launches)")
PY
  i=$((i+1))
done
cat $out
