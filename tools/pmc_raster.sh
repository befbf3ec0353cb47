#!/bin/bash
# PMC counters for the rasterizer render kernels (separate pass from timing; --pmc with --kernel-trace only).
# usage: tools/pmc_raster.sh <outdir> [extra env like GD_RASTER_BWD_PPL=4]
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  env "$@" rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_$tag -o pmc -- python bench.py --raster-only --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pmc_$tag.log 2>&1
  python - "$tag" "$out" <<'PY'
import csv, sys, glob, collections
tag, out = sys.argv[1], sys.argv[2]
f = glob.glob(f"/tmp/pmc_{tag}/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"]
    if "render_" not in k: continue
    k = "render_backward" if "backward" in k else "render_forward"
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
    cnt[(k,row["Counter_Name"])] += 1
with open(f"{out}/pmc_{tag}.txt","w") as o:
    for k in agg:
        for c,v in agg[k].items():
            line=f"{k} {c} per_launch {v/cnt[(k,c)]:.4g}"
            print(line); o.write(line+"\n")
PY
done
