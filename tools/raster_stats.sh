#!/bin/bash
# rocprofv3 per-kernel stats of the raster-only benchmark (8 views x 512^2, 100k Gaussians)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/rstats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rstats -o r -- python bench.py --raster-only --no-cpu-baseline --steps 20 --warmup 3 > /tmp/rstats.log 2>&1
f=$(find /tmp/rstats -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("gd::(anonymous namespace)::", "").split("(")[0][:60]
    if float(r["Percentage"]) > 0.3:
        print(f'{n:62s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  {float(r["Percentage"]):5.1f} %')
PY
