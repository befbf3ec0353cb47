#!/bin/bash
# rocprofv3 steady-state kernel stats of the default bench -> <out.csv>, prints the top kernels.  usage: tools/steady_profile.sh <out.csv> [bench args]
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_sp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_sp -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > /tmp/prof_sp.log 2>&1
tr=$(find /tmp/prof_sp -name "*kernel_trace.csv" | head -1)
python tools/steady_stats.py "$tr" $out --skip 2
python - $out <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:45]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("gd::", "")
    n = n.split("(")[0][:70] if not n.startswith("Cijk") and not n.startswith("Custom") else n[:40] + ".." + n[n.find("MT"):n.find("MT") + 14]
    print(f'{n:72s} {float(r["CallsPerStep"]):6.1f}/step avg {float(r["AverageNs"])/1e3:8.1f} us  {float(r["MsPerStep"]):6.3f} ms/step')
PY
