#!/bin/bash
# per-kernel stats of the VSD iteration (bench.py --vsd), steady part
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/vsdp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vsdp -o v -- python bench.py --vsd --steps 6 --warmup 3 > /tmp/vsdp.log 2>&1
tail -1 /tmp/vsdp.log | cut -c1-300
f=$(find /tmp/vsdp -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print(f"total kernel time {tot/1e6/9:.1f} ms per step (9 steps incl. warm-up), {calls/9:.0f} launches per step")
for r in rows[:22]:
    print(f'{r["Name"][:70]:72s} calls {int(r["Calls"])/9:7.1f}/step avg {float(r["AverageNs"])/1e3:8.1f} us {float(r["Percentage"]):5.1f} %')
PY
