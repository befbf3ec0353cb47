#!/bin/bash
# where the VSD iteration (bench.py --vsd) leaves the GPU idle: gaps between consecutive kernels of the steady iterations,
# largest first, with the kernels either side (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/vsdg
rocprofv3 --kernel-trace --output-format csv -d /tmp/vsdg -o v -- python bench.py ${BENCH_ARGS:---vsd} --steps 8 --warmup 4 > /tmp/vsdg.log 2>&1
tail -1 /tmp/vsdg.log | cut -c100-230
tr=$(find /tmp/vsdg -name "*kernel_trace.csv" | head -1)
python - "$tr" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
rows = rows[int(n * 0.45):int(n * 0.9)]   # steady iterations (capture steps before, health checks after)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
gaps = []
end = int(rows[0]["End_Timestamp"])
for i in range(1, len(rows)):
    s = int(rows[i]["Start_Timestamp"])
    if s > end:
        gaps.append((s - end, i))
    end = max(end, int(rows[i]["End_Timestamp"]))
tot = sum(g for g, _ in gaps)
print(f"span {span/1e6:.2f} ms, kernel time {busy/1e6:.2f} ms, idle {tot/1e6:.2f} ms in {len(gaps)} gaps")
for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e12)):
    sel = [g for g, _ in gaps if lo <= g < hi]
    print(f"  gaps {lo/1e3:6.0f}-{hi/1e3:.0f} us: {len(sel):6d}  sum {sum(sel)/1e6:.3f} ms")
short = lambda k: k.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
for g, i in sorted(gaps, reverse=True)[:40]:
    print(f"{g/1e3:8.1f} us  after {short(rows[i-1]['Kernel_Name'])}  -> before {short(rows[i]['Kernel_Name'])}")
PY
