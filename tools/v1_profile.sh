#!/bin/bash
# GPU-busy time vs wall time of the 1-view-per-GPU step (what every rank runs at N = 8).  usage: tools/v1_profile.sh <outdir>
out=$1; mkdir -p $out
python bench.py --views 1 --steps 20 --warmup 3 --no-cpu-baseline > $out/v1_line.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_v1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v1 -o bench -- python bench.py --views 1 --steps 10 --warmup 3 --no-cpu-baseline > $out/v1_line_under_rocprof.json 2> /tmp/prof_v1.err
tr=$(find /tmp/prof_v1 -name "*kernel_trace.csv" | head -1)
python tools/steady_stats.py "$tr" $out/v1_kernel_stats_steady.csv --skip 3
python - "$tr" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steady region: last 8 render_backward launches delimit 7 steps
idx = [i for i, r in enumerate(rows) if "render_backward_strip_kernel" in r["Kernel_Name"]]
a, b = idx[-8], idx[-1]
seg = rows[a:b]
wall = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
gaps = []
for p, q in zip(seg[:-1], seg[1:]):
    gaps.append(max(0, int(q["Start_Timestamp"]) - int(p["End_Timestamp"])))
import statistics
print(f"steps 7  wall/step {wall/7e6:.3f} ms  kernel-busy/step {busy/7e6:.3f} ms  kernels/step {len(seg)/7:.0f}  "
      f"median gap {statistics.median(gaps)/1e3:.2f} us  sum gaps/step {sum(gaps)/7e6:.3f} ms  gaps>50us/step {sum(g for g in gaps if g>50000)/7e6:.3f} ms")
P
