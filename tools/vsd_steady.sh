#!/bin/bash
# steady-state per-kernel stats of the VSD iteration (bench.py --vsd): rocprofv3 kernel trace, warm-up / capture steps cut off
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/vsdp
rocprofv3 --kernel-trace --output-format csv -d /tmp/vsdp -o v -- python bench.py --vsd --steps 8 --warmup 4 > /tmp/vsdp.log 2>&1
tail -1 /tmp/vsdp.log | cut -c100-200
tr=$(find /tmp/vsdp -name "*kernel_trace.csv" | head -1)
python - "$tr" ${1:-gpurun_out/vsd_steady.csv} <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steady part: the last 6 iterations; an iteration ends with AdamW's multi-tensor kernel -> use time: last 60 % of the span is safe
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
# iteration marks: the first optimizer kernel of each step (gd::adam_kernel since round 5, torch's multi-tensor kernel before;
# successive ones > 5 ms apart)
marks, last = [], -1e18
for i, r in enumerate(rows):
    if "multi_tensor_apply" in r["Kernel_Name"] or "adam_kernel" in r["Kernel_Name"]:
        t = int(r["Start_Timestamp"])
        if t - last > 5e6:
            marks.append(i)
        last = t
steps = min(6, len(marks) - 1)
rows = rows[marks[-1 - steps]:marks[-1]]
agg = collections.OrderedDict()
for r in rows:
    d = agg.setdefault(r["Kernel_Name"], [0, 0]); d[0] += 1; d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(d[1] for d in agg.values())
print(f"steady: {tot / steps / 1e6:.2f} ms of kernel time per iteration, {sum(d[0] for d in agg.values()) / steps:.0f} launches")
with open(sys.argv[2], "w") as f:
    f.write("Name,CallsPerStep,AverageNs,MsPerStep\n")
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write(f'"{k}",{d[0] / steps},{d[1] / d[0]},{d[1] / steps / 1e6}\n')
for k, d in sorted(agg.items(), key=lambda kv: -kv[1][1])[:32]:
    print(f"{d[1] / steps / 1e6:7.3f} ms {d[0] / steps:7.1f}/it {d[1] / d[0] / 1e3:8.1f} us  {k[:110]}")
PY
