#!/bin/bash
# per-kernel steady-state stats of the one-view-per-GPU step (bench.py --views ${VIEWS:-1}): what the 8-GPU run executes per rank
V=${VIEWS:-1}
python bench.py --views $V --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/v1p
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/v1p -o v -- python bench.py --views $V --steps 20 --warmup 4 --no-cpu-baseline > /tmp/v1p.log 2>&1
tail -1 /tmp/v1p.log | cut -c1-200
tr=$(find /tmp/v1p -name "*kernel_trace.csv" | head -1)
mkdir -p gpurun_out
python tools/steady_stats.py "$tr" gpurun_out/v${V}_steady.csv --skip 4
python - gpurun_out/v${V}_steady.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("kernel ms/step", sum(float(r["MsPerStep"]) for r in rows), "launches/step", sum(float(r["CallsPerStep"]) for r in rows))
for r in rows[:45]:
    print(f'{r["Name"][:80]:82s} {float(r["CallsPerStep"]):6.1f}/step avg {float(r["AverageNs"])/1e3:8.1f} us  {float(r["MsPerStep"]):6.3f} ms/step')
PY
python - "$tr" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "render_backward_block_kernel" in r["Kernel_Name"]]
rows = rows[marks[3]:marks[-1]]; steps = len(marks) - 4
for pat in ("conv3x3_nhwc_bf16_kernel<128, 128", "conv3x3_nhwc_bf16_kernel<256, 256", "conv_splitk", "Cijk"):
    agg = collections.OrderedDict()
    for r in rows:
        if pat in r["Kernel_Name"]:
            k = (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
            d = agg.setdefault(k, [0, 0]); d[0] += 1; d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print(pat)
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("   grid", k, "calls/step %.1f avg %.1f us  %.3f ms/step" % (d[0] / steps, d[1] / d[0] / 1e3, d[1] / steps / 1e6))
PY
