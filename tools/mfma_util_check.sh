#!/bin/bash
# Checks the normalisation of `mfma_util` on a stream whose utilisation is known (tools/probes/mfma_util_probe.hip):
#   mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (n_simds * GRBM_GUI_ACTIVE_per_xcd)
# Writes the raw counters, the kernel durations and the derived figures to $1 (default gpurun_out/mfma_util_check.txt).
out=${1:-gpurun_out/mfma_util_check.txt}; mkdir -p $(dirname $out)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
[ -x /tmp/mfma_util_probe ] || hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_util_probe tools/probes/mfma_util_probe.hip
rm -rf /tmp/mup
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/mup -o p -- /tmp/mfma_util_probe 2 100000 > /tmp/mup.log 2>&1
python - "$out" <<'PY'
import csv, glob, sys, collections
out = open(sys.argv[1], "w")
def P(*a):
    s = " ".join(str(x) for x in a); print(s); out.write(s + "\n")
P(open("/tmp/mup.log").read().strip())
cc = glob.glob("/tmp/mup/**/*counter_collection.csv", recursive=True)
kt = glob.glob("/tmp/mup/**/*kernel_trace.csv", recursive=True)
dur = collections.defaultdict(list)
for r in csv.DictReader(open(kt[0])):
    dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc[0])):
    vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in vals.items():
    d = dur[k][-1] * 1e-9
    mf, gui = c["SQ_VALU_MFMA_BUSY_CYCLES"][-1], c["GRBM_GUI_ACTIVE"][-1]
    P(k[:40], {n: v[-1] for n, v in c.items()}, "duration_us %.1f" % (d * 1e6))
    for xcds in (1, 8):
        P("   if GRBM_GUI_ACTIVE is summed over %d XCD(s): clock %.3f GHz, mfma_util = MFMA_BUSY / (1024 * GUI / %d) = %.3f"
          % (xcds, gui / xcds / d / 1e9, xcds, mf / (1024.0 * gui / xcds)))
    P("   MFMA_BUSY / (1024 SIMDs * duration * 2.4 GHz) = %.3f" % (mf / (1024 * d * 2.4e9)))
PY
