#!/bin/bash
# HBM traffic per launch of the two reported kernels, from PMC counters (separate passes, --kernel-trace only):
# FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of a wide coalesced read stream); both are in KiB.
# usage: tools/pmc_traffic.sh <out.json> <bench args...>
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmct_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmct_$c -o pmc -- python bench.py "$@" --no-cpu-baseline --no-graphs > /tmp/pmct_$c.log 2>&1
done
python - "$out" <<'PY'
import csv, glob, json, sys, collections
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmct_{c}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"]
        key = "conv3x3_nhwc_bf16_kernel" if ("conv3x3_nhwc" in k or "conv3x3_gn_patch" in k or "conv3x3_patch_stream" in k) else ("render_backward_kernel" if "render_backward" in k else None)
        if key is None or row["Counter_Name"] != c: continue
        agg[key] += float(row["Counter_Value"]); cnt[key] += 1
    for k in agg:
        res.setdefault(k, {})[c + "_KiB_per_launch_raw"] = agg[k] / cnt[k]
        res[k]["launches_" + c] = cnt[k]
for k, d in res.items():
    fetch = d.get("FETCH_SIZE_KiB_per_launch_raw", 0.0) * 2.0   # gfx950 correction (MI355X_MICROARCH.md, HBM section)
    write = d.get("WRITE_SIZE_KiB_per_launch_raw", 0.0)          # uncalibrated on gfx950, reported as is
    d["hbm_bytes_per_launch"] = (fetch + write) * 1024.0
    d["method"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950), KiB -> bytes"
json.dump(res, open(sys.argv[1], "w"), indent=1)
print(json.dumps(res, indent=1))
PY
