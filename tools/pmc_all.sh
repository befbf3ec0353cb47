#!/bin/bash
# Hardware counters per launch for every hand-written kernel of the step, from rocprofv3 PMC passes of the default
# bench workload (each counter group in its OWN pass, --kernel-trace only -- never together with --stats/sys-trace).
#   usage: tools/pmc_all.sh <out.json> [bench args...]      (run on the GPU box; writes raw CSVs under /tmp)
# HBM bytes: FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section: the counter reports half of a wide coalesced
# read stream), WRITE_SIZE as reported; both are in KiB.
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# GRBM_GUI_ACTIVE rides in every pass (the GRBM block has its own two slots): clock held by a kernel = GUI / 8 XCDs / duration, and
# the denominator of mfma_util (tools/mfma_util_check.sh: 0.976 on an MFMA-only stream, profiles/r05_mfma_util_check.txt)
groups=("FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE" \
        "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
        "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE")
i=0
for g in "${groups[@]}"; do
  rm -rf /tmp/pmca_$i
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/pmca_$i -o pmc -- python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-graphs > /tmp/pmca_$i.log 2>&1
  echo "pass $i ($g): rc $?" >&2
  i=$((i+1))
done
python - "$out" "$@" <<'PY'
import argparse, csv, glob, json, os, re, sys, collections
sys.path.insert(0, os.getcwd())
import bench                      # kernel_source_hash(): the counter file is only quoted for the sources it measured
_out = sys.argv[1]
_a = argparse.ArgumentParser()
for _n, _d in (("--gaussians", 100000), ("--views", 8), ("--res", 512)):
    _a.add_argument(_n, type=int, default=_d)
_w, _ = _a.parse_known_args(sys.argv[2:])
N_SIMDS, N_XCDS = 1024.0, 8.0      # MI355X: 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE is reported summed over the 8 XCDs
KEEP = ("render_backward", "render_forward", "preprocess", "instance_sum", "radix_", "duplicate", "tile_ranges", "tile_count", "tile_scan", "tile_scatter", "tile_sort", "gemm256", "conv3x3", "conv_splitk",
        "gn_", "attn_", "geglu", "add_layernorm", "adam", "activate", "sds_", "vae_prologue", "sparsity", "gemm_", "xattn")
def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("gd::", "")
    n = re.sub(r"^(void\s+)?", "", n.strip())
    return n.split("(")[0].strip()
res = collections.OrderedDict()
dur = collections.defaultdict(list)          # kernel -> launch durations (ns) under the counter passes' kernel traces
gui_mfma = {}                                 # kernel -> GRBM_GUI_ACTIVE of the pass that also counted the MFMA cycles
for d in sorted(glob.glob("/tmp/pmca_*")):
    if not d[len("/tmp/pmca_"):].isdigit():
        continue
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if any(s in k for s in KEEP):
                dur[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(float); cnt = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if not any(s in k for s in KEEP):
                continue
            c = row["Counter_Name"]
            agg[(k, c)] += float(row["Counter_Value"]); cnt[(k, c)] += 1
        has_mfma = any(c == "SQ_VALU_MFMA_BUSY_CYCLES" for (_, c) in agg)
        for (k, c), v in agg.items():
            e = res.setdefault(k, collections.OrderedDict())
            if c == "GRBM_GUI_ACTIVE":
                e.setdefault("_gui", []).append(v / cnt[(k, c)])
                if has_mfma:
                    gui_mfma[k] = v / cnt[(k, c)]
                continue
            e[c] = v / cnt[(k, c)]
            e["launches_sampled"] = max(e.get("launches_sampled", 0), cnt[(k, c)])
for k, e in res.items():
    if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = (2.0 * e.get("FETCH_SIZE", 0.0) + e.get("WRITE_SIZE", 0.0)) * 1024.0
    gui = e.pop("_gui", None)
    if gui:
        e["GRBM_GUI_ACTIVE"] = sum(gui) / len(gui)
    if dur.get(k):
        e["avg_duration_us_profiled"] = sum(dur[k]) / len(dur[k]) / 1e3
        if gui:      # clock the kernel held under the counter passes (MI355X_MICROARCH.md, DVFS): busy cycles per XCD / wall time
            e["clock_ghz"] = e["GRBM_GUI_ACTIVE"] / N_XCDS / (e["avg_duration_us_profiled"] * 1e3)
        if "hbm_bytes_per_launch" in e:
            e["hbm_gbps"] = e["hbm_bytes_per_launch"] / (e["avg_duration_us_profiled"] * 1e3)
    if e.get("SQ_VALU_MFMA_BUSY_CYCLES") and k in gui_mfma and gui_mfma[k] > 0:
        # share of the launch's SIMD-cycles in which the matrix pipe was busy: the counter is summed over the 1024 SIMDs, the
        # GUI count over the 8 XCDs; 0.976 on tools/probes/mfma_util_probe.hip's MFMA-only stream, 0.58 with half the MFMAs
        # replaced by VALU filler (profiles/r05_mfma_util_check.txt)
        e["mfma_util"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMDS * gui_mfma[k] / N_XCDS)
    if "SQ_INSTS_VALU" in e and "SQ_BUSY_CYCLES" in e and e["SQ_BUSY_CYCLES"] > 0:
        # a wave64 VALU instruction holds its SIMD ~4.5 cycles (tools/probes/valu_rate_probe.hip); SQ_BUSY_CYCLES is summed
        # over the shader engines' SQ instances, so the per-launch figure is reported raw, the utilisation derived in DESIGN.md
        e["valu_insts_per_launch"] = e["SQ_INSTS_VALU"]
meta = {"kernel_source_hash": bench.kernel_source_hash(),
        "workload": {"gaussians": _w.gaussians, "views": _w.views, "res": _w.res},
        "method": ("rocprofv3 --kernel-trace --pmc <group> in separate passes of `python bench.py --steps 2 --warmup 1 "
                   "--no-graphs`; values are averages per launch over the sampled launches; FETCH_SIZE x2 (gfx950), KiB; "
                   "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), hbm_gbps = hbm_bytes_per_launch / "
                   "avg_duration_us_profiled, clock_ghz = GRBM_GUI_ACTIVE / 8 / duration (all under the profiler)"),
        "kernels": res}
json.dump(meta, open(_out, "w"), indent=1)
print(json.dumps({k: {c: v for c, v in e.items() if c in ("hbm_bytes_per_launch", "hbm_gbps", "mfma_util", "clock_ghz", "launches_sampled")}
                  for k, e in res.items()}, indent=1))
PY
