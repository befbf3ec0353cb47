#!/bin/bash
# Hardware counters per launch for every hand-written kernel of the step, from rocprofv3 PMC passes of the default
# bench workload (each counter group in its OWN pass, --kernel-trace only -- never together with --stats/sys-trace).
#   usage: tools/pmc_all.sh <out.json> [bench args...]      (run on the GPU box; writes raw CSVs under /tmp)
# HBM bytes: FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section: the counter reports half of a wide coalesced
# read stream), WRITE_SIZE as reported; both are in KiB.
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
groups=("FETCH_SIZE" "WRITE_SIZE" \
        "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
        "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY")
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
KEEP = ("render_backward", "render_forward", "preprocess", "instance_sum", "radix_", "duplicate", "tile_ranges", "conv3x3", "conv_splitk",
        "gn_", "attn_", "geglu", "add_layernorm", "adam", "activate", "sds_", "vae_prologue", "sparsity", "gemm_", "xattn")
def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("gd::", "")
    n = re.sub(r"^(void\s+)?", "", n.strip())
    return n.split("(")[0].strip()
res = collections.OrderedDict()
for d in sorted(glob.glob("/tmp/pmca_*")):
    if not d[len("/tmp/pmca_"):].isdigit():
        continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(float); cnt = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if not any(s in k for s in KEEP):
                continue
            c = row["Counter_Name"]
            agg[(k, c)] += float(row["Counter_Value"]); cnt[(k, c)] += 1
        for (k, c), v in agg.items():
            e = res.setdefault(k, collections.OrderedDict())
            e[c] = v / cnt[(k, c)]
            e["launches_sampled"] = max(e.get("launches_sampled", 0), cnt[(k, c)])
for k, e in res.items():
    if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = (2.0 * e.get("FETCH_SIZE", 0.0) + e.get("WRITE_SIZE", 0.0)) * 1024.0
    if "SQ_INSTS_VALU" in e and "SQ_BUSY_CYCLES" in e and e["SQ_BUSY_CYCLES"] > 0:
        # a wave64 VALU instruction holds its SIMD ~4.5 cycles (tools/probes/valu_rate_probe.hip); SQ_BUSY_CYCLES is summed
        # over the shader engines' SQ instances, so the per-launch figure is reported raw, the utilisation derived in DESIGN.md
        e["valu_insts_per_launch"] = e["SQ_INSTS_VALU"]
meta = {"kernel_source_hash": bench.kernel_source_hash(),
        "workload": {"gaussians": _w.gaussians, "views": _w.views, "res": _w.res},
        "method": ("rocprofv3 --kernel-trace --pmc <group> in separate passes of `python bench.py --steps 2 --warmup 1 "
                   "--no-graphs`; values are averages per launch over the sampled launches; FETCH_SIZE x2 (gfx950), KiB"),
        "kernels": res}
json.dump(meta, open(_out, "w"), indent=1)
print(json.dumps({k: {c: v for c, v in e.items() if c in ("hbm_bytes_per_launch", "SQ_INSTS_VALU", "launches_sampled")}
                  for k, e in res.items()}, indent=1))
PY
