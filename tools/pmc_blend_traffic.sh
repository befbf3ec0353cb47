cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for g in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pq; rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/pq -o p -- python bench.py --raster-only --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
  python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob("/tmp/pq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        key = "bwd" if "render_backward" in k else ("fwd" if "render_forward" in k else None)
        if key: agg[(key, row["Counter_Name"])] += float(row["Counter_Value"]); cnt[(key, row["Counter_Name"])] += 1
for (k, c), v in sorted(agg.items()): print(k, c, "%.0f KiB per launch" % (v / cnt[(k, c)]))
PY
done
