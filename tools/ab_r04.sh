#!/bin/bash
# Same-box A/B of this tree against the round-4 tree (a git worktree of commit 87564f3 built under ablate/r04):
#   tools/ab_r04.sh <outdir> [reps]     interleaved runs of bench.py at 8 / 1 views and --vsd
out=$1; reps=${2:-2}; mkdir -p $out
root=$PWD
for rep in $(seq 1 $reps); do
  for cfg in "v8:" "v1:--views 1" "vsd:--vsd"; do
    name=${cfg%%:*}; args=${cfg#*:}
    python bench.py $args --no-cpu-baseline --steps 20 > $out/new_${name}_$rep.json 2>/dev/null
    (cd ablate/r04 && python bench.py $args --no-cpu-baseline --steps 20 > $root/$out/r04_${name}_$rep.json 2>/dev/null)
  done
done
python - $out $reps <<'PY'
import json, sys
out, reps = sys.argv[1], int(sys.argv[2])
for name in ("v8", "v1", "vsd"):
    for tree in ("r04", "new"):
        ms = []
        for rep in range(1, reps + 1):
            try:
                ms.append(json.loads(open(f"{out}/{tree}_{name}_{rep}.json").read().strip().splitlines()[-1])["ms_per_step"])
            except Exception as e:
                ms.append(float("nan"))
        print(f"{name:4s} {tree:4s} " + " ".join(f"{m:8.3f}" for m in ms) + f"   mean {sum(ms) / len(ms):8.3f} ms")
PY
