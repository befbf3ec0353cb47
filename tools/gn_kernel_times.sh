cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/gnp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gnp -o g -- python tools/gn_bw_bench.py > /tmp/gnp.log 2>&1
tr=$(find /tmp/gnp -name "*kernel_trace.csv" | head -1)
python - "$tr" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "gn_" not in n: continue
    k = (n.replace("(anonymous namespace)::", "").split("(")[0][-24:], r["Grid_Size_X"], r["Grid_Size_Y"])
    d = agg.setdefault(k, []); d.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in agg.items():
    d.sort(); print(k, "n", len(d), "median %.1f us" % (d[len(d)//2] / 1e3))
PY
