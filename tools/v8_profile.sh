#!/bin/bash
# steady-state kernel statistics of the default bench under rocprofv3.  usage: tools/v8_profile.sh <outdir>
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_v8
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v8 -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $out/v8_line_under_rocprof.json 2> /tmp/prof_v8.err
tr=$(find /tmp/prof_v8 -name "*kernel_trace.csv" | head -1)
python tools/steady_stats.py "$tr" $out/v8_kernel_stats_steady.csv --skip 2
