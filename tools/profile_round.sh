#!/bin/bash
# One profiling pass of the default bench for profiles/: bench line, rocprofv3 kernel stats (whole process and
# steady state), PMC HBM traffic of the two reported kernels.   usage: tools/profile_round.sh <outdir>
out=$1; mkdir -p $out
python bench.py > $out/bench_N1_line.json 2> $out/bench_N1.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_final
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_N1_line_under_rocprof.json 2> /tmp/prof_final.err
st=$(find /tmp/prof_final -name "*kernel_stats.csv" | head -1); tr=$(find /tmp/prof_final -name "*kernel_trace.csv" | head -1)
cp "$st" $out/bench_N1_kernel_stats_whole_process.csv
python tools/steady_stats.py "$tr" $out/bench_N1_kernel_stats_steady.csv --skip 2
bash tools/pmc_all.sh $out/pmc.json > /dev/null 2> $out/pmc_passes.log
ls -la $out
