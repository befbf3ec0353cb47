#!/bin/bash
# One profiling pass of the default bench for profiles/: PMC HBM traffic of the reported kernels FIRST (bench.py quotes
# profiles/rNN_pmc.json only when its kernel-source hash is the tree's, so the counters must exist before the bench
# line is taken), then the bench line, then rocprofv3 kernel stats (whole process and steady state).
#   usage: tools/profile_round.sh <outdir> [rNN]      (rNN: the profiles/ prefix of this round, default r03)
out=$1; rnd=${2:-r03}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/pmc_all.sh $out/pmc.json > /dev/null 2> $out/pmc_passes.log
cp $out/pmc.json profiles/${rnd}_pmc.json
python bench.py > $out/bench_N1_line.json 2> $out/bench_N1.err
rm -rf /tmp/prof_final
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_N1_line_under_rocprof.json 2> /tmp/prof_final.err
st=$(find /tmp/prof_final -name "*kernel_stats.csv" | head -1); tr=$(find /tmp/prof_final -name "*kernel_trace.csv" | head -1)
cp "$st" $out/bench_N1_kernel_stats_whole_process.csv
python tools/steady_stats.py "$tr" $out/bench_N1_kernel_stats_steady.csv --skip 2
ls -la $out
