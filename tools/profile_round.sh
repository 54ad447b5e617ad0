#!/bin/bash
# Regenerates the per-round evidence under gpurun_out/<tag>/ on the GPU box (copy the summaries into profiles/ afterwards):
#   1. rocprofv3 --kernel-trace --stats over the default bench command   2./3. separate --pmc FETCH_SIZE / WRITE_SIZE passes
# usage (on the box, from the repo root): bash tools/profile_round.sh v8
tag=${1:-vX}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err || exit 1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o p -- python3 $R/tools/pmc_probe.py 16 > $out/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o p -- python3 $R/tools/pmc_probe.py 16 > $out/pmc_write.log 2>&1 || exit 1
cd $R
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
{ python3 tools/pmc_summary.py $out/fetch FETCH_SIZE; python3 tools/pmc_summary.py $out/write WRITE_SIZE; } > $out/pmc_fetch_write.txt
cat $out/bench_under_rocprof.json; head -6 $out/kernel_stats.csv; cat $out/pmc_fetch_write.txt
