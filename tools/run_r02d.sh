#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r02d
mkdir -p $out
cd $R
timeout -k 10 500 python bench.py --steps 5 --warmup 2 > $out/bench_default.json 2> $out/bench_default.err || echo "bench failed"
cat $out/bench_default.json; tail -3 $out/bench_default.err
timeout -k 10 300 python bench.py --workload cr2_m1000 --steps 10 --warmup 2 --no-cpu > $out/bench_m1000.json 2> $out/bench_m1000.err || echo "bench failed"
cat $out/bench_m1000.json
