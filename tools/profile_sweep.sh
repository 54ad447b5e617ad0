#!/bin/bash
# rocprofv3 kernel trace of whole sweeps of one chain (bench.py --sweep <name>): tools/profile_sweep.sh <name>  -> gpurun_out/prof_sweep/<name>_kernel_stats.csv
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do
  out=$R/gpurun_out/prof_sweep
  mkdir -p $out
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$w -o t -- python3 $R/bench.py --sweep $w > $out/${w}_sweep_under_rocprof.json 2> $out/${w}.err || { echo "trace $w failed"; exit 1; }
  cp $(find $out/trace_$w -name "*kernel_stats.csv" | head -1) $out/${w}_kernel_stats.csv
  rm -rf $out/trace_$w
  head -14 $out/${w}_kernel_stats.csv
done
