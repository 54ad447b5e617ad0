#!/bin/bash
# Per-round evidence (ROUND=r03 by default), one directory per workload under gpurun_out/prof_$ROUND/ (copy the summaries into profiles/ afterwards):
#   rocprofv3 --kernel-trace --stats over the bench command of the workload, then separate --pmc FETCH_SIZE / WRITE_SIZE
#   passes over tools/pmc_probe.py <workload> (one H.psi + the calibration kernel)
R=$GRAFT_REPO_ROOT
ROUND=${ROUND:-r03}
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do
  out=$R/gpurun_out/prof_$ROUND/$w
  mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err || { echo "trace $w failed"; exit 1; }
  cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o p -- python3 $R/tools/pmc_probe.py $w > $out/pmc_fetch.log 2>&1 || { echo "pmc fetch $w failed"; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o p -- python3 $R/tools/pmc_probe.py $w > $out/pmc_write.log 2>&1 || { echo "pmc write $w failed"; exit 1; }
  { python3 $R/tools/pmc_summary.py $out/fetch FETCH_SIZE; python3 $R/tools/pmc_summary.py $out/write WRITE_SIZE; } > $out/pmc_fetch_write.txt
  echo "== $w"; cat $out/bench_under_rocprof.json; head -8 $out/kernel_stats.csv; cat $out/pmc_fetch_write.txt
  rm -rf $out/trace $out/fetch $out/write
done
