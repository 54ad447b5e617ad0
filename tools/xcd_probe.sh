#!/bin/bash
# item-order experiment: time + FETCH_SIZE of one H.psi for B2X_XCD_G in "$@" (0 = plain longest-first)
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/xcd
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for g in "$@"; do
  export B2X_XCD_G=$g
  python3 $R/bench.py --workload ${W:-cr2_m4000} --steps 4 --warmup 1 --no-cpu --site-step 0 > $out/bench_g$g.json 2> $out/bench_g$g.err || echo "bench failed"
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch_g$g -o p -- python3 $R/tools/pmc_probe.py ${W:-cr2_m4000} > $out/pmc_g$g.log 2>&1 || echo "pmc failed"
  python3 - $out/bench_g$g.json $g <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j["roofline"]
print("G=%s ms %.3f k_ms %.3f frac %.4f" % (sys.argv[2], j["ms_per_step"], r["kernel_ms"], r["frac"]), flush=True)
PY
  python3 $R/tools/pmc_summary.py $out/fetch_g$g FETCH_SIZE | grep gg_kernel
  rm -rf $out/fetch_g$g
done
