#!/bin/bash
# split-threshold experiment: B2X_SPLIT_THR in "$1" over workloads "$2"
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/thr
mkdir -p $out
cd $R
for w in $2; do
  for t in $1; do
    B2X_SPLIT_THR=$t timeout -k 10 300 python bench.py --workload $w --steps 8 --warmup 2 --no-cpu --site-step 0 > $out/${w}_$t.json 2> $out/${w}_$t.err || echo "failed $w $t"
    python - $out/${w}_$t.json $t <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j["roofline"]
print("thr=%-5s %-16s ms %9.3f  k_ms %9.3f frac %.4f launches %d" % (sys.argv[2], j["config"]["name"], j["ms_per_step"], r["kernel_ms"], r["frac"], r["launches_per_step"]), flush=True)
PY
  done
done
