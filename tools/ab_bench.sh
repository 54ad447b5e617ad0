#!/bin/bash
# A/B of library builds on one box: tools/ab_bench.sh "<lib1> <lib2> ..." "<workload1> ..."  (libs under block2-preview_amd/)
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/ab
mkdir -p $out
cd $R
for rep in 1 2; do
for w in $2; do
  for l in $1; do
    B2X_LIB=$R/block2-preview_amd/$l timeout -k 10 300 python bench.py --workload $w --steps 6 --warmup 2 --no-cpu --site-step 0 > $out/${l}_${w}_$rep.json 2> $out/${l}_${w}_$rep.err || echo "bench $l $w failed"
    python - $out/${l}_${w}_$rep.json $l <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j["roofline"]
print("%-16s %-14s ms %9.3f  k_ms %9.3f  exec TF %7.3f frac %.4f" % (sys.argv[2], j["config"]["name"], j["ms_per_step"], r["kernel_ms"], r["achieved"], r["frac"]), flush=True)
PY
  done
done
done
