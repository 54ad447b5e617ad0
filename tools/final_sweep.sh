#!/bin/bash
# every bench workload once with the shipped library on ONE box (20 steps, no CPU leg), then the default bench line twice
# (with the CPU baseline: the two runs must agree): tools/final_sweep.sh  -> gpurun_out/final/
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/final
mkdir -p $out
cd $R
for w in cr2_m250 cr2_m500 cr2_m1000 cr2_m2000 cr2_m4000 cr2_true_m1000 cr2_true_m2000 cr2_true_m4000 h10_m500 hubbard_m3000 cr2_noocc_m1000 cr2_noocc_m4000; do
  timeout -k 10 400 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu --site-step 0 > $out/$w.json 2> $out/$w.err || echo "bench $w failed"
  python - $out/$w.json <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j["roofline"]
print("%-16s ms %9.3f  kernel_ms %9.3f  reference-count TF %7.2f  executed TF %7.3f  frac %.4f  useful/issued %s" % (j["config"]["name"], j["ms_per_step"], r["kernel_ms"], j["value"]/1e3, r["achieved"], r["frac"], r["useful_over_issued_mfma"]), flush=True)
PY
done
for i in 1 2; do
  timeout -k 10 900 python bench.py > $out/default_$i.json 2> $out/default_$i.err || echo "default bench failed"
  tail -1 $out/default_$i.json | cut -c1-400
done
