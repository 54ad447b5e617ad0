#!/bin/bash
# GPU session B of round 2: GPU test-suite + bench lines after the kernel fixes
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r02b
mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -v > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
grep -E "^FAILED|passed|failed" $out/pytest.log | tail -15
for w in cr2_m250 cr2_m500 cr2_m1000 cr2_m2000 h10_m500 hubbard_m3000; do
  timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu > $out/bench_$w.json 2> $out/bench_$w.err || echo "bench $w failed"
  python - $out/bench_$w.json <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j["roofline"]
print(j["config"]["name"], "ms", j["ms_per_step"], "value", j["value"], "exec TF", r["achieved"], "frac", r["frac"], "k_ms", r["kernel_ms"], "u/i", r["useful_over_issued_mfma"], "ex/alg", r["executed_over_algorithmic_macs"])
PY
done
timeout -k 10 400 python bench.py --steps 5 --warmup 2 > $out/bench_cr2_m4000.json 2> $out/bench_cr2_m4000.err || echo "bench m4000 failed"
cat $out/bench_cr2_m4000.json
