#!/bin/bash
# same-box sweep of the plan compiler's class knobs over workloads: tools/env_sweep_r03.sh "<workloads>" "<env settings separated by |>"
out=gpurun_out/r03_env_sweep.txt
: > $out
for w in $1; do
  IFS='|' read -ra ENVS <<< "$2"
  for e in "${ENVS[@]}"; do
    line=$(env $e python bench.py --workload $w --no-cpu --site-step 0 --steps 3 2>/dev/null | tail -1)
    python3 - "$w" "$e" "$line" >> $out <<'PY'
import json, sys
w, e, line = sys.argv[1:4]
try:
    j = json.loads(line); r = j["roofline"]
    print("%-16s %-44s %9.3f ms  kernel %9.3f ms  frac %.4f  useful/issued %.3f  launches %d" % (w, e, j["ms_per_step"], r["kernel_ms"], r["frac"], r["useful_over_issued_mfma"] or 0, r["launches_per_step"]))
except Exception as ex:
    print(w, e, "ERR", ex, line[:200])
PY
  done
done
cat $out
