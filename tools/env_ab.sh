#!/bin/bash
# same-box A/B of an environment knob of the plan compiler: tools/env_ab.sh VAR "v1 v2 ..." workload...
# prints ms_per_step / kernel_ms / frac per (workload, value); box-to-box variance is ~4 %, so only same-box pairs count
var=$1; vals=$2; shift 2
for w in "$@"; do
  for v in $vals; do
    env $var=$v python bench.py --workload $w --steps 20 --warmup 5 --no-cpu 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-14s $var=%-4s ms %.3f kernel_ms %.3f frac %.4f u/i %.3f launches %d' % ('$w','$v',j['ms_per_step'],r['kernel_ms'],r['frac'],r['useful_over_issued_mfma'],r['launches_per_step']))"
  done
done
