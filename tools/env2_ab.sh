#!/bin/bash
# same-box runs of bench.py under several full environment settings: tools/env2_ab.sh workload "VAR=a VAR2=b" "VAR=c" ...
w=$1; shift
for e in "$@"; do
  env $e python bench.py --workload $w --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-14s %-40s ms %.3f kernel_ms %.3f frac %.4f u/i %.3f launches %d' % ('$w','$e',j['ms_per_step'],r['kernel_ms'],r['frac'],r['useful_over_issued_mfma'],r['launches_per_step']))"
done
