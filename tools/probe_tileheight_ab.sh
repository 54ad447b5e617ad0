#!/bin/bash
for w in cr2_true_m2000 cr2_true_m4000; do
 for e in "B2X_X=0" "B2X_SHORT_FRAGS=5 B2X_MAX_UNITS=5" "B2X_SHORT_FRAGS=3 B2X_MAX_UNITS=3"; do
  for l in probe_onefrag probe_noloads; do
    export B2X_LIB=$GRAFT_REPO_ROOT/block2-preview_amd/libb2x_$l.so
    env $e python bench.py --workload $w --steps 3 --warmup 1 --no-cpu --site-step 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-14s %-36s %-16s ms %.3f kernel_ms %.3f' % ('$w','$e','$l',j['ms_per_step'],r['kernel_ms']))"
  done
 done
done
