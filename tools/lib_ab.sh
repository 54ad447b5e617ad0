#!/bin/bash
# same-box A/B of an alternative build of the library: tools/lib_ab.sh <lib file under block2-preview_amd/> workload...
alt=$1; shift
for w in "$@"; do
  for l in "" $alt; do
    if [ -n "$l" ]; then export B2X_LIB=$GRAFT_REPO_ROOT/block2-preview_amd/$l; else unset B2X_LIB; fi
    python bench.py --workload $w --steps 20 --warmup 5 --no-cpu 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-16s %-24s ms %.3f kernel_ms %.3f frac %.4f launches %d' % ('$w','${l:-shipped}',j['ms_per_step'],r['kernel_ms'],r['frac'],r['launches_per_step']))"
  done
done
