#!/bin/bash
# same-box timing of ablated builds of the kernel (WRONG results by construction; timing only): tools/probe_ab.sh workload...
for w in "$@"; do
  for l in "" probe_onefrag probe_noloads probe_neither; do
    if [ -n "$l" ]; then export B2X_LIB=$GRAFT_REPO_ROOT/block2-preview_amd/libb2x_$l.so; else unset B2X_LIB; fi
    python bench.py --workload $w --steps 5 --warmup 2 --no-cpu --site-step 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-14s %-16s ms %.3f kernel_ms %.3f' % ('$w','${l:-shipped}',j['ms_per_step'],r['kernel_ms']))"
  done
done
