#!/bin/bash
# same-box timing of the L2-window probe builds (WRONG results by construction): every operand fetch redirected into ~2 MB
# of the arena, so the load path sees L2 hits only.  tools/probe_l2win.sh workload...
for w in "$@"; do
  for l in "" probe_onefrag probe_l2win probe_l2win_onefrag probe_noloads; do
    if [ -n "$l" ]; then export B2X_LIB=$GRAFT_REPO_ROOT/block2-preview_amd/libb2x_$l.so; else unset B2X_LIB; fi
    python bench.py --workload $w --steps 3 --warmup 1 --no-cpu --site-step 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-16s %-22s ms %9.3f kernel_ms %9.3f' % ('$w','${l:-shipped}',j['ms_per_step'],r['kernel_ms']))"
  done
done
