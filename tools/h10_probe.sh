#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for ts in 0 1; do
  python bench.py --workload h10_m500 --steps 20 --warmup 5 --no-cpu --two-stage $ts 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('two_stage=$ts ms', j['ms_per_step'], 'k_ms', r['kernel_ms'], 'hpsi_ms', r['hpsi_ms'], 'frac', r['frac'], 'launches', r['launches_per_step'], 'ex/alg', r['executed_over_algorithmic_macs'])
"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/h10prof -o t -- python3 $R/bench.py --workload h10_m500 --steps 20 --warmup 5 --no-cpu > /dev/null 2>&1
head -12 $(find $R/gpurun_out/h10prof -name "*kernel_stats.csv" | head -1)
