#!/bin/bash
for w in cr2_true_m4000 cr2_m4000; do
 for mb in 0 32768 65536; do
    python bench.py --workload $w --steps 3 --warmup 1 --no-cpu --site-step 0 --scratch-mb $mb 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-16s scratch_mb %-6s ms %.3f kernel_ms %.3f frac %.4f launches %d exec/alg %.3f plan_gb %.1f compile %.2f' % ('$w','$mb',j['ms_per_step'],r['kernel_ms'],r['frac'],r['launches_per_step'],r['executed_over_algorithmic_macs'],j['config']['plan_device_gb'],j['config']['plan_compile_s']))"
 done
done
