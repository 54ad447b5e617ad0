#!/bin/bash
for w in h10_m500 cr2_m250 cr2_m500 cr2_m1000; do
 for im in 0 100000 300000 1000000; do
    python bench.py --workload $w --steps 20 --warmup 5 --no-cpu --site-step 0 --item-macs $im 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-12s item_macs %-8s ms %.4f kernel_ms %.4f frac %.4f' % ('$w','$im',j['ms_per_step'],r['kernel_ms'],r['frac']))"
 done
done
