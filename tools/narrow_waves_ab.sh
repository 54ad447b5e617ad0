#!/bin/bash
# A/B of the occupancy target of the 1-wave workgroups (gg_kernel<2,1,16,.,2>: 120 VGPRs, 4 waves per SIMD as shipped).
# Build the variant first:  cd block2-preview_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
#   -DB2X_NARROW_WAVES=5 -o ../libb2x_narrow5.so b2x_capi.cpp b2x_plan.cpp b2x_comm.cpp b2x_kernels.hip -ldl
# (5 waves: 96 VGPRs + 56 bytes of scratch per lane; 6 is refused by the compiler: occupancy 3).  Result, round 3: slower
# (M=250 0.524 -> 0.564 ms), profiles/r03_narrow_kernel_occupancy_ab.txt — not shipped.
for lib in "" block2-preview_amd/libb2x_narrow5.so; do
  for w in h10_m500 cr2_m250 cr2_m500 cr2_true_m1000; do
    B2X_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python bench.py --workload $w --steps 20 --warmup 3 --no-cpu --site-step 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('lib=%-40s %-16s ms %.4f kernel_ms %.4f frac %.4f' % ('$lib' or 'shipped', j['config']['name'], j['ms_per_step'], r['kernel_ms'], r['frac']))"
  done
done
