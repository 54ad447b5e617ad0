#!/bin/bash
# the measurements of a round on ONE box: tools/final_round.sh  -> gpurun_out/final/ (copy what is to be judged into profiles/)
#   1. every H.psi workload (final_sweep.sh, incl. the default bench line twice with its CPU baseline)
#   2. the sweep leg over every committed chain
#   3. the sum-MPO shards of the default workload, K = 2, 4, 8, one after another on this GPU
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/final
mkdir -p $out
cd $R
tools/final_sweep.sh 2>&1 | tee $out/all_workloads.txt
timeout -k 10 900 python bench.py --sweep all > $out/sweep_all.json 2> $out/sweep_all.err || echo "sweep leg failed"
for k in 2 4 8; do
  timeout -k 10 900 python bench.py --emulate-ranks $k --steps 3 > $out/shards_k$k.json 2> $out/shards_k$k.err || echo "emulate-ranks $k failed"
done
ls -la $out
