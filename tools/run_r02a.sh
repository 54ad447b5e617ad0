#!/bin/bash
# GPU session A of round 2: full GPU test-suite, then one bench line per BASELINE workload (no CPU leg except the default)
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r02a
mkdir -p $out
cd $R
export B2X_BENCH_WATCHDOG=100
timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload cr2_m250 --steps 1 --warmup 1 --no-cpu > $out/n2.json 2> $out/n2.err
echo "n2 rc=$?"; tail -40 $out/n2.err; cat $out/n2.json
unset B2X_BENCH_WATCHDOG
timeout -k 10 900 python -m pytest tests -m gpu -q -v --deselect tests/test_bench_contract_gpu.py::test_bench_two_ranks_share_one_card > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
tail -15 $out/pytest.log
for w in cr2_m250 cr2_m500 cr2_m1000 cr2_m2000 h10_m500 hubbard_m3000; do
  timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu > $out/bench_$w.json 2> $out/bench_$w.err || echo "bench $w failed"
  cat $out/bench_$w.json
done
timeout -k 10 400 python bench.py --steps 5 --warmup 2 > $out/bench_cr2_m4000.json 2> $out/bench_cr2_m4000.err || echo "bench m4000 failed"
cat $out/bench_cr2_m4000.json
