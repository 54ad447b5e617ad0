#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r02c
mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -v > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
grep -E "^FAILED|passed|failed" $out/pytest.log | tail -15
bash tools/ab_bench.sh "libb2x_r01.so libb2x.so" "cr2_m250 cr2_m500 cr2_m1000 cr2_m2000 cr2_m4000 hubbard_m3000 h10_m500"
