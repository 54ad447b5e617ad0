#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r02e
mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x --deselect tests/test_bench_contract_gpu.py > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
grep -E "^FAILED|passed|failed" $out/pytest.log | tail -8
export B2X_XCD_G=16
bash tools/ab_bench.sh "libb2x_prev.so libb2x.so" "cr2_m250 cr2_m500 cr2_m1000 hubbard_m3000 cr2_noocc_m1000 h10_m500"
