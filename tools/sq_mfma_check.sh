#!/bin/bash
# SQ_INSTS_MFMA of one H.psi against the plan compiler's issue-slot count (b2x_plan_stats.macs_issued / 1024):
#   tools/sq_mfma_check.sh workload...      (gpurun; one --pmc pass per workload)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do
  out=$R/gpurun_out/sqchk/$w
  mkdir -p $out
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_MFMA --output-format csv -d $out/sq -o p -- python3 $R/tools/pmc_probe.py $w > $out/probe.log 2>&1 || { echo "pmc $w failed"; tail -5 $out/probe.log; exit 1; }
  echo "== $w"; grep PMC_PROBE $out/probe.log; python3 $R/tools/pmc_summary.py $out/sq SQ_INSTS_MFMA | grep "gg_kernel\|^#"
  rm -rf $out/sq
done
