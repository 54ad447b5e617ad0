#!/bin/bash
# SQ counters of ONE H.psi (tools/pmc_probe.py <workload>), one rocprofv3 --pmc pass per counter pair, gg_kernel rows only:
#   tools/sq_counters.sh workload...   -> gpurun_out/sq/<workload>.txt   (copy into profiles/)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do
  out=$R/gpurun_out/sq/$w
  mkdir -p $out
  res=$R/gpurun_out/sq/$w.txt
  echo "# rocprofv3 --pmc <pair> -- python3 tools/pmc_probe.py $w   (one H.psi; one pass per counter pair; gg_kernel / hpsi_reduce rows)" > $res
  for pair in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVES SQ_WAVE_CYCLES"; do
    d=$out/$(echo $pair | tr ' ' '_')
    if timeout -k 10 300 rocprofv3 --pmc $pair --output-format csv -d $d -o p -- python3 $R/tools/pmc_probe.py $w > $d.log 2>&1; then
      for c in $pair; do python3 $R/tools/pmc_summary.py $d $c | grep "^#\|gg_kernel\|hpsi_reduce" >> $res; done
    else
      echo "# pass '$pair' failed" >> $res
    fi
    rm -rf $d
  done
  grep PMC_PROBE $out/*.log | tail -1 | sed 's/^[^:]*://' >> $res
  cat $res
done
