#!/bin/bash
# FETCH_SIZE of one H.psi under several full environment settings: tools/fetch2_ab.sh workload "VAR=a VAR2=b" ...
R=$GRAFT_REPO_ROOT
w=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for e in "$@"; do
  i=$((i+1)); out=$R/gpurun_out/fetch2/$w.$i; mkdir -p $out
  ( export $e; timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -o p -- python3 $R/tools/pmc_probe.py $w > $out/log 2>&1 ) || { echo "pmc failed"; tail -3 $out/log; exit 1; }
  echo "== $w $e"; python3 $R/tools/pmc_summary.py $out/f FETCH_SIZE | grep "b2x::gg" ; rm -rf $out/f
done
