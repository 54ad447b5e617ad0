#!/bin/bash
# FETCH_SIZE of one H.psi with an environment knob at several values: tools/fetch_ab.sh VAR "v1 v2" workload...
R=$GRAFT_REPO_ROOT
var=$1; vals=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do
  for v in $vals; do
    export $var=$v
    out=$R/gpurun_out/fetch_ab/$w.$var$v
    mkdir -p $out
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -o p -- python3 $R/tools/pmc_probe.py $w > $out/log 2>&1 || { echo "pmc $w failed"; tail -5 $out/log; exit 1; }
    echo "== $w $var=$v"; python3 $R/tools/pmc_summary.py $out/f FETCH_SIZE | grep "b2x::" | grep -v axpy
    rm -rf $out/f
  done
done
