#!/bin/bash
# per-DISPATCH SQ counters of one H.psi (stage 0 / stage 1 launches apart): tools/sq_per_dispatch.sh <workload>
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
w=$1
out=$R/gpurun_out/sqd/$w
mkdir -p $out
for pair in "SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_WAVES"; do
  d=$out/$(echo $pair | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc $pair --output-format csv -d $d -o p -- python3 $R/tools/pmc_probe.py $w > $d.log 2>&1 || echo "pass $pair failed"
done
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = collections.OrderedDict()
for fn in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "gg_kernel" not in r["Kernel_Name"]:
            continue
        key = int(r["Dispatch_Id"])
        rows.setdefault(key, {"name": r["Kernel_Name"].split("(")[0][-34:], "grid": r.get("Grid_Size", "")})[r["Counter_Name"]] = rows.get(key, {}).get(r["Counter_Name"], 0) + float(r["Counter_Value"])
print("# dispatch  kernel  grid  MFMA  VALU  VALU/MFMA  SALU/MFMA  waves  MFMA/wave")
for k in sorted(rows):
    r = rows[k]
    m, v, s, wv = r.get("SQ_INSTS_MFMA", 0), r.get("SQ_INSTS_VALU", 0), r.get("SQ_INSTS_SALU", 0), r.get("SQ_WAVES", 0)
    print("%5d %-34s %9s  %.3e %.3e  %.2f  %.2f  %.3e %.0f" % (k, r["name"], r["grid"], m, v, v / max(m, 1), s / max(m, 1), wv, m / max(wv, 1)))
PY
