#!/bin/bash
# The reference's own sweep wall times for the chains of bench.py --sweep (authoring container, 8 threads, quiet machine):
# block2 (oracle/_ref/ref_dump, reference headers + MKL) with the schedule each chain was recorded with, WITHOUT the event
# dumping, once with the chain's contraction settings (nodelay nocache) and once with block2's defaults (delayed contraction of
# the normal operators, contraction cache).  SWEEP_TIME lines of the logs -> ref_sweep_times.json (tests/golden).
set -e
cd "$(dirname "$0")"
export MKL_THREADING_LAYER=GNU
R=../../oracle/_ref/ref_dump
D=/root/reference/data
T=/tmp/b2x_ref_times
mkdir -p $T
run() { # name, then ref_dump arguments
  name=$1; shift
  $R "$@" nodelay=1 nocache=1 iprint=0 nthreads=8 scratch=$T/scr_$name > $T/$name.chainset.out 2>&1
  cp "$OUT.log" $T/$name.chainset.log
  $R "$@" iprint=0 nthreads=8 scratch=$T/scr_$name > $T/$name.default.out 2>&1
  cp "$OUT.log" $T/$name.default.log
}
OUT=$T/n2;   run n2_m200      $D/N2.STO3G.FCIDUMP su2 200 2 $OUT noise=0,0 tol=1e-12 dav_thrd=1e-13
OUT=$T/h10;  run h10_m500     $D/H10.STO6G.R1.8.FCIDUMP sz 500 2 $OUT noise=0,0 tol=1e-12 dav_thrd=1e-13
OUT=$T/hub;  run hubbard_m500 $D/HUBBARD-L16.FCIDUMP sz 500 4 $OUT noise=0,0,0,0 tol=1e-12 dav_thrd=1e-13
OUT=$T/n2n;  run n2_noisy     $D/N2.STO3G.FCIDUMP su2 200 3 $OUT noise=1e-5,1e-5,0 tol=1e-12 dav_thrd=1e-13
OUT=$T/h10n; run h10_noisy    $D/H10.STO6G.R1.8.FCIDUMP sz 500 3 $OUT noise=1e-5,1e-5,0 tol=1e-12 dav_thrd=1e-13
OUT=$T/cr2s; run cr2_m30      $D/CR2.SVP.FCIDUMP su2 30 2 $OUT noise=0,0 tol=1e-12 dav_thrd=1e-13 occ=$D/CR2.SVP.OCC
[ -n "$SKIP_CR2_M250" ] || { OUT=$T/cr2;  run cr2_m250     $D/CR2.SVP.FCIDUMP su2 250 3 $OUT noise=1e-5,1e-5,0 tol=1e-12 dav_thrd=1e-18 cutoff=1e-9 occ=$D/CR2.SVP.OCC; }
[ -n "$SKIP_CR2_M250" ] || { OUT=$T/cr2h; run cr2_m500 $D/CR2.SVP.FCIDUMP su2 500 2 $OUT noise=1e-5,0 tol=1e-12 dav_thrd=1e-18 cutoff=1e-9 occ=$D/CR2.SVP.OCC; }
python3 - <<'PY'
import glob, json, os
T = "/tmp/b2x_ref_times"
out = {"_note": "block2 reference (oracle/_ref/ref_dump: reference headers + MKL), 8 OpenMP threads of the authoring container, "
                "same schedule as the chain of the same name; 'chain_settings' = nodelay nocache (what the chain was recorded with), "
                "'default' = block2's defaults (delayed contraction, contraction cache); per sweep: wall, Teff, Teig, Tprt, Tblk, Tmve, Tdm, Tsplt"}
for fn in sorted(glob.glob(T + "/*.chainset.log") + glob.glob(T + "/*.default.log")):
    name, kind = os.path.basename(fn)[:-4].rsplit(".", 1)
    sw, fin = {}, None
    for l in open(fn):
        w = l.split()
        if w and w[0] == "SWEEP_TIME":
            sw[int(w[1])] = [round(float(x), 5) for x in w[2:10]]
        if w and w[0] == "FINAL_ENERGY":
            fin = float(w[1])
    out.setdefault(name, {})["chain_settings" if kind == "chainset" else "default"] = {
        "sweeps": [sw[k] for k in sorted(sw)], "final_energy": fin}
json.dump(out, open("ref_sweep_times.json", "w"), indent=1)
print(json.dumps(out)[:600])
PY
