#!/bin/bash
# Davidson iterations (= H.psi applications) of the REFERENCE per sweep, for the chains of bench.py --sweep: block2
# (oracle/_ref/ref_dump) with each chain's schedule and iprint=2, "Ndav =" of every site summed per sweep -> ref_sweep_ndav.json.
# (Counts, not times: the machine need not be quiet.)
set -e
cd "$(dirname "$0")"
export MKL_THREADING_LAYER=GNU
R=../../oracle/_ref/ref_dump
D=/root/reference/data
T=/tmp/b2x_ref_ndav
mkdir -p $T
run() { name=$1; shift; $R "$@" nodelay=1 nocache=1 iprint=2 nthreads=8 scratch=$T/scr_$name > $T/$name.out 2>&1; }
run n2_m200      $D/N2.STO3G.FCIDUMP su2 200 2 $T/n2 noise=0,0 tol=1e-12 dav_thrd=1e-13
run h10_m500     $D/H10.STO6G.R1.8.FCIDUMP sz 500 2 $T/h10 noise=0,0 tol=1e-12 dav_thrd=1e-13
run hubbard_m500 $D/HUBBARD-L16.FCIDUMP sz 500 4 $T/hub noise=0,0,0,0 tol=1e-12 dav_thrd=1e-13
run n2_noisy     $D/N2.STO3G.FCIDUMP su2 200 3 $T/n2n noise=1e-5,1e-5,0 tol=1e-12 dav_thrd=1e-13
run h10_noisy    $D/H10.STO6G.R1.8.FCIDUMP sz 500 3 $T/h10n noise=1e-5,1e-5,0 tol=1e-12 dav_thrd=1e-13
run cr2_m30      $D/CR2.SVP.FCIDUMP su2 30 2 $T/cr2s noise=0,0 tol=1e-12 dav_thrd=1e-13 occ=$D/CR2.SVP.OCC
[ -n "$SKIP_CR2_M250" ] || run cr2_m250 $D/CR2.SVP.FCIDUMP su2 250 3 $T/cr2 noise=1e-5,1e-5,0 tol=1e-12 dav_thrd=1e-18 cutoff=1e-9 occ=$D/CR2.SVP.OCC
[ -n "$SKIP_CR2_M250" ] || run cr2_m500 $D/CR2.SVP.FCIDUMP su2 500 2 $T/cr2h noise=1e-5,0 tol=1e-12 dav_thrd=1e-18 cutoff=1e-9 occ=$D/CR2.SVP.OCC
python3 - <<'PY'
import glob, json, os, re
T = "/tmp/b2x_ref_ndav"
out = {"_note": "block2 reference (oracle/_ref/ref_dump, iprint=2): Davidson iterations ('Ndav =' of every site line, in visiting order) and their sum per sweep, "
                "same schedule as the chain of the same name (tests/golden/make_ref_ndav.sh)"}
for fn in sorted(glob.glob(T + "/*.out")):
    name = os.path.basename(fn)[:-4]
    sites = []  # per sweep, in the order the sweep visits its sites
    for l in open(fn, errors="replace"):
        if re.match(r"\s*Sweep =", l):
            sites.append([])
        m = re.search(r"Ndav =\s*(\d+)", l)
        if m and sites:
            sites[-1].append(int(m.group(1)))
    out[name] = {"per_sweep": [sum(x) for x in sites], "per_site": sites}
json.dump(out, open("ref_sweep_ndav.json", "w"), indent=1)
print(out)
PY
