#!/usr/bin/env python3
"""Where does a replayed chain leave the reference's energies, and is it a tie at the truncation?  Runs a chain fixture
through sweep.DMRG with check_truncation: per site |dE|, whether this loop's own global choice of kept states gives the
fixture's per-sector bond dimensions, the gap between the last kept and the first discarded density-matrix weight, and the
distance of the whole spectrum from the one the reference truncated (SPECTRA lines).  usage: trunc_diag.py <prefix> <su2|sz> <n_sweeps> [Davidson threshold, default 1e-13]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from block2_preview_amd import capi  # noqa: E402
from block2_preview_amd.sweep import DMRG, ChainFixture  # noqa: E402

prefix, sym, n_sw = sys.argv[1], sys.argv[2], int(sys.argv[3])
capi.device_init(0)
fx = ChainFixture(prefix)
dm = DMRG(fx, sym, conv_thrd=float(sys.argv[4]) if len(sys.argv) > 4 else 1e-13)
dm.check_truncation = True
dm.init_environments()
for isw in range(n_sw):
    dm.sweep(isw, isw % 2 == 0)
print("# sweep site  |dE|  k/n_states same_counts  last_kept first_discarded  (kept-discarded)/w_max  band_rel_width  spectrum_max_abs_diff  dw(here) dw(ref)")
for key in sorted(dm.energies):
    de = abs(dm.energies[key] - fx.ref_energy[key])
    t = dm.trunc_log.get(key)
    if t is None:
        print("%d %2d  %.2e  (bond not split: turn-around site)" % (key[0], key[1], de))
        continue
    print("%d %2d  %.2e  %4d/%-5d %-5s  %.3e %.3e  %.2e  %s  %s  %.3e %s" % (
        key[0], key[1], de, t["k"], t["n_states"], t["same_counts"], t["last_kept"], t["first_discarded"],
        (t["last_kept"] - t["first_discarded"]) / t["w_max"],
        "%.2e" % t["band_rel_width"] if "band_rel_width" in t else "-",
        "%.2e" % t["spectrum_max_abs_diff"] if "spectrum_max_abs_diff" in t else "-",
        t["discarded_weight"], "%.3e" % t["ref_discarded_weight"] if "ref_discarded_weight" in t else "-"))
    for kk, theirs, mine in t.get("mismatch", []):
        print("        sector %s: fixture keeps %d, own choice %d" % (kk, theirs, mine))
print("final energy here %.10f reference %.10f" % (min(dm.energies.values()), fx.final_energy))
