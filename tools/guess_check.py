"""per site of a committed chain: how Davidson's starting vector was made, its overlap with the solution, the iteration
count with and without the carried wavefunction, |dE| against the reference:  python tools/guess_check.py <chain prefix> <su2|sz> [conv]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sw = importlib.import_module("block2-preview_amd.sweep")


def run(prefix, sym, conv, use_previous):
    fx = sw.ChainFixture(prefix).preload()
    dm = sw.DMRG(fx, sym, conv_thrd=conv)
    dm.use_previous = use_previous
    dm.init_environments()
    n_sweeps = 1 + max(k[0] for k in fx.ref_energy)
    times = []
    for isw in range(n_sweeps):
        t0 = time.perf_counter()
        dm.sweep(isw, isw % 2 == 0)
        times.append(time.perf_counter() - t0)
    return dm, fx, times


if __name__ == "__main__":
    prefix, sym = sys.argv[1], sys.argv[2]
    conv = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-13
    a, fx, ta = run(prefix, sym, conv, True)
    b, _, tb = run(prefix, sym, conv, False)
    for k in sorted(a.energies):
        how, ov = a.guess_log[k]
        print("sweep %d site %2d  %-9s overlap %.9f  ndav %3d (diagonal start: %3d)  |dE| %.2e (diagonal start: %.2e)" % (
            k[0], k[1], how, ov, a.ndav[k], b.ndav[k], abs(a.energies[k] - fx.ref_energy[k]), abs(b.energies[k] - fx.ref_energy[k])))
    print("sweep times  carried: %s   diagonal: %s" % (["%.3f" % t for t in ta], ["%.3f" % t for t in tb]))
    print("H.psi total  carried: %d   diagonal: %d;  guess time %.3f s of eigs %.3f s" % (
        sum(a.ndav.values()), sum(b.ndav.values()), a.tm.get("guess", 0.0), a.tm.get("eigs", 0.0)))
