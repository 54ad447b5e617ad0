#!/usr/bin/env python3
"""bench.py — H·psi throughput of the MI355X-native plan replay on the BASELINE workloads.

One "step" = one H·psi (one replay of the GEMM-pair plan, sigma = H psi) — the unit the reference
executes 5–90 times per site inside Davidson (src/core/iterative_matrix_functions.hpp:965-973).

Workloads (--workload, config.workload in the output; all from plan structures the running reference recorded,
operator blocks / psi filled uniform [0,1) on the device like Random::fill, src/core/utils.hpp:247-252):
  cr2_true_m4000 (default; the configuration the BASELINE metric is quoted on) Cr2/SVP SU(2) mid-chain plan captured from the
                 reference at its TRUE M=2000 (sweep 1, site 20; the largest bond dimension the reference can be run at in
                 the authoring container) with every sector dimension x2 -> M=4000: 249 495 pairs, 24.1 TMAC, 90 GB
  cr2_true_m2000 / cr2_true_m1000   the reference's captures at M=2000 / M=1000 as they are
  cr2_m4000      (the default of rounds 1-2) the M=250 capture x16 -> M=4000: 98 722 pairs, 20.7 TMAC, 73 GB; every dimension
                 is a multiple of 16, i.e. of the MFMA fragment: no padding at all, which a real M=4000 structure has
  cr2_m2000      the M=250 structure x8 -> M=2000 (BASELINE configs[2])
  h10_m500       H10/STO-6G SZ at its TRUE M=500 mid-chain structure (configs[1]): 8 276 pairs, 0.48 GMAC
  hubbard_m3000  1D Hubbard L=16 U/t=4 SZ at its TRUE M=3000 structure (configs[4]): 692 pairs, 54 GMAC
With N GPUs the operator terms of the plan are sharded sum-MPO style (every rank owns a subset of the left-operator
blocks and only their data), each rank replays its share and the partial sigma is summed with ONE all-reduce per step
through the C ABI's RCCL communicator (b2x_allreduce_sum, over xGMI) — strong scaling, as
ParallelTensorFunctions::operator() does with MPI (src/core/parallel_tensor_functions.hpp:51-55).

`value` counts the ALGORITHMIC flops of the workload, 2 x the reference's nflop (2 * sum over pairs of
m0 n0 k0 + m1 n1 k1, the reference's order of operations) — the BASELINE metric.  The plan compiler executes fewer: per
pair it takes the cheaper association of op(Z).X.op(Y), it computes a stage-0 product shared by several pairs once and
sums products before a common factor (`roofline.executed_over_algorithmic_macs`; same result up to rounding, nothing is
cached across steps).  `roofline.achieved` / `frac` are the HARDWARE roofline — flops the dominant kernel EXECUTES / its
HIP-event time — and `roofline.algorithmic_tflops` is the same kernel time in the reference's flop count (it may exceed
the MFMA peak, because that work is not executed).  `--keep-order 1` replays the reference's order pair by pair.

Launch: `python bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks ITSELF (child processes under
torch.distributed.run, 127.0.0.1; decided before this process touches the GPU or imports torch) and fails — exit code 3,
no JSON line — when fewer than N devices are visible: a one-GPU result is never printed under n_gpus = N.
(`B2X_BENCH_SHARED_CARD=1` lets the ranks share cards, sigma summed through the host: a rehearsal of the data path for
one-GPU boxes, marked `"rehearsal_shared_card": true`.)  The N > 1 line also carries `allreduce_ms` (HIP events around
b2x_allreduce_sum), the per-rank kernel times and rooflines, and `ranks_seen` as the communicator reports them.
`--emulate-ranks K` runs the K sum-MPO shards of the workload one after another on ONE GPU and prints per-shard time,
MACs and operator bytes (load balance and how much of the shared-product saving survives the sharding) — a
measurement of the shards, not a scaling curve.  `--sweep NAME` times whole DMRG sweeps of a committed chain (own leg,
outside the H.psi timed region; see sweep_leg).

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events around the dominant kernel;
`cpu_baseline` replays a bounded, deterministic sample of the same plan on the host cores, with the reference's own
BatchGEMMSeq executor (oracle/_ref/ref_replay, kind "reference") when that binary travelled with the repo,
else with the repo's CPU restatement (kind "port").  Only that leg touches oracle/.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix peak (vendor figure; measured ceiling in DESIGN.md)
HBM_PEAK_TBS = 8.0            # HBM3E peak (MI355X_MICROARCH.md)
GOLD = os.path.join(ROOT, "tests", "golden")
WORKLOADS = {
    "cr2_m4000": ("cr2_su2_m250_sw1_site20.struct.npz", 16, 4000,
                  "Cr2/SVP SU2 mid-chain H.psi plan (reference capture M=250 sw1 site20) x16 -> M=4000"),
    "cr2_m2000": ("cr2_su2_m250_sw1_site20.struct.npz", 8, 2000,
                  "Cr2/SVP SU2 mid-chain H.psi plan (reference capture M=250 sw1 site20) x8 -> M=2000"),
    "cr2_m1000": ("cr2_su2_m250_sw1_site20.struct.npz", 4, 1000,
                  "Cr2/SVP SU2 mid-chain H.psi plan (reference capture M=250 sw1 site20) x4 -> M=1000"),
    "cr2_m500": ("cr2_su2_m250_sw1_site20.struct.npz", 2, 500,
                 "Cr2/SVP SU2 mid-chain H.psi plan (reference capture M=250 sw1 site20) x2 -> M=500"),
    "cr2_m250": ("cr2_su2_m250_sw1_site20.struct.npz", 1, 250,
                 "Cr2/SVP SU2 mid-chain H.psi plan (reference capture M=250 sw1 site20)"),
    # the same molecule and site WITHOUT occupation-guided initial bond dimensions (MPSInfo::set_bond_dimension instead of
    # set_bond_dimension_using_occ): 457 470 pairs, more than half of the MACs in blocks narrower than 8 (SURVEY's counts)
    "cr2_noocc_m4000": ("cr2_su2_m250_noocc_sw1_site20.struct.npz", 16, 4000,
                        "Cr2/SVP SU2 mid-chain H.psi plan, uniform initial bond dimensions (reference capture M=250 sw1 site20) x16 -> M=4000"),
    "cr2_noocc_m1000": ("cr2_su2_m250_noocc_sw1_site20.struct.npz", 4, 1000,
                        "Cr2/SVP SU2 mid-chain H.psi plan, uniform initial bond dimensions (reference capture M=250 sw1 site20) x4 -> M=1000"),
    # TRUE Cr2/SVP structures above the M=250 capture (fixed-M run of the reference from a random MPS, two Davidson iterations
    # per site, occupation-guided initial bond dimensions, captured in sweep 1 at site 20 like the M=250 plan): what the
    # "x f" workloads assume — pair count constant, sector populations proportional to M — is checked against these
    "cr2_true_m1000": ("cr2_su2_m1000_sw1_site20.struct.npz", 1, 1000,
                       "Cr2/SVP SU2 mid-chain H.psi plan, reference capture at its TRUE M=1000 (sw1 site20, fixed-M run)"),
    "cr2_true_m2000": ("cr2_su2_m2000_sw1_site20.struct.npz", 1, 2000,
                       "Cr2/SVP SU2 mid-chain H.psi plan, reference capture at its TRUE M=2000 (sw1 site20, fixed-M run)"),
    "cr2_true_m4000": ("cr2_su2_m2000_sw1_site20.struct.npz", 2, 4000,
                       "Cr2/SVP SU2 mid-chain H.psi plan, reference capture at TRUE M=2000 (sw1 site20) x2 -> M=4000"),
    "h10_m500": ("h10_sz_m500_sw1_site4.struct.npz", 1, 500,
                 "H10/STO-6G R=1.8 SZ mid-chain H.psi plan, reference capture at M=500 (sw1 site4)"),
    "hubbard_m3000": ("hubbard_l16_u4_sz_m3000_sw0_site7.struct.npz", 1, 3000,
                      "1D Hubbard L=16 U/t=4 SZ H.psi plan, reference capture at M=3000 (sw0 site7, fixed-M random MPS)"),
}


# committed event chains of reference runs (tests/golden/chain_*; block2-preview_amd/sweep.py replays them with every operator
# resident in HBM): name -> (fixture prefix, symmetry, sweeps in the chain, what it is)
SWEEP_CHAINS = {
    "n2_m200": ("chain_n2su2/n2c", "su2", 2, "N2/STO-3G SU2 M=200 (BASELINE configs[0]), sweeps 0-1, no noise"),
    "h10_m500": ("chain_h10sz/h10c", "sz", 2, "H10/STO-6G R=1.8 SZ M=500 (BASELINE configs[1]), sweeps 0-1, no noise"),
    "hubbard_m500": ("chain_hubu2/hubc", "sz", 4, "1D Hubbard L=16 U/t=2 (bundled FCIDUMP) SZ M=500, sweeps 0-3, no noise"),
    "cr2_m30": ("chain_cr2/cr2c", "su2", 2, "Cr2/SVP SU2 M=30, sweeps 0-1, no noise"),
    "cr2_m250": ("chain_cr2_m250_cut9/cr2g", "su2", 3, "Cr2/SVP SU2 M=250 (SURVEY 8d(i)), noises 1e-5, 1e-5, 0, cutoff 1e-9, Davidson 1e-18 (the energy-gate chain)"),
    "cr2_m500": ("chain_cr2_m500_cut9/cr2h", "su2", 2, "Cr2/SVP SU2 M=500, noises 1e-5, 0, cutoff 1e-9, Davidson 1e-18 (the M=500 energy-gate chain)"),
    "n2_noisy": ("chain_n2su2_noisy/n2n", "su2", 3, "N2/STO-3G SU2 M=200, noises 1e-5, 1e-5, 0 (perturbative noise on)"),
    "h10_noisy": ("chain_h10sz_noisy/h10n", "sz", 3, "H10/STO-6G SZ M=500, noises 1e-5, 1e-5, 0 (perturbative noise on)"),
}


def sweep_leg(args):
    """`--sweep NAME|all`: wall time of whole two-site DMRG sweeps on the device (sweep.DMRG over a committed event chain:
    blocking, rotation, operator sums, H_eff plan + diagonal, device-resident Davidson, [perturbative noise,] density-matrix
    split), per sweep, with the breakdown the reference prints (Teff / Teig / Tprt / Tblk / Tsplt, sweep_algorithm.hpp:
    3208-3217) and, beside it, the reference's own clock for the same sweep as its generator logged it (SWEEP_TIME lines of
    the fixture's log: 8 threads of the authoring container — a stated baseline, not the target).  One JSON line."""
    from block2_preview_amd import capi
    from block2_preview_amd.sweep import DMRG, ChainFixture

    capi.device_init(0)
    names = sorted(SWEEP_CHAINS) if args.sweep == "all" else [x for x in args.sweep.split(",") if x]
    res = {}
    # the reference's own clock for the same schedules (tests/golden/make_ref_times.sh: block2 on 8 threads of the authoring
    # container, quiet machine, no event dumping; "default" = block2's default contraction settings)
    rt_file = os.path.join(GOLD, "ref_sweep_times.json")
    ref_times = json.load(open(rt_file)) if os.path.exists(rt_file) else {}
    # ... and its Davidson iteration counts per sweep (tests/golden/make_ref_ndav.sh: "Ndav =" of block2's own site lines)
    nd_file = os.path.join(GOLD, "ref_sweep_ndav.json")
    ref_ndav = json.load(open(nd_file)) if os.path.exists(nd_file) else {}
    for name in names:
        if name not in SWEEP_CHAINS:
            raise SystemExit("unknown chain %r (have: %s)" % (name, ", ".join(sorted(SWEEP_CHAINS))))
        prefix, sym, n_sw, what = SWEEP_CHAINS[name]
        path = os.path.join(GOLD, prefix)
        if not (os.path.exists(path + ".log") or os.path.exists(path + ".zip")):
            res[name] = {"error": "fixture %s not present" % prefix}
            continue
        for rep in range(2):  # the second pass finds every plan in the compiled-plan cache: a calculation at settled M
            if rep == 0:
                capi.plan_cache_clear()
            fx = ChainFixture(path).preload()  # (the symbolic side in memory, as block2 holds its MPO: not read inside the clock)
            dm = DMRG(fx, sym, conv_thrd=1e-18 if name in ("cr2_m250", "cr2_m500") else 1e-13)  # (the thresholds the chains were recorded with)
            t0 = time.perf_counter()
            dm.init_environments()
            capi.device_sync()
            t_init = time.perf_counter() - t0
            sweeps = []
            for isw in range(n_sw):
                before = dict(dm.tm)
                nd0 = sum(dm.ndav.values())
                t0 = time.perf_counter()
                es = dm.sweep(isw, isw % 2 == 0)
                capi.device_sync()
                wall = time.perf_counter() - t0
                tm = {k: round(dm.tm.get(k, 0.0) - before.get(k, 0.0), 4) for k in dm.tm if k != "site_total"}
                worst = max(abs(dm.energies[k] - e) for k, e in fx.ref_energy.items() if k[0] == isw)
                row = {"sweep": isw, "wall_s": round(wall, 4), "n_sites": len(es), "n_hpsi": sum(dm.ndav.values()) - nd0,
                       "energy": min(es), "worst_site_dE_vs_reference": worst,
                       # the reference's grouping: Teff = effective-Hamiltonian set-up, Teig = Davidson, Tprt = noise,
                       # Tblk = blocking + rotation + operator sums (incl. the move to the next site), Tsplt = decomposition
                       "Teff": round(tm.get("eff_ham.record", 0) + tm.get("eff_ham.device", 0), 4),
                       "Teig": tm.get("eigs", 0.0),
                       "Tprt": round(tm.get("noise.record", 0) + tm.get("noise.device", 0), 4),
                       "Tblk": round(tm.get("block", 0) + tm.get("rotate", 0) + tm.get("transform", 0) + tm.get("assign", 0), 4),
                       "Tsplt": tm.get("split", 0.0),
                       # inside Teig: moving the previous site's wavefunction to this site (sweep.DMRG._guess, host)
                       "Tguess": tm.get("guess", 0.0),
                       "starts": {h: sum(1 for k, v in dm.guess_log.items() if k[0] == isw and v and v[0] == h)
                                  for h in ("previous", "same", "diagonal")}}
                rts = ref_times.get(name, {}).get("default", {}).get("sweeps", [])
                if isw < len(rts):
                    rt = rts[isw]
                    row["reference_cpu"] = {"wall_s": rt[0], "Teff": rt[1], "Teig": rt[2], "Tprt": rt[3], "Tblk": rt[4],
                                            "Tsplt": rt[7], "threads": 8, "settings": "block2 defaults"}
                    row["speedup_vs_reference_cpu"] = round(rt[0] / wall, 3)
                    nd = ref_ndav.get(name, {}).get("per_sweep", [])
                    if isw < len(nd):
                        row["reference_cpu"]["n_hpsi"] = nd[isw]
                sweeps.append(row)
            assert fx.pos == len(fx.events), "the chain was not replayed to its end"
            key = name if rep == 0 else name + "_plans_cached"
            res[key] = {"what": what + ("" if rep == 0 else " — second pass, every plan from the compiled-plan cache"),
                        "init_environments_s": round(t_init, 4), "sweeps": sweeps, "final_energy": min(dm.energies.values()),
                        "reference_final_energy": getattr(fx, "final_energy", None),
                        "reference_total_time_s": fx.ref_total_time}
            for t in list(dm.L.values()) + list(dm.R.values()):
                t.close()
    print(json.dumps({"mode": "sweep wall time (sweep.DMRG over committed reference event chains, one MI355X)",
                      "chains": res}), flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cr2_true_m4000", choices=sorted(WORKLOADS))
    ap.add_argument("--scale", type=int, default=0, help="override the sector-dimension multiplier of the workload")
    ap.add_argument("--struct", default="", help="override the plan structure file of the workload")
    ap.add_argument("--cpu-gmac", type=float, default=0.0, help="MAC budget of the cpu_baseline sample (0 = auto)")
    ap.add_argument("--cpu-reps", type=int, default=3, help="timed replays of the cpu_baseline sample (after one warm-up)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--site-step", type=int, default=-1,
                    help="1: also time the other device steps of a site (noise, rotation, blocking) at this bond dimension; "
                         "default: on for the one-GPU Cr2 workloads up to M=4000")
    ap.add_argument("--tile-n", type=int, default=0)
    ap.add_argument("--item-macs", type=int, default=0)
    ap.add_argument("--tile-m", type=int, default=0, help="tallest sector kept on the fused wave kernel (0 = default)")
    ap.add_argument("--keep-order", type=int, default=0, help="1: always X.op(Y) first, as the reference (no per-pair reassociation)")
    ap.add_argument("--two-stage", type=int, default=0, help="0 auto, 1 all sectors through the grouped-GEMM path, -1 never")
    ap.add_argument("--scratch-mb", type=int, default=0, help="W scratch budget of the two-stage path (MiB, 0 = default)")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="K: run the K sum-MPO shards of the workload one after another on one GPU (per-shard table)")
    ap.add_argument("--sweep", default="", help="time the sweeps of a committed chain: " + ", ".join(sorted(SWEEP_CHAINS)) + ", or all")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` started plainly (no RANK / WORLD_SIZE): start the N ranks as children of this process,
    which has not imported torch or touched the GPU (never an exec of a process that initialised the device).  The
    device count comes from a short-lived child as well."""
    import socket

    probe = subprocess.run([sys.executable, "-c", "from block2_preview_amd import capi; print(capi.device_count())"],
                           cwd=ROOT, capture_output=True, text=True, timeout=600)
    try:
        ndev = int(probe.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        print("[bench] cannot count the devices: %s" % (probe.stderr[-500:],), file=sys.stderr)
        sys.exit(3)
    if ndev < args.gpus and os.environ.get("B2X_BENCH_SHARED_CARD") != "1":
        print("[bench] --gpus %d asked for, %d device(s) visible: refusing to run (set B2X_BENCH_SHARED_CARD=1 for a "
              "shared-card rehearsal of the data path)" % (args.gpus, ndev), file=sys.stderr)
        sys.exit(3)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    sys.exit(subprocess.run(cmd, cwd=ROOT).returncode)


def sample_pairs(plan_pairs, budget_macs, max_op_bytes=12e9):
    """Deterministic sample of the plan, stratified by pair size: the pairs are ordered by their MAC count and every
    k-th one is taken (k chosen so that the sample holds about `budget_macs`), so every size class of the plan is
    represented in proportion and two runs time exactly the same pairs.  Returns (indices, macs)."""
    from block2_preview_amd import synth

    pmac = (plan_pairs["m0"].astype(np.int64) * plan_pairs["n0"] * plan_pairs["k0"]
            + plan_pairs["m1"].astype(np.int64) * plan_pairs["n1"] * plan_pairs["k1"])
    order = np.argsort(-pmac, kind="stable")
    total = int(pmac.sum())
    stride = max(1, int(round(total / max(budget_macs, 1.0))))
    while True:
        sel = np.sort(order[stride // 2::stride])
        if len(sel) == 0:
            sel = order[:1]
        _, alen = synth.compact_arena(plan_pairs[sel])
        if alen * 8 <= max_op_bytes or len(sel) <= 1:
            return sel, int(pmac[sel].sum())
        stride *= 2  # keep the host copy of the sample's operator blocks under the cap


def cpu_baseline(plan_pairs, psi_len, sigma_len, budget_macs, reps, log):
    """Replay a bounded deterministic sample of the pairs on the host.  Returns the JSON object."""
    from block2_preview_amd import synth
    from block2_preview_amd.planfile import PlanFile, write_plan

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    cores = max(1, min(cores, 16))  # the CPU share of a one-GPU box
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "ref_replay")
    libdir = os.path.join(ROOT, "oracle", "_ref", "lib")
    use_ref = os.path.exists(ref_bin)
    if use_ref and not os.path.exists(os.path.join(libdir, "libmkl_rt.so")) and os.path.isdir("/opt/conda/lib"):
        os.makedirs(libdir, exist_ok=True)
        for f in os.listdir("/opt/conda/lib"):
            if f.startswith("libmkl_") and not os.path.exists(os.path.join(libdir, f)):
                os.symlink(os.path.join("/opt/conda/lib", f), os.path.join(libdir, f))
    sel, macs = sample_pairs(plan_pairs, budget_macs)
    pairs, alen = synth.compact_arena(plan_pairs[sel])
    secs = None
    if use_ref:
        try:
            pf = PlanFile()
            pf.pairs, pf.psi_len, pf.sigma_len, pf.arena_len = pairs, psi_len, sigma_len, alen
            pf.max_work = int((pairs["m0"].astype(np.int64) * pairs["n0"]).max())
            with tempfile.TemporaryDirectory() as td:
                fn = os.path.join(td, "sample.plan")
                write_plan(fn, pf)
                env = dict(os.environ, MKL_THREADING_LAYER="GNU", OMP_NUM_THREADS=str(cores))
                out = subprocess.run([ref_bin, fn, "threads=%d" % cores, "reps=%d" % (reps + 1)], env=env,
                                     capture_output=True, text=True, timeout=900)
            secs = [float(l.split("sec=")[1]) for l in out.stdout.splitlines() if l.startswith("REP ")]
            if out.returncode != 0 or len(secs) != reps + 1:
                raise RuntimeError("ref_replay failed: %s %s" % (out.stdout[-300:], out.stderr[-300:]))
            secs = secs[1:]  # the first replay is the warm-up (thread start, first touches)
        except Exception as e:  # fall back to the port if the reference binary cannot run here
            log("cpu_baseline: %s; falling back to the CPU restatement" % e)
            use_ref, secs = False, None
    if secs is None:
        from oracle import oracle

        g = np.random.default_rng(1)
        arena, psi, sig = g.random(alen), g.random(psi_len), np.zeros(sigma_len)
        secs = []
        for r in range(reps + 1):
            t0 = time.time()
            oracle.replay(pairs, arena, psi, sig, 1.0, cores)
            secs.append(time.time() - t0)
        secs = secs[1:]
    med, best = float(np.median(secs)), float(min(secs))
    return {
        "value": round(2.0 * macs / med / 1e9, 3), "unit": "GFLOP/s", "cores": cores,
        "kind": "reference" if use_ref else "port",
        "best": round(2.0 * macs / best / 1e9, 3),
        "sample": "every k-th pair of the same plan ordered by size: %d pairs, %.2f GMAC, %.1f GB of operators; median of "
                  "%d timed replays after one warm-up (%.2f s median, %.2f s best; `best` = rate of the fastest), %d "
                  "threads, %s" % (
            len(sel), macs / 1e9, alen * 8 / 1e9, reps, med, best, cores,
            "block2 BatchGEMMSeq Tasked + MKL dgemm" if use_ref else "oracle/hpsi_oracle.c OpenMP loops"),
    }


def site_step(scale, M, hpsi_ms, compile_s, dev, log):
    """The other device steps of ONE site of a two-site sweep at the bond dimension of the workload, each timed on its own
    with the structures the reference recorded at the same Cr2/SVP site (M=250, sweep 1, site / center 20) scaled like the
    H.psi plan: the perturbative-noise GEMM list, the rotation of the enlarged block (a pair plan), the blocking (an
    element-wise list; synthetic blocks with the element count of the captured list x scale^2, because that list is
    not scaled exactly).  Returns the `site_step_ms` object: what a site costs besides Ndav x H.psi."""
    import torch

    from block2_preview_amd import capi, synth
    from block2_preview_amd.planfile import OUTER_TERM_DTYPE, read_gemm_list, read_outer_struct_npz, read_struct_npz

    stream = torch.cuda.current_stream().cuda_stream
    out = {"M": M, "hpsi_ms": round(hpsi_ms, 3), "hpsi_plan_compile_ms": round(compile_s * 1e3, 1)}

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    # perturbative noise (once per site while the noise is on)
    gl = synth.scale_gemm_list(read_gemm_list(os.path.join(GOLD, "cr2_su2_m250_sw1_site20.pnoise_struct.npz")), scale)
    arena_t = torch.rand(gl.arena_len, dtype=torch.float64, device=dev)
    vin = torch.rand(gl.in_len, dtype=torch.float64, device=dev)
    vout = torch.zeros(gl.out_len, dtype=torch.float64, device=dev)
    arena = capi.Arena.adopt_device(arena_t.data_ptr(), gl.arena_len, keep=arena_t)
    t0 = time.perf_counter()
    plan = capi.GemmPlan(arena, gl.gemms, gl.in_len, gl.out_len)
    out["noise_compile_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    out["noise_ms"] = round(timed(lambda: plan.execute_device(vin.data_ptr(), vout.data_ptr(), 1.0, stream)), 3)
    out["noise_operator_gb"] = round(gl.arena_len * 8 / 1e9, 2)
    plan.close()
    t0 = time.perf_counter()
    plan = capi.GemmPlan(arena, gl.gemms, gl.in_len, gl.out_len)  # the same list again: from the compiled-plan cache
    out["noise_cached_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    plan.close(), arena.close()
    del arena_t, vin, vout
    torch.cuda.empty_cache()
    # rotation of the enlarged block into the truncated basis (once per site)
    rp = synth.scale_plan(read_struct_npz(os.path.join(GOLD, "cr2_su2_m250_sw1_c20_rrot.rotstruct.npz")), scale)
    arena_t = torch.rand(rp.arena_len, dtype=torch.float64, device=dev)  # the MPS tensor
    x = torch.rand(rp.psi_len, dtype=torch.float64, device=dev)           # the enlarged operators
    v = torch.zeros(rp.sigma_len, dtype=torch.float64, device=dev)        # the rotated operators
    arena = capi.Arena.adopt_device(arena_t.data_ptr(), rp.arena_len, keep=arena_t)
    t0 = time.perf_counter()
    plan = capi.Plan(arena, rp.pairs, rp.psi_len, rp.sigma_len)
    out["rotate_compile_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    out["rotate_ms"] = round(timed(lambda: plan.execute_device(x.data_ptr(), v.data_ptr(), 1.0, stream)), 3)
    out["rotate_enlarged_gb"] = round(rp.psi_len * 8 / 1e9, 2)
    plan.close()
    t0 = time.perf_counter()
    plan = capi.Plan(arena, rp.pairs, rp.psi_len, rp.sigma_len)
    out["rotate_cached_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    plan.close(), arena.close()
    del arena_t, x, v
    torch.cuda.empty_cache()
    # blocking: block (x) scalar site operator terms, synthetic blocks carrying the captured list's element counts
    t, lens = read_outer_struct_npz(os.path.join(GOLD, "cr2_su2_m250_sw1_c20_rblk.blkstruct.npz"))
    term_elems = int((t["m"].astype(np.int64) * t["n"]).sum()) * scale * scale
    out_elems = int(lens[3]) * scale * scale
    b = 16 * scale  # block edge (the captured blocks are 8-30 wide at M=250)
    g = 3
    n_sec = max(1, out_elems // (g * b) ** 2)
    per_sub = max(1, round(term_elems / (n_sec * g * g * b * b)))
    n_blocks = 64
    rows = []
    rng = np.random.default_rng(0)
    for s_ in range(n_sec):
        for i in range(g):
            for j in range(g):
                for k in range(per_sub):
                    tr = k % 6 == 5  # a sixth of the captured terms read their block transposed
                    rows.append((b, b, 1 if tr else b, b if tr else 1, 0, 0, g * b, 1, 0, (0, 0), 0.5 + k,
                                 int(rng.integers(n_blocks)) * b * b, int(rng.integers(16)),
                                 s_ * (g * b) ** 2 + i * b * g * b + j * b))
    terms = np.array(rows, OUTER_TERM_DTYPE)
    arena_t = torch.rand(16, dtype=torch.float64, device=dev)
    vin = torch.rand(n_blocks * b * b, dtype=torch.float64, device=dev)
    vout = torch.zeros(n_sec * (g * b) ** 2, dtype=torch.float64, device=dev)
    arena = capi.Arena.adopt_device(arena_t.data_ptr(), 16, keep=arena_t)
    out["block_ms"] = round(timed(lambda: capi.outer_build(arena, terms, vin.data_ptr(), vout.data_ptr(), True,
                                                           len(vin), len(vout), stream)), 3)
    out["block_note"] = "synthetic: %d sectors of %dx%d blocks of %d^2, %d terms per block (%.2f G term elements, %.2f G outputs as the captured list x%d^2); includes the host compile of the list" % (
        n_sec, g, g, b, per_sub, len(terms) * b * b / 1e9, len(vout) / 1e9, scale)
    arena.close()
    del arena_t, vin, vout
    torch.cuda.empty_cache()
    out["site_ms_ndav10"] = round(10 * hpsi_ms + out["noise_ms"] + out["rotate_ms"] + out["block_ms"]
                                  + out["hpsi_plan_compile_ms"] + out["noise_compile_ms"] + out["rotate_compile_ms"], 1)
    out["note"] = ("one site = plan compile + Ndav x H.psi + noise + rotation + blocking, every operator resident in HBM "
                   "(no re-upload); site_ms_ndav10 assumes 10 Davidson iterations (the reference needs 5-90 per site)")
    return out


def fill_arena(arena_len, runs, dev):
    """operator data of one rank's compact arena, generated on the device: an element is a function of its offset in the
    UNSHARDED arena (a 64-bit multiplicative hash -> uniform [0,1)), so every decomposition computes the same H"""
    import torch

    arena_t = torch.empty(max(arena_len, 1), dtype=torch.float64, device=dev)
    old_start = torch.from_numpy(np.ascontiguousarray(runs[0], np.int64)).to(dev)
    new_start = torch.from_numpy(np.ascontiguousarray(runs[1], np.int64)).to(dev)
    step = 1 << 27
    for a in range(0, arena_len, step):
        e = min(arena_len, a + step)
        idx = torch.arange(a, e, dtype=torch.int64, device=dev)
        r = torch.searchsorted(new_start, idx, right=True) - 1  # the run an element of the compact arena belongs to
        gidx = old_start[r] + (idx - new_start[r])
        h = gidx * 6364136223846793005 + 1442695040888963407  # (wraps mod 2^64)
        arena_t[a:e] = ((h >> 11) & 0x1FFFFFFFFFFFFF).to(torch.float64) * (1.0 / 9007199254740992.0)
        del idx, r, gidx, h
    return arena_t


def load_workload(args):
    from block2_preview_amd import synth
    from block2_preview_amd.planfile import read_struct_npz

    sfile, scale, M, wname = WORKLOADS[args.workload]
    if args.scale:
        M, scale = M // scale * args.scale, args.scale
        wname = "%s, sector dimensions x%d -> M=%d" % (sfile, scale, M)
    if args.struct:
        sfile, wname = args.struct, "pair plan %s x%d" % (os.path.basename(args.struct), scale)
    base = read_struct_npz(sfile if os.path.isabs(sfile) else os.path.join(GOLD, sfile))
    full = synth.scale_plan(base, scale) if scale != 1 else base
    return full, scale, M, wname


def emulate_ranks(args):
    """The K shards of `synth.shard_pairs` (the sum-MPO split of bench.py --gpus K) one after another on ONE GPU: per-shard
    H.psi time, executed and algorithmic MACs, operator bytes.  Pins, without a multi-GPU node, (i) the load balance of
    the split and (ii) how much of the plan compiler's shared-product saving survives it; the sum of
    the shards' sigma is checked against the unsharded plan.  NOT a scaling curve: no collective runs, nothing overlaps."""
    import torch

    from block2_preview_amd import capi, synth

    K = args.emulate_ranks
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    capi.device_init(0)
    stream = torch.cuda.current_stream().cuda_stream
    full, scale, M, wname = load_workload(args)
    gp = torch.Generator(device=dev)
    gp.manual_seed(7)
    psi_t = torch.empty(full.psi_len, dtype=torch.float64, device=dev).uniform_(0.0, 1.0, generator=gp)
    total = torch.zeros(full.sigma_len, dtype=torch.float64, device=dev)
    rows = []
    kw = dict(tile_n=args.tile_n, item_macs=args.item_macs, scratch_mb=args.scratch_mb, two_stage=args.two_stage,
              tile_m=args.tile_m, keep_order=args.keep_order)

    def run(pairs_r):
        mine, arena_len, runs = synth.compact_arena(pairs_r, return_runs=True)
        arena_t = fill_arena(arena_len, runs, dev)
        arena = capi.Arena.adopt_device(arena_t.data_ptr(), arena_len, keep=arena_t)
        t0 = time.time()
        plan = capi.Plan(arena, mine, full.psi_len, full.sigma_len, **kw)
        cs = time.time() - t0
        st = plan.stats
        sig = torch.zeros(full.sigma_len, dtype=torch.float64, device=dev)
        for _ in range(max(1, args.warmup)):
            plan.execute_device(psi_t.data_ptr(), sig.data_ptr(), 1.0, stream)
        torch.cuda.synchronize()
        sig.zero_()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            plan.execute_device(psi_t.data_ptr(), sig.data_ptr(), 1.0, stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        sig /= args.steps
        tmp = torch.zeros_like(sig)
        k_ms, _ = plan.time_kernel(psi_t.data_ptr(), tmp.data_ptr(), max(1, min(args.steps, 3)), stream)
        del tmp
        row = {"pairs": int(len(mine)), "operator_gb": round(arena_len * 8 / 1e9, 3), "hpsi_ms": round(ms, 3),
               "kernel_ms": round(k_ms, 3), "gmac_algorithmic": round(st["macs"] / 1e9, 2),
               "gmac_executed": round(st["macs_executed"] / 1e9, 2),
               "executed_tflops": round(2.0 * st["macs_dominant"] / (k_ms * 1e-3) / 1e12, 2),
               "plan_compile_ms": round(cs * 1e3, 1)}
        plan.close(), arena.close()
        capi.plan_cache_clear()
        del arena_t
        torch.cuda.empty_cache()
        return row, sig

    one, sig1 = run(full.pairs)
    for r in range(K):
        row, sig = run(synth.shard_pairs(full.pairs, r, K))
        row["shard"] = r
        rows.append(row)
        total += sig
        del sig
    err = float((total - sig1).abs().max() / sig1.abs().max())
    slow = max(r["hpsi_ms"] for r in rows)
    out = {"mode": "emulate-ranks (shards run one after another on one GPU; not a scaling measurement)",
           "workload": wname, "name": args.workload, "M": M, "K": K, "steps": args.steps,
           "one_rank": one, "shards": rows,
           "slowest_shard_ms": slow, "mean_shard_ms": round(float(np.mean([r["hpsi_ms"] for r in rows])), 3),
           "balance_mean_over_max": round(float(np.mean([r["hpsi_ms"] for r in rows])) / slow, 4),
           "ideal_speedup_without_allreduce": round(one["hpsi_ms"] / slow, 3),
           "executed_macs_shards_over_one_rank": round(sum(r["gmac_executed"] for r in rows) / one["gmac_executed"], 4),
           "operator_gb_shards_over_one_rank": round(sum(r["operator_gb"] for r in rows) / one["operator_gb"], 4),
           "sum_of_shard_sigma_vs_one_rank_rel_err": err}
    assert err < 1e-11, "the shards' partial sigma do not sum to the unsharded H.psi: %g" % err
    print(json.dumps(out), flush=True)


def traffic_of(workload):
    """HBM bytes per H.psi of this workload from the committed PMC passes (separate rocprofv3 --pmc runs, corrected as the
    micro-architecture guide prescribes; profiles/README.md).  The counters cannot be read from inside a timed run, so the
    figure is carried with its source."""
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(tf):
        return None, None
    tj = json.load(open(tf)).get(workload)
    if not tj:
        return None, None
    return tj["fetch_bytes_per_hpsi"] + tj["write_bytes_per_hpsi"], tj.get("source")


def main():
    args = parse()
    if os.environ.get("B2X_BENCH_WATCHDOG"):  # debugging aid: dump every thread's Python stack and exit after N seconds
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["B2X_BENCH_WATCHDOG"]), exit=True)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)  # does not return
    if args.sweep:
        return sweep_leg(args)
    if args.emulate_ranks:
        return emulate_ranks(args)
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE = %d but --gpus %d: launch with --nproc-per-node equal to --gpus" % (world, args.gpus))
    log = (lambda *a: print("[bench]", *a, file=sys.stderr, flush=True)) if rank == 0 else (lambda *a: None)
    ndev = torch.cuda.device_count()
    # rehearsal of the N>1 path on a box with fewer cards than ranks (B2X_BENCH_SHARED_CARD=1 only): the ranks share cards
    shared_card = world > 1 and int(os.environ.get("LOCAL_WORLD_SIZE", world)) > ndev
    if shared_card and os.environ.get("B2X_BENCH_SHARED_CARD") != "1":
        raise SystemExit("bench.py: %d ranks on this node but %d device(s) visible — refusing to run (a result on fewer "
                         "cards than ranks is not an n_gpus = %d measurement; B2X_BENCH_SHARED_CARD=1 rehearses the data "
                         "path on shared cards)" % (int(os.environ.get("LOCAL_WORLD_SIZE", world)), ndev, world))
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from block2_preview_amd import capi, synth

    capi.device_init(local)
    comm, comm_kind = None, "none"
    if world > 1:
        import torch.distributed as dist

        # control plane (barrier, max of the rank times): gloo.  Data plane: the C ABI's RCCL communicator.
        dist.init_process_group("gloo")
        if not shared_card and os.environ.get("B2X_BENCH_COMM", "b2x") == "b2x":
            id_file = os.path.join(tempfile.gettempdir(), "b2x_rccl_id_%s_%s" % (
                os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "0")))
            nonce_t = torch.tensor([int.from_bytes(os.urandom(7), "little") | 1], dtype=torch.int64)
            dist.broadcast(nonce_t, 0)  # one session nonce per launch: a file left by an earlier run is rejected
            try:
                comm = capi.Comm(rank, world, id_file=id_file, nonce=int(nonce_t.item()))
                comm_kind = "b2x_allreduce_sum (RCCL through the C ABI)"
            except capi.B2XError as e:
                log("b2x_comm_init failed (%s)" % e)
            dist.barrier()
            if rank == 0 and os.path.exists(id_file):
                os.remove(id_file)
            ok = torch.tensor([1 if comm is not None else 0])
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                raise SystemExit("the RCCL communicator of the C ABI could not be created on every rank")
        else:
            comm_kind = "gloo via host (rehearsal: ranks share one card, RCCL refuses that)"
    t0 = time.time()
    full, scale, M, wname = load_workload(args)
    mine, arena_len, runs = synth.compact_arena(synth.shard_pairs(full.pairs, rank, world), return_runs=True)
    log("plan: %d pairs (%d on rank 0), %.3f TMAC, psi %d, operators %.2f GB on rank 0, M=%d" % (
        len(full.pairs), len(mine), full.macs / 1e12, full.psi_len, arena_len * 8 / 1e9, M))
    # synthetic data generated on the device: uniform [0,1) like Random::fill (src/core/utils.hpp:247-252).  An operator
    # element is a function of its offset in the UNSHARDED arena (a 64-bit multiplicative hash), so a rank's blocks hold the
    # same numbers in every decomposition and the summed sigma of N ranks equals the one-rank sigma
    arena_t = fill_arena(arena_len, runs, dev)
    gp = torch.Generator(device=dev)
    gp.manual_seed(7)
    psi_t = torch.empty(full.psi_len, dtype=torch.float64, device=dev).uniform_(0.0, 1.0, generator=gp)
    sigma_t = torch.zeros(full.sigma_len, dtype=torch.float64, device=dev)
    arena = capi.Arena.adopt_device(arena_t.data_ptr(), arena_len, keep=arena_t)
    torch.cuda.synchronize()
    t0 = time.time()  # plan compile = host segmentation of the pair list + upload of the work lists + scratch allocation
    plan = capi.Plan(arena, mine, full.psi_len, full.sigma_len, tile_n=args.tile_n, item_macs=args.item_macs,
                     scratch_mb=args.scratch_mb, two_stage=args.two_stage, tile_m=args.tile_m, keep_order=args.keep_order)
    st = plan.stats
    compile_s = time.time() - t0
    log("compiled in %.1f s: %s" % (compile_s, st))
    stream = torch.cuda.current_stream().cuda_stream
    host_sigma = torch.empty(full.sigma_len, dtype=torch.float64, pin_memory=True) if world > 1 and comm is None else None

    def one_step():
        sigma_t.zero_()  # Davidson clears sigma before every op() (iterative_matrix_functions.hpp:972)
        plan.execute_device(psi_t.data_ptr(), sigma_t.data_ptr(), 1.0, stream)
        if comm is not None:  # == comm->allreduce_sum(c.data, c.size()) of ParallelTensorFunctions::operator()
            comm.allreduce_sum(sigma_t.data_ptr(), full.sigma_len, stream)
        elif world > 1:
            host_sigma.copy_(sigma_t)
            dist.all_reduce(host_sigma)
            sigma_t.copy_(host_sigma)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    dt = dt_local = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    checksum = float(sigma_t.sum().item())
    # roofline of the dominant kernel on this rank: HIP events on the launch stream
    k_ms, tot_ms = plan.time_kernel(psi_t.data_ptr(), sigma_t.data_ptr(), max(1, min(args.steps, 3)), stream)
    per_rank = None
    if world > 1:
        # the all-reduce of sigma on its own: HIP events on the caller's stream around b2x_allreduce_sum (the collective runs
        # on the communicator's stream, forked from / joined to this one), every repetition entered together
        ar = []
        if comm is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(max(3, min(args.steps, 10))):
                fence()
                ev0.record()
                comm.allreduce_sum(sigma_t.data_ptr(), full.sigma_len, stream)
                ev1.record()
                torch.cuda.synchronize()
                ar.append(ev0.elapsed_time(ev1))
        me = {"rank": rank, "comm_rank_size": list(comm.rank_size()) if comm is not None else None,
              "device": local, "pairs": int(len(mine)), "operator_gb": round(arena_len * 8 / 1e9, 3),
              "macs_algorithmic": int(st["macs"]), "macs_executed": int(st["macs_executed"]),
              "kernel_ms": round(k_ms, 3), "hpsi_ms": round(tot_ms, 3), "step_ms": round(dt_local / args.steps * 1e3, 3),
              "executed_tflops": round(2.0 * st["macs_dominant"] / (k_ms * 1e-3) / 1e12, 3),
              "frac": round(2.0 * st["macs_dominant"] / (k_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
              "allreduce_ms": round(float(np.mean(ar)), 4) if ar else None,
              "allreduce_ms_min": round(float(np.min(ar)), 4) if ar else None}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, me)
    if rank == 0:
        traffic, traffic_source = traffic_of(args.workload) if world == 1 and not args.scale and not args.struct else (None, None)
        flops_step = 2.0 * full.macs
        value = flops_step * args.steps / dt / 1e9
        alg = 2.0 * st["macs_alg_dominant"] / (k_ms * 1e-3) / 1e12  # reference flop count of the pairs in that kernel
        exe = 2.0 * st["macs_dominant"] / (k_ms * 1e-3) / 1e12     # flops the kernel really executes
        out = {
            "metric": "H.psi GFLOP/s at fixed bond dim M (DMRG effective-Hamiltonian contraction)",
            "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wname, "name": args.workload, "M": M,
                       "pairs": int(len(full.pairs)), "tmac_per_step": round(full.macs / 1e12, 4),
                       "psi_len": int(full.psi_len), "operator_gb": round(full.arena_len * 8 / 1e9, 2),
                       "parallelism": "sum-MPO x%d" % world, "allreduce": comm_kind,
                       "plan_compile_s": round(compile_s, 2), "plan_device_gb": round(st["device_bytes"] / 1e9, 2)},
            # `achieved` / `frac` are the HARDWARE roofline: flops the dominant kernel executes / its HIP-event time.
            # The plan executes fewer MACs than the reference's order of operations counts, so the
            # same kernel time expressed in the reference's (algorithmic) flops is `algorithmic_tflops`, which is what
            # `value` counts and which may exceed the MFMA peak.
            "roofline": {"bound": "mfma", "achieved": round(exe, 3), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(exe / FP64_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_tflops": round(alg, 3),
                         "executed_over_algorithmic_macs": round(st["macs_executed"] / max(1, st["macs"]), 3),
                         "launches_per_step": st["n_launches"],
                         "kernel": ("gg_kernel (two-stage grouped GEMM, all launches of one H.psi)"
                                    if st["macs_issued"] else "hpsi_wave class %d" % st["dominant_class"]),
                         "kernel_ms": round(k_ms, 3), "hpsi_ms": round(tot_ms, 3),
                         "useful_over_issued_mfma": round(st["macs_dominant"] / st["macs_issued"], 3) if st["macs_issued"] else None,
                         "atomic_fallback": st["fallback"]},
            "sigma_checksum": checksum,
        }
        if traffic:
            # the other roof, from the measured FETCH traffic of one H.psi (a PMC pass of an earlier run of this workload,
            # profiles/pmc_traffic.json) and THIS run's kernel time: flops per fetched byte against the ridge of the card
            # (78.6 TFLOP/s / 8 TB/s = 9.8).  Below the ridge the kernel is bound by the bytes it fetches, whatever the MFMA
            # fraction says — the plans of M <= 500 are.
            fpb = exe * 1e12 * (k_ms * 1e-3) / traffic
            out["roofline"]["hbm_view"] = {"fetch_tb_per_s": round(traffic / (k_ms * 1e-3) / 1e12, 3), "hbm_peak_tb_per_s": HBM_PEAK_TBS,
                                           "frac_of_hbm_peak": round(traffic / (k_ms * 1e-3) / 1e12 / HBM_PEAK_TBS, 4),
                                           "flop_per_fetched_byte": round(fpb, 2),
                                           "ridge_flop_per_byte": round(FP64_MFMA_PEAK_TFLOPS / HBM_PEAK_TBS, 2),
                                           "binding_roof": "hbm" if fpb < FP64_MFMA_PEAK_TFLOPS / HBM_PEAK_TBS else "mfma",
                                           "frac_of_binding_roof": round(exe / min(FP64_MFMA_PEAK_TFLOPS, fpb * HBM_PEAK_TBS), 4)}
        if per_rank is not None:
            # the N > 1 line: what every rank measured (the headline `value` is total flops / the slowest rank's wall time)
            kms = [r["kernel_ms"] for r in per_rank]
            ars = [r["allreduce_ms"] for r in per_rank if r["allreduce_ms"] is not None]
            out["ranks_seen"] = sorted(r["comm_rank_size"][0] if r["comm_rank_size"] else r["rank"] for r in per_rank)
            out["comm_size_seen"] = sorted(set(r["comm_rank_size"][1] for r in per_rank if r["comm_rank_size"])) or None
            out["allreduce_ms"] = round(max(ars), 4) if ars else None
            out["allreduce_bytes"] = int(full.sigma_len) * 8
            out["kernel_ms_min_max"] = [min(kms), max(kms)]
            out["roofline_frac_min_max"] = [min(r["frac"] for r in per_rank), max(r["frac"] for r in per_rank)]
            out["executed_macs_all_ranks_over_one_rank_algorithmic"] = round(
                sum(r["macs_executed"] for r in per_rank) / max(1, int(full.macs)), 4)
            out["per_rank"] = per_rank
            out["rehearsal_shared_card"] = bool(shared_card)
            assert out["ranks_seen"] == list(range(world)), "not every rank reported"
        want_site = args.site_step == 1 or (args.site_step < 0 and world == 1 and args.workload.startswith("cr2_")
                                            and "noocc" not in args.workload and not args.scale and not args.struct)
        if want_site:
            # a sweep creates one plan per site and destroys it before the next: the second plan of a process re-uses the
            # device buffers of the first (buffer pool in b2x_capi.cpp), so its creation is what a site pays
            # a Davidson iteration = H.psi + the device-resident vector algebra around it (fixed 10 iterations; the operator
            # data are random, so the eigenvalue means nothing)
            dav_ms = None
            if full.psi_len == full.sigma_len:
                try:
                    from block2_preview_amd import b2x_host
                    diag_t = torch.rand(full.psi_len, dtype=torch.float64, device=dev) + 1.0
                    ket_t = psi_t.clone()
                    b2x_host.davidson_device(plan._h.value, diag_t.data_ptr(), ket_t.data_ptr(), full.psi_len, 1e-30, 5000, 3)
                    ket_t.copy_(psi_t)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    _, nd = b2x_host.davidson_device(plan._h.value, diag_t.data_ptr(), ket_t.data_ptr(), full.psi_len, 1e-30,
                                                     5000, 10)
                    dav_ms = (time.perf_counter() - t0) / max(nd, 1) * 1e3
                    del diag_t, ket_t
                except Exception as e:
                    log("davidson timing skipped: %r" % (e,))

            def recreate():
                t0 = time.time()
                pl = capi.Plan(arena, mine, full.psi_len, full.sigma_len, tile_n=args.tile_n, item_macs=args.item_macs,
                               scratch_mb=args.scratch_mb, two_stage=args.two_stage, tile_m=args.tile_m,
                               keep_order=args.keep_order)
                dt_c = time.time() - t0
                pl.close()
                return dt_c

            plan.close()
            capi.plan_cache_clear()   # -> compiled again, device buffers from the pool: a site whose structure is new
            recycled_s = recreate()
            cached_s = recreate()     # -> the same records again: the plan comes back from the compiled-plan cache
            capi.plan_cache_clear()
            arena.close()  # free the H.psi operators before the other steps' operands are generated
            del arena_t, psi_t, sigma_t
            torch.cuda.empty_cache()
            try:
                # (the noise / rotation / blocking lists are the M=250 captures scaled to this M, whatever the H.psi structure)
                out["site_step_ms"] = site_step(max(1, M // 250), M, dt / args.steps * 1e3, recycled_s, dev, log)
                out["site_step_ms"]["hpsi_plan_create_first_ms"] = round(compile_s * 1e3, 1)
                ss = out["site_step_ms"]
                ss["hpsi_plan_cached_ms"] = round(cached_s * 1e3, 1)
                it_ms = ss["hpsi_ms"]
                if dav_ms is not None:  # 10 Davidson iterations instead of 10 bare H.psi
                    ss["davidson_iter_ms"] = round(dav_ms, 3)
                    ss["site_ms_ndav10"] = round(ss["site_ms_ndav10"] + 10 * (dav_ms - ss["hpsi_ms"]), 1)
                    it_ms = dav_ms
                ss["site_ms_ndav10_cached"] = round(10 * it_ms + ss["noise_ms"] + ss["rotate_ms"] + ss["block_ms"]
                                                    + ss["hpsi_plan_cached_ms"] + ss["noise_cached_ms"] + ss["rotate_cached_ms"], 1)
                out["site_step_ms"]["note_cache"] = ("hpsi_plan_compile_ms: the records are compiled (device buffers recycled); "
                                                     "hpsi_plan_cached_ms: the same records as an earlier, destroyed plan (a "
                                                     "site revisited at settled bond dimensions): taken from the plan cache; site_ms_ndav10_cached: such a site")
            except Exception as e:  # never lose the bench line over the extra measurement
                out["site_step_ms"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_cpu:
            # ~1 TMAC of the M=4000 plan (5 s per replay at 200 GMAC/s), never more than the whole plan
            budget = args.cpu_gmac * 1e9 if args.cpu_gmac > 0 else min(float(full.macs), 1.0e12)
            out["cpu_baseline"] = cpu_baseline(full.pairs, full.psi_len, full.sigma_len, budget, args.cpu_reps, log)
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
