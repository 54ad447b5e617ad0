#!/usr/bin/env python3
"""bench.py — H·psi throughput of the MI355X-native plan replay on the BASELINE workload.

One "step" = one H·psi (one replay of the GEMM-pair plan, sigma = H psi) — the unit the reference
executes 5–90 times per site inside Davidson (src/core/iterative_matrix_functions.hpp:965-973).

Workload (config.workload): the Cr2/SVP SU(2) mid-chain plan captured from the real reference at M=250
(sweep 1, site 20; tests/golden/cr2_su2_m250_sw1_site20.struct.npz) with every sector dimension scaled
x16 -> bond dimension M=4000, operator blocks / psi filled uniform [0,1) on the device.
With N GPUs the operator terms of the plan are sharded sum-MPO style (every rank owns a subset of the
left-operator blocks and only their data), each rank replays its share and the partial sigma is summed
with ONE all-reduce (RCCL over xGMI) per step — strong scaling, as ParallelTensorFunctions::operator()
does with MPI (src/core/parallel_tensor_functions.hpp:51-55).

`value` counts the ALGORITHMIC flops of the workload, 2 x the reference's nflop (2 * sum over pairs of
m0 n0 k0 + m1 n1 k1, the reference's order of operations) — the BASELINE metric.  The plan compiler executes fewer: per
pair it takes the cheaper association of op(Z).X.op(Y) and it computes a stage-0 product shared by several pairs once
(`roofline.executed_over_algorithmic_macs`, 0.58 on this plan; same result up to rounding, nothing is cached across
steps).  `roofline.achieved` / `frac` are the HARDWARE roofline — executed flops of the dominant kernel / its time —
and `roofline.algorithmic_tflops` is the same kernel time in the reference's flop count (it may exceed the MFMA peak).
`--keep-order 1` replays the reference's order pair by pair (executed == algorithmic).

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events around the dominant kernel;
`cpu_baseline` replays a bounded sample of the same plan on the host cores, with the reference's own
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scale", type=int, default=16, help="sector-dimension multiplier (16 -> M=4000)")
    ap.add_argument("--struct", default=os.path.join(ROOT, "tests", "golden", "cr2_su2_m250_sw1_site20.struct.npz"))
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target host time of the cpu_baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--tile-n", type=int, default=0)
    ap.add_argument("--item-macs", type=int, default=0)
    ap.add_argument("--tile-m", type=int, default=0, help="tallest sector kept on the fused wave kernel (0 = default)")
    ap.add_argument("--keep-order", type=int, default=0, help="1: always X.op(Y) first, as the reference (no per-pair reassociation)")
    ap.add_argument("--two-stage", type=int, default=0, help="0 auto, 1 all sectors through the grouped-GEMM path, -1 never")
    ap.add_argument("--scratch-mb", type=int, default=0, help="W scratch budget of the two-stage path (MiB, 0 = default)")
    return ap.parse_args()


def cpu_baseline(plan_pairs, psi_len, sigma_len, seconds, log):
    """Replay a bounded random sample of the pairs on the host.  Returns the JSON object."""
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
    rng = np.random.default_rng(20240)
    order = rng.permutation(len(plan_pairs))
    pmac = (plan_pairs["m0"].astype(np.int64) * plan_pairs["n0"] * plan_pairs["k0"]
            + plan_pairs["m1"].astype(np.int64) * plan_pairs["n1"] * plan_pairs["k1"])
    cum = np.cumsum(pmac[order])

    def run(target_macs, reps=1):
        n = int(np.searchsorted(cum, target_macs)) + 1
        n = min(n, len(order))
        sel = np.sort(order[:n])
        pairs, alen = synth.compact_arena(plan_pairs[sel])
        while alen * 8 > 12e9 and n > 1:  # keep the host operator sample under 12 GB
            n //= 2
            sel = np.sort(order[:n])
            pairs, alen = synth.compact_arena(plan_pairs[sel])
        macs = int(pmac[sel].sum())
        if use_ref:
            pf = PlanFile()
            pf.pairs, pf.psi_len, pf.sigma_len, pf.arena_len = pairs, psi_len, sigma_len, alen
            pf.max_work = int((pairs["m0"].astype(np.int64) * pairs["n0"]).max())
            with tempfile.TemporaryDirectory() as td:
                fn = os.path.join(td, "sample.plan")
                write_plan(fn, pf)
                env = dict(os.environ, MKL_THREADING_LAYER="GNU", OMP_NUM_THREADS=str(cores))
                out = subprocess.run([ref_bin, fn, "threads=%d" % cores, "reps=%d" % reps], env=env,
                                     capture_output=True, text=True, timeout=900)
            line = [l for l in out.stdout.splitlines() if l.startswith("REPLAY")]
            if out.returncode != 0 or not line:
                raise RuntimeError("ref_replay failed: %s %s" % (out.stdout[-300:], out.stderr[-300:]))
            sec = float(line[0].split("sec_per_replay=")[1].split()[0])
        else:
            from oracle import oracle

            g = np.random.default_rng(1)
            arena, psi, sig = g.random(alen), g.random(psi_len), np.zeros(sigma_len)
            t0 = time.time()
            for _ in range(reps):
                oracle.replay(pairs, arena, psi, sig, 1.0, cores)
            sec = (time.time() - t0) / reps
        return macs, sec, n

    reps = 1
    try:
        macs, sec, n = run(4e9)  # calibration sample
        rate = macs / max(sec, 1e-6)
        macs, sec, n = run(max(4e9, rate * seconds))  # as many pairs as fit the host-memory cap
        reps = max(1, int(round(seconds / max(sec, 1e-3))))  # ... replayed until ~`seconds` of CPU work
        if reps > 1:
            macs, sec, n = run(macs, reps)
    except Exception as e:  # fall back to the port if the reference binary cannot run here
        log("cpu_baseline: %s; falling back to the CPU restatement" % e)
        use_ref = False
        macs, sec, n = run(2e9)
        reps = 1
    return {
        "value": round(2.0 * macs / sec / 1e9, 3), "unit": "GFLOP/s", "cores": cores,
        "kind": "reference" if use_ref else "port",
        "sample": "%d randomly chosen pairs of the same plan (%.1f GMAC per replay, %.2f s per replay, %d replays, "
                  "%d threads, %s)" % (
            n, macs / 1e9, sec, reps, cores,
            "block2 BatchGEMMSeq Tasked + MKL dgemm" if use_ref else "oracle/hpsi_oracle.c OpenMP loops"),
    }


def main():
    args = parse()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("launch with --nproc-per-node equal to --gpus")
    log = (lambda *a: print("[bench]", *a, file=sys.stderr, flush=True)) if rank == 0 else (lambda *a: None)
    ndev = torch.cuda.device_count()
    if local >= ndev:  # rehearsal of the N>1 path on a 1-GPU box (B2X_DIST_BACKEND=gloo): ranks share the card
        local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = os.environ.get("B2X_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    from block2_preview_amd import capi, synth
    from block2_preview_amd.planfile import read_struct_npz

    capi.device_init(local)
    t0 = time.time()
    base = read_struct_npz(args.struct)
    full = synth.scale_plan(base, args.scale)
    mine, arena_len = synth.compact_arena(synth.shard_pairs(full.pairs, rank, world))
    M = 250 * args.scale
    log("plan: %d pairs (%d on rank 0), %.2f TMAC, psi %d, operators %.1f GB on rank 0, M=%d" % (
        len(full.pairs), len(mine), full.macs / 1e12, full.psi_len, arena_len * 8 / 1e9, M))
    # synthetic data generated on the device: uniform [0,1) like Random::fill (src/core/utils.hpp:247-252)
    g = torch.Generator(device=dev)
    g.manual_seed(1969 + rank)
    arena_t = torch.empty(max(arena_len, 1), dtype=torch.float64, device=dev)
    step = 1 << 28
    for a in range(0, arena_len, step):
        arena_t[a:a + step].uniform_(0.0, 1.0, generator=g)
    gp = torch.Generator(device=dev)
    gp.manual_seed(7)
    psi_t = torch.empty(full.psi_len, dtype=torch.float64, device=dev).uniform_(0.0, 1.0, generator=gp)
    sigma_t = torch.zeros(full.sigma_len, dtype=torch.float64, device=dev)
    arena = capi.Arena.adopt_device(arena_t.data_ptr(), arena_len, keep=arena_t)
    plan = capi.Plan(arena, mine, full.psi_len, full.sigma_len, tile_n=args.tile_n, item_macs=args.item_macs,
                     scratch_mb=args.scratch_mb, two_stage=args.two_stage, tile_m=args.tile_m, keep_order=args.keep_order)
    st = plan.stats
    log("compiled in %.1f s: %s" % (time.time() - t0, st))
    stream = torch.cuda.current_stream().cuda_stream

    def one_step():
        sigma_t.zero_()  # Davidson clears sigma before every op() (iterative_matrix_functions.hpp:972)
        plan.execute_device(psi_t.data_ptr(), sigma_t.data_ptr(), 1.0, stream)
        if world > 1:
            dist.all_reduce(sigma_t)  # RCCL sum over xGMI == comm->allreduce_sum(c.data, c.size())

    def fence():
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
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    checksum = float(sigma_t.sum().item())
    # roofline of the dominant kernel on this rank: HIP events on the launch stream
    k_ms, tot_ms = plan.time_kernel(psi_t.data_ptr(), sigma_t.data_ptr(), max(1, min(args.steps, 3)), stream)
    if rank == 0:
        traffic = None  # HBM bytes per H.psi from the PMC passes (profiles/README.md), same workload only
        tf = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tf) and world == 1 and args.struct.endswith("cr2_su2_m250_sw1_site20.struct.npz"):
            tj = json.load(open(tf))
            if tj.get("scale") == args.scale:
                traffic = tj["fetch_bytes_per_hpsi"] + tj["write_bytes_per_hpsi"]
        flops_step = 2.0 * full.macs
        value = flops_step * args.steps / dt / 1e9
        alg = 2.0 * st["macs_alg_dominant"] / (k_ms * 1e-3) / 1e12  # reference flop count of the pairs in that kernel
        exe = 2.0 * st["macs_dominant"] / (k_ms * 1e-3) / 1e12     # flops the kernel really executes
        out = {
            "metric": "H.psi GFLOP/s at fixed bond dim M (DMRG effective-Hamiltonian contraction)",
            "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("Cr2/SVP SU2 mid-chain H.psi plan (reference capture M=250 sw1 site20) x%d -> M=%d"
                                    % (args.scale, M)) if args.struct.endswith("cr2_su2_m250_sw1_site20.struct.npz") else
                                   ("pair plan %s x%d -> M=%d" % (os.path.basename(args.struct), args.scale, M)),
                       "pairs": int(len(full.pairs)), "tmac_per_step": round(full.macs / 1e12, 3),
                       "psi_len": int(full.psi_len), "operator_gb": round(full.arena_len * 8 / 1e9, 1),
                       "parallelism": "sum-MPO x%d" % world},
            # `achieved` / `frac` are the HARDWARE roofline: flops the dominant kernel executes / its HIP-event time.
            # The plan executes fewer MACs than the reference's order of operations counts, so the
            # same kernel time expressed in the reference's (algorithmic) flops is `algorithmic_tflops`, which is what
            # `value` counts and which may exceed the MFMA peak.
            "roofline": {"bound": "mfma", "achieved": round(exe, 3), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(exe / FP64_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                         "algorithmic_tflops": round(alg, 3),
                         "executed_over_algorithmic_macs": round(st["macs_executed"] / max(1, st["macs"]), 3),
                         "launches_per_step": st["n_launches"],
                         "kernel": ("gg_kernel (two-stage grouped GEMM, all launches of one H.psi)"
                                    if st["macs_issued"] else "hpsi_wave class %d" % st["dominant_class"]),
                         "kernel_ms": round(k_ms, 3),
                         "useful_over_issued_mfma": round(st["macs_dominant"] / st["macs_issued"], 3) if st["macs_issued"] else None},
            "sigma_checksum": checksum,
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(full.pairs, full.psi_len, full.sigma_len, args.cpu_seconds, log)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
