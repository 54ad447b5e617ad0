"""Child process of tests/test_comm_gpu.py: ONE sum-MPO rank.  Runs its plan on the GPU through the C ABI (the HIP
path, device-resident psi / sigma), then sums the partial sigma over the ranks.  Several ranks share card 0 on the test
box, which RCCL refuses, so the sum goes through the host mirror's gloo transport; with world == 1 the RCCL communicator
of the C ABI is used.  usage: sum_mpo_worker.py <rank> <world> <port> <shard 0|1> <plan of rank 0> [<plan of rank 1> ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    rank, world, port, shard = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    fns = sys.argv[5:]
    from block2_preview_amd import capi, synth
    from block2_preview_amd.parallel import ParallelCommunicator, ParallelTensorFunctions
    from block2_preview_amd.planfile import read_plan

    capi.device_init(0)
    if world > 1:
        import torch.distributed as dist

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        comm = ParallelCommunicator.from_gloo()
    else:
        comm = ParallelCommunicator.from_rccl(0, 1, None)
    pf = read_plan(fns[rank % len(fns)])
    pairs = synth.shard_pairs(pf.pairs, rank, world) if shard else pf.pairs
    pairs, alen = synth.compact_arena(pairs) if shard else (pairs, pf.arena_len)
    arena_h = pf.arena
    if shard:  # keep only this rank's operator blocks (what a sum-MPO rank holds): gather them into the compact arena
        full = synth.shard_pairs(pf.pairs, rank, world)
        arena_h = np.zeros(alen)
        # (compact_arena keeps the relative order of the blocks; copy block by block)
        for po, pn in zip(full, pairs):
            ey = (int(po["n0"]) - 1) * int(po["ldb0"]) + int(po["k0"]) if po["tb0"] else (int(po["k0"]) - 1) * int(po["ldb0"]) + int(po["n0"])
            ez = (int(po["k1"]) - 1) * int(po["lda1"]) + int(po["m1"]) if po["ta1"] else (int(po["m1"]) - 1) * int(po["lda1"]) + int(po["k1"])
            arena_h[int(pn["y_off"]):int(pn["y_off"]) + ey] = pf.arena[int(po["y_off"]):int(po["y_off"]) + ey]
            arena_h[int(pn["z_off"]):int(pn["z_off"]) + ez] = pf.arena[int(po["z_off"]):int(po["z_off"]) + ez]
    arena = capi.Arena.from_host([arena_h])
    plan = capi.Plan(arena, pairs, pf.psi_len, pf.sigma_len, two_stage=int(os.environ.get("B2X_TEST_TWO_STAGE", "0")))
    psi = capi.DeviceBuffer(pf.psi_len, pf.psi)
    sigma = capi.DeviceBuffer(pf.sigma_len)
    if world > 1:
        plan.execute_device(psi.ptr, sigma.ptr, 1.0)
        capi.device_sync()
        part = sigma.download()
        tot = part.copy()
        comm.allreduce_sum(tot)
        comm.barrier()
    else:  # RCCL through the C ABI, device-resident, asynchronous on the default stream
        ParallelTensorFunctions(plan, comm)(psi.ptr, sigma.ptr, 1.0)
        comm.allreduce_sum(sigma.ptr, pf.sigma_len)  # a second, explicit one: x world (= 1) stays the same
        capi.device_sync()
        part = tot = sigma.download()
    if comm.is_root():
        mx = float(np.abs(pf.sigma_ref).max())
        print(json.dumps({"n_mine": int(len(pairs)), "n_all": int(len(pf.pairs)), "err": float(np.abs(tot - pf.sigma_ref).max()),
                          "max": mx, "part_err": float(np.abs(part - pf.sigma_ref).max()), "fallback": plan.stats["fallback"]}))
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()
    comm.close()


if __name__ == "__main__":
    main()
