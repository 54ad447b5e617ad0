"""bench.py keeps its contract: one JSON line with the metric, the roofline object and the cpu_baseline object
(run on the M=250 workload so that it takes seconds; the default run is the M=4000 workload), and its N > 1 path —
per-rank shard of the plan, compact arena, all-reduce of the device sigma — gives the one-rank result."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _json_line(out):
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_json_contract(gpu):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cr2_m250", "--steps", "2",
                          "--warmup", "1", "--cpu-gmac", "2", "--cpu-reps", "2"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    j = _json_line(out)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["unit"] == "GFLOP/s" and j["dtype"] == "f64" and j["data"] == "synthetic" and j["higher_is_better"] is True
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["vs_baseline"] is None
    assert j["value"] > 0 and j["ms_per_step"] > 0 and "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 78.6
    assert 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["atomic_fallback"] == 0
    assert (r["traffic"] is None) == (r["traffic_source"] is None)  # a PMC figure always names the profile it comes from
    c = j["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == "GFLOP/s" and c["sample"]
    assert c["best"] >= c["value"]


def test_bench_two_ranks_share_one_card(gpu):
    """the N = 2 path of bench.py (torch.distributed.run, as the driver launches it; the two ranks share card 0, so the
    all-reduce goes through the host): sharded sigma summed over the ranks == the one-rank sigma"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    common = ["--workload", "cr2_m250", "--steps", "1", "--warmup", "1", "--no-cpu"]
    one = _json_line(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True,
                                    text=True, timeout=600, cwd=ROOT))
    env = dict(os.environ, B2X_BENCH_SHARED_CARD="1")  # (without it bench.py refuses more ranks than cards: next test)
    two = _json_line(subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                     "--master-addr", "127.0.0.1", "--master-port", str(port),
                                     os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, capture_output=True,
                                    text=True, timeout=900, cwd=ROOT, env=env))
    assert two["n_gpus"] == 2 and "x2" in two["config"]["parallelism"]
    assert abs(two["sigma_checksum"] - one["sigma_checksum"]) <= 1e-10 * abs(one["sigma_checksum"])


def test_bench_gpus_2_started_plainly_launches_two_ranks_itself(gpu):
    """`python bench.py --gpus 2` the way the driver starts `--gpus 1` (no torchrun, WORLD_SIZE unset): bench.py starts the
    two ranks itself and the line says n_gpus 2 with both ranks seen and the per-rank figures — or, with fewer cards than
    ranks and no rehearsal switch, it REFUSES (exit code 3, no JSON line): a one-GPU result is never printed as n_gpus 2."""
    common = ["--gpus", "2", "--workload", "cr2_m250", "--steps", "1", "--warmup", "1", "--no-cpu"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "B2X_BENCH_SHARED_CARD")}
    if gpu.device_count() < 2:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True,
                             timeout=600, cwd=ROOT, env=env)
        assert out.returncode == 3 and "refusing to run" in out.stderr
        assert not [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
        env["B2X_BENCH_SHARED_CARD"] = "1"
    two = _json_line(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True,
                                    text=True, timeout=900, cwd=ROOT, env=env))
    assert two["n_gpus"] == 2 and two["ranks_seen"] == [0, 1] and len(two["per_rank"]) == 2
    assert two["rehearsal_shared_card"] == (gpu.device_count() < 2)
    for r in two["per_rank"]:
        assert r["kernel_ms"] > 0 and r["pairs"] > 0 and 0 < r["frac"] < 1
    assert two["kernel_ms_min_max"][0] <= two["kernel_ms_min_max"][1]
    if not two["rehearsal_shared_card"]:  # real RCCL ranks: the all-reduce was timed and every rank reports the communicator
        assert two["allreduce_ms"] > 0 and two["comm_size_seen"] == [2]


def test_bench_emulate_ranks_shards_sum_to_the_plan(gpu):
    """--emulate-ranks K: the K sum-MPO shards one after another on one GPU; the shards' sigma sum to the unsharded H.psi
    (asserted inside bench.py), every pair is in exactly one shard, the table has one row per shard"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cr2_m250", "--emulate-ranks", "4",
                          "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    j = _json_line(out)
    assert j["K"] == 4 and len(j["shards"]) == 4 and j["sum_of_shard_sigma_vs_one_rank_rel_err"] < 1e-11
    assert sum(r["pairs"] for r in j["shards"]) == j["one_rank"]["pairs"]
    assert abs(sum(r["gmac_algorithmic"] for r in j["shards"]) - j["one_rank"]["gmac_algorithmic"]) < 0.05
    assert 0 < j["balance_mean_over_max"] <= 1
