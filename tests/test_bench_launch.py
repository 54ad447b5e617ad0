"""bench.py --gpus N started plainly (no torchrun) never prints a one-GPU result under n_gpus = N: with fewer devices than
ranks it refuses before touching anything (here: a container without any device)."""
import os
import subprocess
import sys

from conftest import ROOT
from block2_preview_amd import capi


def test_gpus_n_without_enough_devices_is_refused(built):
    if capi.device_count() >= 2:
        import pytest

        pytest.skip("two devices are present")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "B2X_BENCH_SHARED_CARD")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "cr2_m250"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 3 and "refusing to run" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    # under a launcher whose world size disagrees with --gpus the ranks stop as well
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "cr2_m250"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
