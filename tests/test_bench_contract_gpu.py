"""bench.py keeps its contract: one JSON line with the metric, the roofline object and the cpu_baseline object
(run at M=250 so that it takes seconds; the default run is the M=4000 workload)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_json_contract(gpu):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scale", "1", "--steps", "2", "--warmup", "1",
                          "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["unit"] == "GFLOP/s" and j["dtype"] == "f64" and j["data"] == "synthetic" and j["higher_is_better"] is True
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["vs_baseline"] is None
    assert j["value"] > 0 and j["ms_per_step"] > 0 and "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 78.6
    assert 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None  # the PMC figure belongs to the M=4000 workload only
    c = j["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == "GFLOP/s" and c["sample"]
