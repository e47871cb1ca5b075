"""bench.py's output contract: exactly one JSON line on stdout (libraries that write to file
descriptor 1, like RCCL's version banner, must not leak into it) with the fields the driver reads."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("forced_dist", [False, True])
def test_one_json_line_with_roofline_and_cpu_baseline(built, forced_dist):
    env = dict(os.environ)
    if forced_dist:                      # the sharded path through the real RCCL backend, one rank
        env["GKM_BENCH_FORCE_DIST"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--n-pos", "400",
           "--n-neg", "400", "--cpu-sample", "150"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
    assert r.returncode == 0
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "pairs/s" and d["n_gpus"] == 1 and d["value"] > 0 and d["higher_is_better"] is True
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["kernel_ms"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    assert "workload" in d["config"] and "model" not in d["config"]
