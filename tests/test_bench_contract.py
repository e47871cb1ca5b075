"""bench.py's output contract: exactly one JSON line on stdout (libraries that write to file
descriptor 1, like RCCL's version banner, must not leak into it) with the fields the driver reads;
`python bench.py --gpus N` must start its own ranks."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "1", "--warmup", "1", "--n-pos", "400", "--n-neg", "400", "--cpu-sample", "150"]


def _run(extra, env=None, timeout=900):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    return r


def _one_line(r):
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("forced_dist", [False, True])
def test_one_json_line_with_roofline_end_to_end_and_cpu_baseline(built, forced_dist):
    env = {"GKM_BENCH_FORCE_DIST": "1"} if forced_dist else {}   # the sharded path through real RCCL, one rank
    d = _one_line(_run(SMALL, env))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "end_to_end"):
        assert key in d, key
    assert d["unit"] == "pairs/s" and d["n_gpus"] == 1 and d["value"] > 0 and d["higher_is_better"] is True
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_Gops", "pmc_source"):
        assert key in rf, key
    # a custom problem size has no PMC summary: the executed-instruction figures must be null, never stale
    assert rf["frac"] is None and rf["achieved"] is None and rf["kernel_ms"] > 0
    assert rf["algorithmic_frac"] == pytest.approx(rf["algorithmic_Gops"] / rf["peak"])
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    e2e = d["end_to_end"]
    assert e2e["boundary_ms"] > 0 and e2e["pipeline_ms"] > 0 and 0.0 <= e2e["pipeline_auc"] <= 1.0
    assert "workload" in d["config"] and "model" not in d["config"]
    if forced_dist:
        assert d["config"]["env"].get("GKM_BENCH_FORCE_DIST") == "1"   # GKM_* knobs are on record


@pytest.mark.gpu
@pytest.mark.parametrize("assembly", ["torch", "cabi"])
def test_plain_gpus_2_launches_its_own_ranks(built, assembly):
    """No launcher, no WORLD_SIZE: `python bench.py --gpus 2` (two ranks sharing the box's one GPU; the
    torch path then moves the slabs through gloo, the one-process C-ABI path by peer copies) prints one
    line with n_gpus 2, and --check holds the assembled matrix to the single-GPU one bit for bit."""
    env = {"GKM_BENCH_SHARE_GPU": "1", "GKM_BENCH_BACKEND": "gloo"}
    d = _one_line(_run(["--gpus", "2", "--assembly", assembly, "--check", "--no-cpu-baseline"] + SMALL, env))
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["roofline"]["frac"] is None   # custom size: no PMC summary applies
    assert "end_to_end" not in d and "cpu_baseline" not in d
    # what an N > 1 line says about itself: ranks, transport, every rank's own kernel, the collective, the assembly
    rf, cfg = d["roofline"], d["config"]
    assert cfg["ranks"] == 2 and cfg["chunks"] >= 1
    assert cfg["transport"] == ("gloo" if assembly == "torch" else "p2p")
    assert len(rf["kernel_ms_per_rank"]) == 2 and min(rf["kernel_ms_per_rank"]) > 0
    assert rf["kernel_ms_min"] <= rf["kernel_ms_max"] and rf["allgather_ms"] > 0 and rf["assemble_ms"] > 0
    assert sum(rf["comparisons_per_rank"]) == pytest.approx(2.0 * 290 * 290 * (800 * 801 / 2), rel=1e-9)
    if assembly == "cabi":   # the buffers of gkmhip_gram_allgather are kept between calls (warm-up call included)
        assert rf["allgather_hipmalloc_calls_in_timed_region"] == 0


def test_headline_roofline_uses_only_a_matching_pmc_summary(tmp_path, monkeypatch):
    """frac comes from rocprofv3 SQ_INSTS_VALU of the committed summary; a summary taken on other kernel
    code (hash mismatch) or another workload must be refused."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    for rel in bench.KERNEL_SOURCES:
        os.makedirs((tmp_path / rel).parent, exist_ok=True)
        (tmp_path / rel).write_text("v1 " + rel)
    good = bench.kernel_source_hash()
    (tmp_path / "profiles" / "r2_pmc_c2.json").write_text(json.dumps(
        {"workload": "c2", "kernel_source_sha256": good, "per_launch": {"SQ_INSTS_VALU": 1.0}}))
    d, src = bench.pmc_summary("c2")
    assert d is not None and src == "r2_pmc_c2.json"
    assert bench.pmc_summary("peaks")[0] is None
    (tmp_path / bench.KERNEL_SOURCES[0]).write_text("v2: the kernel changed")
    d, why = bench.pmc_summary("c2")
    assert d is None and "stale" in why


def test_multi_gpu_roofline_fields():
    """N > 1: frac = the one-GPU launch's executed lane-ops / (max-over-ranks step time x N x peak), and each
    rank's own kernel against one GPU's peak."""
    sys.path.insert(0, ROOT)
    import bench
    per_rank = [{"kernel_ms": 40.0, "comparisons": 4.0e12, "allgather_ms": 3.0, "assemble_ms": 0.5},
                {"kernel_ms": 50.0, "comparisons": 4.4e12, "allgather_ms": 3.5, "assemble_ms": 0.4}]
    insts = 7.0e10
    m = bench.multi_gpu_roofline(insts, per_rank, 0.060, 2)
    assert m["frac"] == pytest.approx(insts * 64 / 0.060 / 1e9 / (2 * bench.PEAK_INT32_GOPS))
    ipc = insts * 64 / 8.4e12
    assert m["executed_insts_per_comparison"] == pytest.approx(ipc)
    assert m["frac_per_rank_kernel"][1] == pytest.approx(ipc * 4.4e12 / 0.050 / 1e9 / bench.PEAK_INT32_GOPS)
    assert (m["kernel_ms_min"], m["kernel_ms_max"], m["allgather_ms"], m["assemble_ms"]) == (40.0, 50.0, 3.5, 0.5)
    none = bench.multi_gpu_roofline(None, per_rank, 0.060, 2)       # no matching PMC summary: nulls, never guesses
    assert none["frac"] is None and none["frac_per_rank_kernel"] == [None, None] and none["kernel_ms_max"] == 50.0


def test_launcher_stops_the_other_ranks_when_one_fails():
    """A rank that dies before the process group forms must not leave its peers (and the parent) waiting: the
    parent polls every child, ends the rest on the first non-zero exit and reports failure within seconds; the
    same for the overall timeout.  Rank 0's stdout is relayed only on success."""
    import time
    import types
    sys.path.insert(0, ROOT)
    import bench
    args = types.SimpleNamespace(gpus=3)
    prog = ("import os, sys, time\nr = int(os.environ['RANK'])\nassert os.environ['WORLD_SIZE'] == '3'\n"
            "if r == 1:\n    time.sleep(0.3); sys.exit(3)\nprint('{}', flush=True)\ntime.sleep(120)\n")
    t0 = time.time()
    assert bench.launch_ranks(args, [], cmd=[sys.executable, "-c", prog]) != 0
    assert time.time() - t0 < 20
    t0 = time.time()
    assert bench.launch_ranks(args, [], cmd=[sys.executable, "-c", "import time; time.sleep(120)"], timeout=1.0) != 0
    assert time.time() - t0 < 20
    assert bench.launch_ranks(args, [], cmd=[sys.executable, "-c", "print('{}')"]) == 0


def test_default_cpu_baseline_is_the_headline_workload():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.parse_args([]).cpu_sample == 5000                      # the whole of config 2 (~90 s on 16 cores)
    assert bench.parse_args(["--workload", "peaks"]).cpu_sample == 2000  # bounded elsewhere
    assert bench.parse_args(["--cpu-sample", "300"]).cpu_sample == 300


def test_workload_table_matches_baseline_configs():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args([])
    assert (a.n_pos, a.n_neg, a.length, a.kernel_type, a.L, a.k, a.d, a.custom) == (5000, 5000, 300, 4, 11, 7, 3, False)
    assert a.label == "configs[1]"
    p = bench.parse_args(["--workload", "peaks"])     # reference bin/gkmqc.py:150-154,181-185
    assert (p.n_pos, p.n_neg, p.length, p.L, p.k, p.d, p.generator) == (5000, 5000, 600, 10, 6, 3, "peaks")
    assert bench.parse_args(["--n-pos", "400"]).custom


def test_plain_gpus_n_without_gpus_fails_cleanly():
    """The parent must spawn, wait for and report its ranks -- here they all fail (no GPU) -- not hang
    and not claim success."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without GPUs")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--n-pos", "20", "--n-neg", "20"], timeout=300)
    assert r.returncode != 0 and r.stdout.decode().strip() == ""
