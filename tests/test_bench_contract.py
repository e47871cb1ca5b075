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
    if not forced_dist:
        # the dominant kernel and the rest of a step are parts of the TIMED steps (gkmhip_kernel_timeline + events around
        # every step): a line whose kernel is longer than its step is not self-consistent
        assert "timed steps" in rf["kernel_ms_source"] and rf["small_kernels_ms"] >= 0
        assert rf["kernel_ms"] + rf["small_kernels_ms"] == pytest.approx(rf["step_span_ms"]) and rf["step_span_ms"] <= d["ms_per_step"] * 1.001
    e2e = d["end_to_end"]
    assert e2e["boundary_ms"] > 0 and e2e["pipeline_ms"] > 0 and 0.0 <= e2e["pipeline_auc"] <= 1.0
    # the reference caller's own geometry: a FRESH zeroed 15 000 x 15 000 matrix per call (scripts/gkmsvm.py:75-77)
    assert e2e["boundary_fresh_matrix_ms"] > 0 and e2e["boundary_fresh_matrix_rows"] == 15000
    assert e2e["boundary_fresh_matrix_untouched_outside"] is True and e2e["boundary_parity"]["fixture"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    # the parity gate is part of every line; a custom size has no reference digest and says so (never a guess)
    assert d["parity"]["fixture"] is None and d["parity"]["sha256_matches_reference"] is None and "parity_failed" not in d
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
    # packed slabs: what a rank receives is about half of what full-width rows cost, and both are in the line
    assert 0 < rf["allgather_bytes_per_rank"] < 0.62 * rf["allgather_bytes_full_width_rows"]
    assert rf["allgather_GBps_per_rank"] > 0
    assert d["parity"]["checked_copies"] == 2


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["ok", "fail", "kill"])
def test_gpus_2_measures_the_c_abi_entry_and_falls_back_to_torch(built, how):
    """`python bench.py --gpus 2` (--assembly auto): the line's numbers come from the product's one-process entry
    gkmhip_gram_allgather, run in a fresh child, the torch.distributed ranks are the cross-check under also.torch_dist.
    A child that fails (GKM_BENCH_CABI_FAIL) or is killed at its timeout does not cost the run: the line says so and
    carries the torch numbers."""
    env = {"GKM_BENCH_SHARE_GPU": "1", "GKM_BENCH_BACKEND": "gloo"}
    if how == "fail":
        env["GKM_BENCH_CABI_FAIL"] = "1"
    if how == "kill":
        env["GKM_BENCH_CABI_TIMEOUT"] = "0.5"
    d = _one_line(_run(["--gpus", "2", "--no-cpu-baseline"] + SMALL, env))
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["ranks"] == 2
    if how == "ok":
        assert d["config"]["transport"] == "p2p" and d["assembly"].startswith("cabi") and "cabi_error" not in d
        t = d["also"]["torch_dist"]
        assert t["transport"] == "gloo" and t["ranks"] == 2 and t["value"] > 0 and t["allgather_bytes_per_rank"] > 0
    else:
        assert d["config"]["transport"] == "gloo" and "FALLBACK" in d["assembly"]
        assert ("injected" in d["cabi_error"]) if how == "fail" else ("killed" in d["cabi_error"])


@pytest.mark.gpu
def test_gpus_2_under_the_launcher_prints_its_line_on_stdout(built):
    """The driver's own invocation for N > 1: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`
    (here two ranks sharing the box's one GPU).  Rank 0 runs the C-ABI child first while the other rank waits, then both
    run the torch.distributed path; ONE line, on STDOUT (round 4's first version left descriptor 1 pointing at stderr
    after run_rank and printed the merged line there)."""
    import socket
    env = dict(os.environ, GKM_BENCH_SHARE_GPU="1", GKM_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    for attempt in range(3):
        with socket.socket() as sk:      # a free port: the number is also part of the ranks' verdict-file name
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                            "--gpus", "2", "--no-cpu-baseline"] + SMALL, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=900)
        # (someone else may take the port between the probe and the launcher's own bind: the launcher then ends before
        # any rank exists -- probe again; anything else is the test's business)
        if r.returncode == 0 or b"EADDRINUSE" not in r.stderr:
            break
    d = _one_line(r)
    assert d["n_gpus"] == 2 and d["config"]["transport"] == "p2p" and d["assembly"].startswith("cabi")
    assert d["also"]["torch_dist"]["transport"] == "gloo" and d["also"]["torch_dist"]["value"] > 0
    assert d["parity"]["checked_copies"] == 2


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
    same for the overall timeout."""
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


def test_parity_gate_hashes_the_cells_the_reference_writes(tmp_path, monkeypatch):
    """bench.parity_check: SHA-256 of the strict lower triangle, row-major, against the fixture's digest (the layout
    tests/golden/make_golden.py --full writes) + the sampled cells; one flipped bit anywhere below the diagonal fails
    it, the upper triangle and the diagonal do not take part."""
    import hashlib
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    rng = np.random.default_rng(5)
    n = 97
    K = np.tril(rng.random((n, n)), -1) + np.eye(n)
    i, j = np.tril_indices(n, -1)
    tri = K[i, j]
    sel = rng.choice(tri.size, 200, replace=False)
    os.makedirs(tmp_path / "tests" / "golden")
    np.savez(tmp_path / "tests" / "golden" / "c2_full_digest.npz",
             sha256=np.frombuffer(hashlib.sha256(tri.tobytes()).digest(), dtype=np.uint8), sample_idx=sel, sample_val=tri[sel])
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    p = bench.parity_check("c2", False, K)
    assert p["ok"] is True and p["sha256_matches_reference"] is True and p["max_rel_err_sample"] == 0.0
    K2 = K.copy()
    K2[3, 50] = 7.0                      # upper triangle: not a cell the reference writes
    K2[5, 5] = 0.5                       # diagonal: not hashed
    assert bench.parity_check("c2", False, K2)["ok"] is True
    K2[60, 7] = np.nextafter(K2[60, 7], 2.0)
    bad = bench.parity_check("c2", False, K2)
    assert bad["ok"] is False and bad["sha256_matches_reference"] is False and bad["max_rel_err_sample"] < 1e-12
    assert bench.parity_check("c2", True, K)["ok"] is None            # custom size: no digest applies
    assert bench.parity_check("d600", False, K)["fixture"] is None    # a workload without a fixture
    both = bench.merge_parity([p, bad])
    assert both["ok"] is False and both["checked_copies"] == 2 and both["sha256_matches_reference_per_copy"] == [True, False]


def test_merge_of_the_two_assemblies():
    """--assembly auto: the C-ABI line is the line; torch is the cross-check; a CRASHED or killed C-ABI child is named
    in `cabi_error` and the torch numbers take its place with exit code 0; a PARITY failure of either assembly keeps
    the other's numbers but is never silent (parity_failed, exit code 3); a torch launch that ended non-zero never
    returns 0 with its line."""
    sys.path.insert(0, ROOT)
    import bench
    cabi = {"value": 9.0, "config": {"ranks": 8, "transport": "rccl"}, "roofline": {"allgather_ms": 2.0}, "parity": {"ok": True}}
    tor = {"value": 8.0, "config": {"ranks": 8, "transport": "rccl"}, "roofline": {"allgather_ms": 3.0}, "parity": {"ok": True}}
    out, rc = bench.merge_assemblies(dict(cabi), None, dict(tor), None)
    assert rc == 0 and out["value"] == 9.0 and out["also"]["torch_dist"]["value"] == 8.0 and "cabi_error" not in out
    out, rc = bench.merge_assemblies(None, "killed after 300 s", dict(tor), None)
    assert rc == 0 and out["value"] == 8.0 and out["cabi_error"] == "killed after 300 s" and "FALLBACK" in out["assembly"]
    # the product's matrix is wrong: torch's numbers are kept, but the run FAILS
    out, rc = bench.merge_assemblies(dict(cabi, parity_failed=True, value=None), None, dict(tor), None)
    assert rc == 3 and out["value"] == 8.0 and out["parity_failed"] is True and "parity" in out["cabi_error"] \
        and out["also"]["cabi"]["value"] is None and out["parity_failed_in"].startswith("gkmhip_gram_allgather")
    # the cross-check's matrix is wrong while the product's is fine: also a failure
    out, rc = bench.merge_assemblies(dict(cabi), None, dict(tor, parity_failed=True, value=None), None)
    assert rc == 3 and out["value"] == 9.0 and out["parity_failed"] is True and "torch" in out["parity_failed_in"]
    out, rc = bench.merge_assemblies(dict(cabi), None, None, "ranks failed")
    assert rc == 0 and out["value"] == 9.0 and out["also"]["torch_dist"] == {"error": "ranks failed"}
    # a line captured from a launch whose ranks ended non-zero is not a result
    out, rc = bench.merge_assemblies(None, "killed", dict(tor), "rank 1 exited with code 1", torch_rc=1)
    assert rc == 1 and out["ranks_failed"] is True and out["value"] == 8.0
    out, rc = bench.merge_assemblies(dict(cabi), None, dict(tor), "rank 1 exited with code 1", torch_rc=1)
    assert rc == 1 and out["value"] == 9.0 and out["also"]["torch_dist"]["ranks_exit_code"] == 1
    assert bench.merge_assemblies(None, "x", None, "y") == (None, 1)
    assert bench._strip_flag(["--gpus", "2", "--assembly", "auto", "--check", "--assembly=torch"], "--assembly") == ["--gpus", "2", "--check"]


def test_committed_pmc_summaries_match_the_kernel_source():
    """The line's roofline.frac comes from profiles/r*_pmc_<workload>.json, which bench.py accepts only if it was taken
    on THIS kernel source (SHA-256 of bench.KERNEL_SOURCES: the hot kernel, its headers, the launch geometry).  An edit to those files after the last
    tools/finalize_r4.sh run would silently turn frac into null in the driver's line: caught here."""
    sys.path.insert(0, ROOT)
    import bench
    for wl in ("c2", "peaks", "c5"):
        d, src = bench.pmc_summary(wl)
        assert d is not None, "%s: %s -- re-run tools/finalize_r4.sh on the GPU box" % (wl, src)
        assert d["per_launch"]["SQ_INSTS_VALU"] > 0
        assert bench.issue_model(wl) is not None, "profiles/r*_issue_model.json is stale: python3 tools/issue_model.py --round r4"
