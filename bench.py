#!/usr/bin/env python3
"""bench.py -- gkm kernel pairs/sec (N=10k, 300 bp, L=11, k=7, d=3) on N GPUs of one node.

One "step" = one complete pass of the hot path over the synthetic problem with the sequences
already resident in HBM: build the per-call row tables, run the Gram kernel for this rank's rows,
all-gather the row slabs over RCCL (N > 1), normalise the assembled matrix (division by the self
norms, unit diagonal) on every rank.

    python bench.py [--gpus N --steps K --warmup W] [--workload c2|peaks|c3|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1 (`--assembly auto`, the default) measures BOTH multi-GPU drivers, each in fresh processes with a timeout:
  primary      the product's own entry behind the C ABI, gkmhip_gram_allgather (ONE process, one host thread per
               device, RCCL all-gather of the packed row slabs) -- run as a child `--assembly cabi` of rank 0 before
               any rank has touched a GPU;
  cross-check  one process per GPU, torch.distributed (backend nccl = RCCL) all_gather_into_tensor: the ranks the
               launcher started (or, without a launcher, N ranks this script starts itself) -> `also.torch_dist`.
If the primary fails, times out or fails its parity check the line says so (`cabi_error`) and carries the torch
numbers instead.  `--assembly torch|cabi` runs one of them alone.

PARITY GATE: after the timed region the matrix the last timed step produced is copied to the host and the SHA-256 of
its strict lower triangle (the cells the reference writes, src/gkmkern_pylib.c:83,218-221) is compared with the digest
of the reference's own matrix (tests/golden/<workload>_full_digest.npz) -- on every rank's / device's copy for N > 1.
A mismatch sets `value` aside (`parity_failed`, exit code 3).

Rank 0 prints ONE JSON line.  `value` = N(N-1)/2 pairs / (max-over-ranks seconds per step).
The total work is fixed as the GPU count grows (the N x N matrix is sharded by row block), so
`scaling` is "strong".
"""
import argparse
import glob
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Integer-VALU peak of MI355X: 256 CUs x 4 SIMDs x 32 lanes/clk x 2.4 GHz = 78 643 Gop/s (lane-ops: one
# 32-bit operation of one lane).  tools/valu_peak.hip measures what a pure v_bitop3_b32 / v_xor_b32 stream
# sustains on the box (profiles/*valu_peak*): both are reported.
PEAK_INT32_GOPS = 256 * 4 * 32 * 2.4
OPS_PER_COMPARISON = 6            # op model of SURVEY.md §8(d): xor, shift, or, and, popcount, compare
# the hot kernel, its per-lane arithmetic, the row packing and the launch geometry (work-item order, tables)
KERNEL_SOURCES = ("gkmqc_amd/csrc/gkm_gram_bitslice.hip", "gkmqc_amd/csrc/gkm_gram_bitslice.h", "gkmqc_amd/csrc/gkm_bitslice.h",
                  "gkmqc_amd/csrc/gkm_pack.h", "gkmqc_amd/csrc/gkm_gram.hip")

# name: (n_pos, n_neg, length, length_range, kernel_type, L, k, d, generator, label)
WORKLOADS = {
    "c2": (5000, 5000, 300, None, 4, 11, 7, 3, "iid", "configs[1]"),
    "c1": (200, 200, 300, None, 2, 10, 6, 3, "iid", "configs[0]"),
    "c3": (10000, 10000, 300, None, 4, 11, 7, 3, "iid", "configs[2]"),
    "c5": (5000, 5000, 300, (150, 600), 4, 12, 8, 4, "iid", "configs[4] on one GPU"),
    # configs[3] stand-in: ONE subset of `gkmqc.py evaluate` at its real size and parameters (reference
    # bin/gkmqc.py:150-154,181-185) on peak-like synthetic sequences (gkmqc_amd/synth.py); the genome is not here
    "peaks": (5000, 5000, 600, None, 4, 10, 6, 3, "peaks", "configs[3] stand-in (one evaluate subset)"),
    # the same size and parameters on iid ACGT: what the peak-like composition costs is peaks vs d600
    "d600": (5000, 5000, 600, None, 4, 10, 6, 3, "iid", "gkmQC's default parameters on iid sequences"),
}


def host_cores():
    """Cores this process may really use: affinity clipped by the cgroup CPU quota (the GPU
    box exposes 256 logical CPUs but grants a 16-CPU share per GPU)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def kernel_source_hash():
    """Identifies the hot kernel's code: a PMC summary taken on other code must not be used."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()


def pmc_summary(workload):
    """Per-launch rocprofv3 counters of the dominant kernel for THIS workload and THIS kernel source
    (profiles/r*_pmc_*.json written by tools/summarize_profiles.py).  -> (dict or None, why)."""
    want = kernel_source_hash()
    stale = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_*.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("workload", "c2") != workload:
            continue
        if d.get("kernel_source_sha256") == want:
            return d, os.path.basename(path)
        stale = stale or os.path.basename(path)
    if stale:
        return None, "stale: %s was taken on other kernel code (re-run tools/collect_profiles.sh)" % stale
    return None, "no PMC summary for workload %s under profiles/" % workload


def issue_model(workload):
    """tools/issue_model.py's summary for this workload and THIS kernel source (profiles/r*_issue_model.json): the hot
    kernel's VALU instructions priced with gfx950's per-opcode issue rates against the SIMD-cycles its launch had."""
    want = kernel_source_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_issue_model.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        w = d.get("workloads", {}).get(workload)
        if d.get("kernel_source_sha256") == want and w:
            m = w["measured"]
            return {"source": os.path.basename(path), "issue_cycles_over_simd_cycles": m["issue_frac"],
                    # the clock the profiled run held (GRBM_GUI_ACTIVE / 8 / kernel time) and its kernel time: the boxes of
                    # the pool hold 2.18-2.35 GHz under this kernel, which is most of the spread between runs
                    "profiled_run_clock_GHz": m.get("clock_GHz"), "profiled_run_kernel_ms": m.get("kernel_ms"),
                    "valu_instructions_in_trips": m["valu_in_trips"], "issue_cycles_in_trips": m["issue_cycles_in_trips"],
                    "cycles_full_rate": d["cycles_full_rate"], "cycles_sgpr_operand": d["cycles_sgpr_operand"],
                    "cycles_half_rate": d["cycles_half_rate"],
                    "per_shift": {k: w["per_shift"][k] for k in ("full_rate", "sgpr_operand", "half_rate", "cycles")},
                    "per_trip": {k: w["trip"][k] for k in ("full_rate", "sgpr_operand", "half_rate", "cycles")},
                    "note": "frac above prices every VALU instruction alike (peak = one per 2 cycles and SIMD); on gfx950 only "
                            "v_and/or/xor/add/sub/not/mov/lshrrev/ashrrev/bitop3 issue that fast, everything else takes about "
                            "twice as long (profiles/r3_valu_ops.txt). Priced per opcode, the kernel's VALU issue fills this "
                            "fraction of the SIMD-cycles of its launch."}
    return None


def measured_valu_peak():
    """Lane-ops/s a pure full-rate VALU stream sustains on the box (tools/valu_peak.hip), Gop/s."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu_peak.json")), reverse=True):
        try:
            return float(json.load(open(path))["best_full_rate_Gops"]), os.path.basename(path)
        except Exception:
            continue
    return None, None


# workload -> digest fixture of the REFERENCE's own full-size matrix (tests/golden/make_golden.py --full)
PARITY_FIXTURES = {"c2": "c2_full_digest.npz", "c3": "c3_full_digest.npz", "c5": "c5_full_digest.npz",
                   "peaks": "c4_full_digest.npz"}


def parity_check(workload, custom, K):
    """The parity gate of the line: K = the n x n matrix (numpy, host) the timed step produced.
    SHA-256 of the strict lower triangle in row-major order -- exactly the cells the reference writes
    (src/gkmkern_pylib.c:83) -- against the digest of the reference's own matrix, plus the largest relative error
    on the fixture's 4 000 sampled cells.  Data only: nothing under oracle/ is touched."""
    import numpy as np
    n = K.shape[0]
    golden = os.path.join(ROOT, "tests", "golden")
    if custom or workload not in PARITY_FIXTURES:
        if workload == "c1" and not custom and os.path.exists(os.path.join(golden, "synthetic_expected.npz")):
            want = np.load(os.path.join(golden, "synthetic_expected.npz"))["c1_full_K"]
            i, j = np.tril_indices(n, -1)
            got = K[i, j]
            err = float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-300)))
            return {"fixture": "tests/golden/synthetic_expected.npz:c1_full_K", "sha256_matches_reference": None,
                    "max_rel_err": err, "ok": err < 1e-6}
        return {"fixture": None, "sha256_matches_reference": None, "ok": None,
                "why": "no reference digest for this problem (custom sizes, or a workload without a fixture)"}
    path = os.path.join(golden, PARITY_FIXTURES[workload])
    if not os.path.exists(path):
        return {"fixture": None, "sha256_matches_reference": None, "ok": None, "why": "fixture file missing: " + path}
    z = np.load(path)
    h = hashlib.sha256()
    for a in range(1, n):
        h.update(memoryview(K[a, :a]))
    same = h.digest() == z["sha256"].tobytes()
    idx = z["sample_idx"].astype(np.int64)
    a = ((1.0 + np.sqrt(1.0 + 8.0 * idx)) / 2.0).astype(np.int64)
    a = np.where(a * (a - 1) // 2 > idx, a - 1, a)
    a = np.where((a + 1) * a // 2 <= idx, a + 1, a)
    j = idx - a * (a - 1) // 2
    want = z["sample_val"]
    err = float(np.max(np.abs(K[a, j] - want) / np.maximum(np.abs(want), 1e-300)))
    return {"fixture": "tests/golden/" + PARITY_FIXTURES[workload], "sha256_matches_reference": bool(same),
            "max_rel_err_sample": err, "sampled_cells": int(len(idx)),
            "cells": "strict lower triangle, row-major: what the reference writes (src/gkmkern_pylib.c:83)",
            "ok": bool(same)}


def parity_of_device_matrix(args, full):
    """D2H of a torch matrix + parity_check."""
    K = full.cpu().numpy()
    try:
        return parity_check(args.workload, args.custom, K)
    finally:
        del K


def merge_parity(per_copy):
    """One `parity` object for a line from the checks of every rank's / device's assembled copy."""
    if not per_copy:
        return None
    out = dict(per_copy[0])
    oks = [p.get("ok") for p in per_copy]
    out["checked_copies"] = len(per_copy)
    if any(o is None for o in oks):
        out["ok"] = None
    else:
        out["ok"] = all(oks)
        if out.get("sha256_matches_reference") is not None:
            out["sha256_matches_reference"] = all(p.get("sha256_matches_reference") for p in per_copy)
            out["sha256_matches_reference_per_copy"] = [p.get("sha256_matches_reference") for p in per_copy]
        for key in ("max_rel_err_sample", "max_rel_err"):
            if key in out:
                out[key] = max(p[key] for p in per_copy)
    return out


def make_problem(args):
    from gkmqc_amd import synth
    if args.generator == "peaks":
        return (synth.make_peak_sequences(11, args.n_pos, args.length, True) +
                synth.make_peak_sequences(12, args.n_neg, args.length, False))
    lr = tuple(args.length_range) if args.length_range else None
    return synth.make_sequences(1, args.n_pos, args.length, lr) + synth.make_sequences(2, args.n_neg, args.length, lr)


def write_problem_files(args, n_pos, n_neg, tmp):
    from gkmqc_amd import synth
    pf, nf = os.path.join(tmp, "p.fa"), os.path.join(tmp, "n.fa")
    if args.generator == "peaks":
        synth.write_peak_problem(pf, nf, n_pos, n_neg, args.length)
    else:
        synth.write_problem(pf, nf, n_pos, n_neg, args.length, tuple(args.length_range) if args.length_range else None)
    return pf, nf


def cpu_baseline(args):
    """The reference's own CPU path (oracle/_ref, unmodified sources built by oracle/Makefile)
    timed on this box's host cores on a bounded sample of the same workload; falls back to
    the C restatement (kind "port") if the reference build did not travel."""
    from oracle import oracle as O
    cores = host_cores()
    npos = min(args.cpu_sample, args.n_pos)
    nneg = min(args.cpu_sample, args.n_neg)
    tmp = tempfile.mkdtemp(prefix="gkm_bench_")
    if O.have_ref():
        kind, fn = "reference", O.ref_pywrapper
    else:
        kind, fn = "port", O.oracle_pywrapper
        npos = nneg = min(npos, 150)   # brute-force port: keep it to seconds
    pf, nf = write_problem_files(args, npos, nneg, tmp)
    n = npos + nneg
    opt = O.make_opt(args.kernel_type, args.L, args.k, args.d, 50, 50.0, 1.0, pf, nf, nthreads=cores, verbosity=0)
    t0 = time.time()
    rc, _, _, _ = fn(opt, n)
    wall = time.time() - t0
    assert rc == 0
    whole = (npos, nneg) == (args.n_pos, args.n_neg)
    return {"value": (n * (n - 1) / 2) / wall, "unit": "pairs/s", "cores": cores, "kind": kind,
            "n_sequences": n, "headline_workload": whole, "wall_s": wall, "cpu_model": cpu_model(),
            "sample": "%s: %d+%d x %d bp, same generator and parameters, whole gkm_main_pywrapper call "
                      "(FASTA read + tree + rows), %d row threads, %.1f s wall%s"
                      % ("the WHOLE workload" if whole else "bounded sample", npos, nneg, args.length, cores, wall,
                         "" if whole else "; the reference's pairs/s rises with N, so a bounded sample understates it")}


def end_to_end(args, dev):
    """The two PCIe-inclusive walls SURVEY.md §8(d) asks for beside the device-resident `value`, each
    from FASTA files on disk, each measured on a warm second call (the first one pays HIP start-up):
      boundary_ms  one gkm_main_pywrapper call (reference src/gkmkern_pylib.c:92-246): entry -> the
                   caller's pageable row pointers hold the lower triangle
      pipeline_ms  gkmqc_amd.gkmsvm.main: FASTA -> matrix in HBM -> 5-fold C-SVC cross-validation -> AUC
    """
    import ctypes
    import numpy as np
    from gkmqc_amd import device, gkmsvm
    tmp = tempfile.mkdtemp(prefix="gkm_e2e_")
    pf, nf = write_problem_files(args, args.n_pos, args.n_neg, tmp)
    n = args.n_pos + args.n_neg
    out = {}
    kmat = np.zeros((n, n))
    rows = (kmat.ctypes.data + np.arange(n) * kmat.strides[0]).astype(np.uintp)
    sizes = np.zeros(2, dtype=np.int32)
    opt = device.gkmOpt(args.kernel_type, args.L, args.k, args.d, 50, 50.0, 1.0, os.fsencode(pf), os.fsencode(nf),
                        host_cores(), 0)
    walls = []
    for _ in range(3):
        t0 = time.perf_counter()
        rc = device.load().gkm_main_pywrapper(ctypes.byref(opt), rows.ctypes.data, sizes.ctypes.data)
        walls.append((time.perf_counter() - t0) * 1e3)
        assert rc == 0 and int(sizes[0]) == args.n_pos
    out["boundary_ms"] = min(walls[1:])
    out["boundary_first_call_ms"] = walls[0]
    out["boundary_pairs_per_s"] = (n * (n - 1) / 2) / (out["boundary_ms"] * 1e-3)
    assert kmat[n - 1, n - 1] == 1.0 and kmat[n - 1, 0] != 0.0 and kmat[0, n - 1] == 0.0
    # the gate of `value` on the boundary's matrix too: the cells the call wrote, against the reference's digest
    out["boundary_parity"] = parity_check(args.workload, args.custom, kmat)
    del kmat, rows
    # The reference's caller allocates a FRESH zeroed 15 000 x 15 000 matrix for every call (scripts/gkmsvm.py:75-77:
    # np.zeros, i.e. untouched pages; row r at byte 120 000 r) -- the scatter into it takes the page faults that a reused
    # matrix (boundary_ms) has already paid.  Timed around the call only, as the caller's own clock would.
    cap = max(15000, n)
    fresh = []
    for _ in range(2):
        big = np.zeros((cap, cap))
        rp = (big.ctypes.data + np.arange(cap) * big.strides[0]).astype(np.uintp)
        t0 = time.perf_counter()
        rc = device.load().gkm_main_pywrapper(ctypes.byref(opt), rp.ctypes.data, sizes.ctypes.data)
        fresh.append((time.perf_counter() - t0) * 1e3)
        assert rc == 0 and int(sizes[0]) == args.n_pos
        if len(fresh) == 2:
            out["boundary_fresh_matrix_parity"] = parity_check(args.workload, args.custom, big[:n, :n])
            out["boundary_fresh_matrix_untouched_outside"] = bool((big[n:] == 0).all() and (big[:n, n:] == 0).all())
        del big, rp
    out["boundary_fresh_matrix_ms"] = min(fresh)
    out["boundary_fresh_matrix_rows"] = cap
    argv = ["-p", pf, "-n", nf, "-w", os.path.join(tmp, "e2e"), "-t", str(args.kernel_type), "-L", str(args.L),
            "-k", str(args.k), "-d", str(args.d), "-s", "1", "-v", "0"]
    walls, auc = [], None
    for _ in range(2):
        t0 = time.perf_counter()
        auc, _std = gkmsvm.main(argv)
        walls.append((time.perf_counter() - t0) * 1e3)
    out["pipeline_ms"] = walls[-1]
    out["pipeline_first_call_ms"] = walls[0]
    out["pipeline_auc"] = float(auc)
    out["note"] = ("warm calls from FASTA on disk; boundary = gkm_main_pywrapper into pageable numpy rows (PCIe "
                   "+ host scatter included, %d helper threads); pipeline = FASTA -> Gram matrix in HBM -> 5-fold "
                   "C-SVC cross-validation on the GPU -> AUC; neither is `value`" % host_cores())
    return out


def multi_gpu_roofline(insts, per_rank, sec_per_step, n_gpus, bytes_per_rank=None, bytes_full_width=None):
    """Roofline fields of an N > 1 line from (a) SQ_INSTS_VALU of the hash-matched ONE-GPU summary of the same
    workload and (b) what every rank measured itself.  The instructions executed per l-mer comparison do not
    depend on which rank computes a row, so the whole job executes what the one-GPU launch executed; it is spread
    over N GPUs for the max-over-ranks step time (kernels + collectives + assembly): frac = that / (N x peak).
    frac_per_rank_kernel holds each rank's own kernel against ONE GPU's peak."""
    total_cmp = sum(r["comparisons"] for r in per_rank) if per_rank else None
    ipc = (insts * 64 / total_cmp) if insts and total_cmp else None
    executed = (insts * 64 / sec_per_step / 1e9) if insts else None
    out = {"achieved": executed, "frac": (executed / (PEAK_INT32_GOPS * n_gpus)) if executed else None,
           "executed_insts_per_comparison": ipc, "peak_is": "%d GPUs x %.1f Gop/s" % (n_gpus, PEAK_INT32_GOPS)}
    if per_rank:
        km = [r["kernel_ms"] for r in per_rank]
        out.update({
            "kernel_ms_min": min(km), "kernel_ms_max": max(km), "kernel_ms_per_rank": km,
            "comparisons_per_rank": [r["comparisons"] for r in per_rank],
            "frac_per_rank_kernel": [(ipc * r["comparisons"] / (r["kernel_ms"] * 1e-3) / 1e9 / PEAK_INT32_GOPS)
                                     if ipc and r["kernel_ms"] > 0 else None for r in per_rank],
            "allgather_ms": max(r["allgather_ms"] for r in per_rank),
            "allgather_ms_per_rank": [r["allgather_ms"] for r in per_rank],
            "assemble_ms": max(r["assemble_ms"] for r in per_rank)})
    if bytes_per_rank is not None:
        # what every rank RECEIVES from its peers per matrix (packed row slabs: row a = a + 1 doubles), what full-width
        # rows cost in round 3, and the rate the collectives sustained on their stream (sum over the chunks)
        ag = out.get("allgather_ms")
        out.update({"allgather_bytes_per_rank": int(bytes_per_rank), "allgather_bytes_full_width_rows": bytes_full_width,
                    "allgather_GBps_per_rank": (bytes_per_rank / (ag * 1e-3) / 1e9) if ag else None})
    return out


def side_workload(name, dev, steps=3):
    """A short single-GPU measurement of another workload inside the headline line (`also`): the same step
    (row tables + Gram kernel + untile + normalise, inputs resident in HBM), `steps` timed passes after one
    warm-up, the hot kernel's HIP-event time and the hash-gated executed-instruction fraction."""
    import numpy as np
    import torch
    from gkmqc_amd import device
    a = parse_args(["--workload", name])
    seqs = [device.encode(x) for x in make_problem(a)]
    n = len(seqs)
    stream = torch.cuda.current_stream().cuda_stream
    ctx = device.GramContext(a.kernel_type, a.L, a.k, a.d, 50, 50.0, 1.0, dev.index or 0)
    try:
        ctx.set_sequences(seqs, stream)
        full = torch.zeros((n, n), dtype=torch.float64, device=dev)
        sq = torch.zeros(n, dtype=torch.float64, device=dev)
        rows = np.arange(n)

        def step():
            ctx.gram_rows(rows, full.data_ptr(), n, None, 0, False, stream)
            ctx.normalize(full.data_ptr(), n, sq.data_ptr(), False, stream)

        step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        kms = []
        for _ in range(steps):
            step()
            torch.cuda.synchronize(dev)      # (per step, so that the kernel's own events can be read)
            kms.append(ctx.last_kernel_ms())
        wall = (time.perf_counter() - t0) / steps
        comparisons = ctx.last_comparisons()
        kern_s = float(np.mean(kms)) * 1e-3
        pmc, src = pmc_summary(name)
        insts = pmc["per_launch"].get("SQ_INSTS_VALU") if pmc else None
        parity = parity_of_device_matrix(a, full)
        return {"workload": a.label, "n_sequences": n, "length": a.length_range or a.length, "L": a.L, "k": a.k, "d": a.d,
                "steps": steps, "ms_per_step": wall * 1e3, "pairs_per_s": (n * (n - 1) / 2) / wall, "parity": parity,
                "kernel": ctx.last_kernel_name(), "kernel_ms": kern_s * 1e3, "comparisons_per_s": comparisons / kern_s,
                "frac": (insts * 64 / kern_s / 1e9 / PEAK_INT32_GOPS) if insts else None, "pmc_source": src,
                # what `frac` hides: a kernel that needs fewer instructions for the same comparisons is the faster one
                "executed_insts_per_comparison": (insts * 64 / comparisons) if insts else None,
                "comparisons_per_launch": comparisons,
                "issue_cycles_over_simd_cycles": (issue_model(name) or {}).get("issue_cycles_over_simd_cycles")}
    finally:
        ctx.close()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS),
                    help="c2 = BASELINE configs[1] (the headline metric); peaks = configs[3] stand-in")
    ap.add_argument("--n-pos", type=int, default=None)
    ap.add_argument("--n-neg", type=int, default=None)
    ap.add_argument("--length", type=int, default=None)
    ap.add_argument("--length-range", type=int, nargs=2, default=None, help="uniform random lengths (config 5: 150 600)")
    ap.add_argument("--kernel-type", type=int, default=None)
    ap.add_argument("-L", type=int, default=None)
    ap.add_argument("-k", type=int, default=None)
    ap.add_argument("-d", type=int, default=None)
    ap.add_argument("--kernel", default="auto", choices=["auto", "direct", "bitslice"])
    ap.add_argument("--assembly", default="auto", choices=["auto", "torch", "cabi"],
                    help="N > 1: auto (default) = both, each in fresh processes with a timeout: the product's one-process "
                         "entry gkmhip_gram_allgather as the line's numbers, torch.distributed as also.torch_dist (and as "
                         "the fallback); cabi = one process, gkmhip_gram_allgather (one host thread per device) alone; "
                         "torch = one process per GPU, torch.distributed all-gather alone")
    ap.add_argument("--cpu-sample", type=int, default=None,
                    help="pos (=neg) sequences of the CPU baseline sample; default: the WHOLE workload for the headline "
                         "(c2: 5 000 + 5 000, ~90 s on 16 cores), 2 000 + 2 000 otherwise")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the short side measurements of gkmQC's own shape (peaks) and config 5 in the headline line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--check", action="store_true", help="compare the assembled matrix with a 1-GPU run (debug)")
    args = ap.parse_args(argv)
    w = WORKLOADS[args.workload]
    custom = False
    for i, name in enumerate(("n_pos", "n_neg", "length", "length_range", "kernel_type", "L", "k", "d")):
        if getattr(args, name) is None:
            setattr(args, name, w[i])
        elif getattr(args, name) != w[i] and not (name == "length_range" and tuple(getattr(args, name)) == w[i]):
            custom = True
    args.generator, args.label = w[8], ("custom (from %s)" % args.workload if custom else w[9])
    args.custom = custom
    if args.cpu_sample is None:
        args.cpu_sample = 5000 if (args.workload == "c2" and not custom) else 2000
    return args


def launch_ranks(args, argv, cmd=None, timeout=None, capture=None):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this
    process has not imported torch nor touched the GPU, and it never replaces itself), relay rank 0's
    stdout, fail if any rank fails.  All children are polled: the first one that exits non-zero ends the
    others at once (a rank that dies before the process group forms would otherwise leave its peers in the
    rendezvous until the driver's limit), and so does the overall timeout (GKM_BENCH_TIMEOUT, default 540 s).
    Every rank's stderr is relayed line by line behind a "[rank r]" prefix."""
    import socket
    import threading
    timeout = float(os.environ.get("GKM_BENCH_TIMEOUT", "540")) if timeout is None else timeout
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, os.path.abspath(__file__)] + argv if cmd is None else cmd
    procs, threads, out0 = [], [], []

    def relay(stream, r):
        for raw in iter(stream.readline, b""):
            sys.stderr.write("[rank %d] %s" % (r, raw.decode(errors="replace")))
            sys.stderr.flush()

    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                             stderr=subprocess.PIPE)
        procs.append(p)
        threads.append(threading.Thread(target=relay, args=(p.stderr, r), daemon=True))
        if r == 0:
            threads.append(threading.Thread(target=lambda out=p.stdout: out0.append(out.read()), daemon=True))
    for t in threads:
        t.start()
    t_end = time.time() + timeout
    why = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad or time.time() > t_end:
            why = ("rank %d exited with code %d" % (bad[0], codes[bad[0]])) if bad else \
                "no result after %.0f s" % timeout
            for p in procs:          # the exact processes started above, nothing else
                if p.poll() is None:
                    p.terminate()
            t_kill = time.time() + 5
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, t_kill - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            break
        time.sleep(0.1)
    for t in threads:
        t.join(timeout=5)
    codes = [p.returncode for p in procs]
    # rank 0's line is relayed even when a rank failed: a line whose parity check failed says so itself (exit code 3)
    text = out0[0].decode() if out0 else ""
    if why:   # a caller that parses stdout without looking at the exit code must still see that the run failed
        line = _last_json_line(text)
        if line is not None and not line.get("parity_failed"):
            line["ranks_failed"] = True
            line["ranks_failed_why"] = why
            text = json.dumps(line) + "\n"
    if capture is not None:      # (run_auto merges rank 0's line with the other assembly's)
        capture.append(text)
    else:
        sys.stdout.write(text)
        sys.stdout.flush()
    if why:
        sys.stderr.write("bench.py: %s; the other ranks were stopped (exit codes %s)\n" % (why, codes))
        return codes[0] if codes and codes[0] == 3 else 1
    return max(abs(c) for c in codes)


def _strip_flag(argv, flag, has_value=True):
    out, skip = [], 0
    for x in argv:
        if skip:
            skip -= 1
            continue
        if x == flag:
            skip = 1 if has_value else 0
            continue
        if has_value and x.startswith(flag + "="):
            continue
        out.append(x)
    return out


def _last_json_line(text):
    for ln in reversed((text or "").strip().splitlines()):
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                return json.loads(ln)
            except ValueError:
                continue
    return None


def run_cabi_child(args, argv):
    """The product's one-process multi-GPU entry (gkmhip_gram_allgather) in a FRESH child process with a timeout of its
    own: this process has not touched a GPU and does not replace itself.  -> (line dict or None, error string or None).
    GKM_BENCH_CABI_TIMEOUT (seconds, default 300) bounds it; the child is ended by its exact PID."""
    timeout = float(os.environ.get("GKM_BENCH_CABI_TIMEOUT", "300"))
    cmd = [sys.executable, os.path.abspath(__file__)] + _strip_flag(argv, "--assembly") + ["--assembly", "cabi"]
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR",
                        "MASTER_PORT") and not k.startswith("TORCHELASTIC_")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    t0 = time.time()
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    except OSError as e:
        return None, "cannot start the cabi child: %s" % e
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        p.kill()
        out, err = p.communicate()
        sys.stderr.write("[cabi] killed after %.0f s\n%s" % (timeout, err.decode(errors="replace")[-2000:]))
        return None, "gkmhip_gram_allgather child gave no result within %.0f s and was killed" % timeout
    tail = err.decode(errors="replace")
    for ln in tail.splitlines()[-40:]:
        sys.stderr.write("[cabi] %s\n" % ln)
    line = _last_json_line(out.decode(errors="replace"))
    if p.returncode != 0 or line is None:
        last = [x for x in tail.strip().splitlines() if x.strip()][-1:] or ["no output"]
        why = "parity check failed in the child" if (line or {}).get("parity_failed") else last[0][-300:]
        return line if (line or {}).get("parity_failed") else None, \
            "gkmhip_gram_allgather child exited with code %s after %.0f s: %s" % (p.returncode, time.time() - t0, why)
    return line, None


def brief_line(line):
    """What `also.torch_dist` (or `also.cabi`) keeps of a whole line."""
    if not line:
        return None
    rf = line.get("roofline") or {}
    return {"value": line.get("value"), "unit": line.get("unit"), "ms_per_step": line.get("ms_per_step"),
            "n_gpus": line.get("n_gpus"), "ranks": (line.get("config") or {}).get("ranks"),
            "transport": (line.get("config") or {}).get("transport"), "chunks": (line.get("config") or {}).get("chunks"),
            "row_sharding": (line.get("config") or {}).get("row_sharding"), "parity": line.get("parity"),
            "frac": rf.get("frac"), "kernel_ms_per_rank": rf.get("kernel_ms_per_rank"),
            "allgather_ms": rf.get("allgather_ms"), "assemble_ms": rf.get("assemble_ms"),
            "allgather_bytes_per_rank": rf.get("allgather_bytes_per_rank"),
            "allgather_GBps_per_rank": rf.get("allgather_GBps_per_rank")}


def merge_assemblies(cabi_line, cabi_err, torch_line, torch_err, torch_rc=0):
    """-> (the line to print, exit code).  Primary = the product's own entry (gkmhip_gram_allgather); the
    torch.distributed ranks are the cross-check and, when the primary CRASHES or is killed at its timeout, the fallback
    (exit code 0: the run is not lost).  A PARITY failure of either assembly is never silent: the other assembly's
    numbers are kept, but the line says `parity_failed` and the exit code is 3.  A torch launch that ended non-zero
    (torch_rc) is an error of its own: its line, if one was captured, is reported but never returns 0."""
    cabi_parity_failed = bool(cabi_line and cabi_line.get("parity_failed"))
    torch_parity_failed = bool(torch_line and torch_line.get("parity_failed"))
    torch_failed = bool(torch_rc) and not torch_parity_failed
    cabi_ok = cabi_line is not None and not cabi_err and not cabi_parity_failed
    if cabi_ok:
        out = cabi_line
        out["assembly"] = "cabi: gkmhip_gram_allgather (one process, one host thread per device)"
        also = brief_line(torch_line) if torch_line else {"error": torch_err or "no line"}
        if torch_line and torch_rc:
            also["ranks_exit_code"] = torch_rc
        out.setdefault("also", {})["torch_dist"] = also
        code = 0
        if torch_parity_failed:
            out["parity_failed"] = True
            out["parity_failed_in"] = "torch.distributed cross-check"
            code = 3
        elif torch_line is not None and torch_failed:
            out["torch_dist_error"] = torch_err or "the torch.distributed ranks ended with exit code %d" % torch_rc
            code = 1
        return out, code
    if torch_line is not None:
        out = torch_line
        out["assembly"] = "torch: one process per GPU, torch.distributed all_gather_into_tensor (FALLBACK)"
        out["cabi_error"] = cabi_err or "parity check failed on the gkmhip_gram_allgather matrix"
        if cabi_line:
            out.setdefault("also", {})["cabi"] = brief_line(cabi_line)
        if cabi_parity_failed:
            out["parity_failed"] = True
            out["parity_failed_in"] = "gkmhip_gram_allgather" + (" and torch.distributed" if torch_parity_failed else "")
            return out, 3
        if torch_parity_failed:
            return out, 3
        if torch_failed:
            out["ranks_failed"] = True
            out["torch_dist_error"] = torch_err or "the torch.distributed ranks ended with exit code %d" % torch_rc
            return out, 1
        return out, 0
    sys.stderr.write("bench.py: both assemblies failed: cabi: %s; torch: %s\n" % (cabi_err, torch_err))
    return None, 1


def run_auto(args, argv):
    """N > 1, --assembly auto: the product's one-process entry first (child of rank 0, while no rank has touched a
    GPU), then the torch.distributed ranks; one merged line."""
    torch_argv = _strip_flag(argv, "--assembly") + ["--assembly", "torch"]
    targs = parse_args(torch_argv)
    if "WORLD_SIZE" not in os.environ:
        # no launcher: this process orchestrates and never touches a GPU
        cabi_line, cabi_err = run_cabi_child(args, argv)
        captured = []
        rc = launch_ranks(targs, torch_argv, capture=captured)
        torch_line = _last_json_line(captured[0]) if captured else None
        torch_err = "the torch.distributed ranks failed (exit code %d)" % rc if rc else None
        out, code = merge_assemblies(cabi_line, cabi_err, torch_line, torch_err, torch_rc=rc)
        if out is not None:
            print(json.dumps(out), flush=True)
        return code
    # under torch.distributed.run: rank 0 runs the child first; the other ranks wait for its verdict in a file
    # (one node, by contract) before anybody imports torch or touches a GPU
    rank = int(os.environ.get("RANK", "0"))
    # The verdict file is keyed by this run (launcher pid, port, elastic run id), and a waiting rank only accepts a file
    # written AFTER it started waiting: a stale file of a crashed earlier run would let ranks != 0 touch their GPUs
    # while rank 0's child is still measuring on all of them.
    run_id = "".join(ch for ch in os.environ.get("TORCHELASTIC_RUN_ID", "none") if ch.isalnum())[:32]
    sync = os.path.join(tempfile.gettempdir(), "gkm_bench_cabi_%d_%s_%s.json"
                        % (os.getppid(), os.environ.get("MASTER_PORT", "0"), run_id))
    wait_s = float(os.environ.get("GKM_BENCH_CABI_TIMEOUT", "300")) + 60.0
    cabi_line = cabi_err = None
    t_start = time.time()
    if rank == 0:
        try:
            for stale in [sync] + glob.glob(sync + ".timeout.*"):
                if os.path.exists(stale):
                    os.unlink(stale)
            cabi_line, cabi_err = run_cabi_child(args, argv)
        finally:
            with open(sync + ".tmp", "w") as f:
                json.dump({"line": cabi_line, "error": cabi_err, "written_at": time.time()}, f)
            os.replace(sync + ".tmp", sync)
    else:
        def fresh():
            try:
                return os.path.getmtime(sync) >= t_start - 1.0
            except OSError:
                return False
        t_end = time.time() + wait_s
        while not fresh() and time.time() < t_end:
            time.sleep(0.2)
        if not fresh():   # reported in rank 0's line: this rank went ahead without the verdict
            sys.stderr.write("bench.py: rank %d waited %.0f s for rank 0's C-ABI measurement and goes ahead without it\n"
                             % (rank, wait_s))
            try:
                open(sync + ".timeout.%d" % rank, "w").close()
            except OSError:
                pass
    holder = []
    rc = run_rank(targs, emit=holder.append)
    if rank == 0:
        timed_out = sorted(int(x.rsplit(".", 1)[1]) for x in glob.glob(sync + ".timeout.*"))
        for x in [sync] + glob.glob(sync + ".timeout.*"):
            try:
                os.unlink(x)
            except OSError:
                pass
        torch_line = holder[0] if holder else None
        out, code = merge_assemblies(cabi_line, cabi_err, torch_line, None if torch_line else "rank 0 produced no line",
                                     torch_rc=rc if rc != 3 else 0)
        if out is not None:
            if timed_out:
                out["cabi_wait_timed_out_on_ranks"] = timed_out
            print(json.dumps(out), flush=True)
        return code
    return rc


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and args.assembly == "auto":
        return run_auto(args, argv)
    if args.gpus > 1 and args.assembly == "torch" and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)
    return run_rank(args)


def run_rank(args, emit=None):
    """One rank (or the one process of --assembly cabi).  emit: receives rank 0's line instead of stdout."""
    # stdout carries exactly one JSON line.  Libraries write there too (RCCL prints its version
    # banner on file descriptor 1 when stderr is not a file), so descriptor 1 is pointed at stderr
    # for the duration of the run and the line goes to a private copy of the real stdout.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    from gkmqc_amd import device, sharding

    cabi = args.assembly == "cabi" and args.gpus > 1
    if cabi and os.environ.get("GKM_BENCH_CABI_FAIL") == "1":   # tests: the fallback of --assembly auto
        raise SystemExit("injected failure of the cabi child (GKM_BENCH_CABI_FAIL)")
    world = 1 if cabi else int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0")) if not cabi else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if not cabi else 0
    # GKM_BENCH_FORCE_DIST=1: take the sharded path (process group, slabs, all-gather, permutation)
    # even with one rank -- exercises the real RCCL backend on a one-GPU box
    dist_on = world > 1 or os.environ.get("GKM_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:   # only the one-rank rehearsal comes without one
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not cabi and args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start the ranks with torch.distributed.run, or plainly as "
                         "`python bench.py --gpus N` (it spawns them itself)" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    # Rehearsal knobs (not used by the driver): GKM_BENCH_BACKEND=gloo runs the collective through
    # host memory, GKM_BENCH_SHARE_GPU=1 puts every rank on GPU 0 -- lets the N>1 path be exercised
    # on a one-GPU box.
    backend = os.environ.get("GKM_BENCH_BACKEND", "nccl")
    share_gpu = os.environ.get("GKM_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    ndev = torch.cuda.device_count()
    if (world > ndev or (cabi and args.gpus > ndev)) and not share_gpu:
        raise SystemExit("%d ranks but %d visible GPU(s); GKM_BENCH_SHARE_GPU=1 (with GKM_BENCH_BACKEND=gloo for "
                         "--assembly torch) rehearses the N > 1 path on one GPU" % (max(world, args.gpus), ndev))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist_on:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # synthetic problem (identical on every rank), resident in HBM before timing starts
    seqs = [device.encode(s) for s in make_problem(args)]
    n = len(seqs)
    stream = torch.cuda.current_stream().cuda_stream

    if cabi:
        import ctypes
        devices = [0] * args.gpus if share_gpu else list(range(args.gpus))
        ctxs = []
        for dv in devices:
            c = device.GramContext(args.kernel_type, args.L, args.k, args.d, 50, 50.0, 1.0, dv)
            c.set_kernel({"auto": 0, "direct": 1, "bitslice": 2}[args.kernel])
            with torch.cuda.device(dv):
                c.set_sequences(seqs, torch.cuda.current_stream().cuda_stream)
            ctxs.append(c)
        Ks = [torch.zeros((n, n), dtype=torch.float64, device=torch.device("cuda", dv)) for dv in devices]
        for dv in set(devices):
            torch.cuda.synchronize(dv)
        handles = (ctypes.c_void_p * len(ctxs))(*[c.handle for c in ctxs])
        outs = (ctypes.c_void_p * len(ctxs))(*[K.data_ptr() for K in Ks])
        lib = device.load()
        # chunks per rank: GKM_BENCH_CHUNKS, else what the library chooses from (n, ranks) (gkm_shard.h auto_chunks)
        chunks = max(1, int(os.environ["GKM_BENCH_CHUNKS"])) if os.environ.get("GKM_BENCH_CHUNKS") else \
            sharding.auto_chunks(n, args.gpus)
        ctx, full = ctxs[0], Ks[0]

        def step():
            rc = lib.gkmhip_gram_allgather(handles, len(ctxs), outs, n, 0, chunks)
            if rc:
                raise SystemExit("gkmhip_gram_allgather: " + lib.gkmhip_last_error().decode())
    else:
        ctx = device.GramContext(args.kernel_type, args.L, args.k, args.d, 50, 50.0, 1.0, local_rank)
        ctx.set_kernel({"auto": 0, "direct": 1, "bitslice": 2}[args.kernel])
        ctx.set_sequences(seqs, stream)

        # Row sharding: folded row blocks per rank; with more than one rank the rank's rows are cut
        # into interleaved chunks so that the RCCL all-gather of one chunk overlaps the kernel of the next.
        chunks = 1 if not dist_on else max(1, int(os.environ["GKM_BENCH_CHUNKS"])) if os.environ.get("GKM_BENCH_CHUNKS") \
            else sharding.auto_chunks(n, world)
        parts, _ = sharding.chunked_layout(n, world, rank, chunks)
        full = torch.zeros((n, n), dtype=torch.float64, device=dev)
        sq = torch.zeros(n, dtype=torch.float64, device=dev)
        if dist_on:
            # PACKED slabs (gkmqc_amd/sharding.py): row a of a chunk = a + 1 doubles, rows back to back, every chunk
            # padded to the largest one over all ranks -- half the bytes of full-width rows
            pe = sharding.packed_chunk_elems(n, world, chunks)
            row_off = [sharding.packed_row_offsets(parts[c]) for c in range(chunks)]
            slab = torch.zeros((chunks, pe), dtype=torch.float64, device=dev)
            gathered = torch.zeros((chunks, world * pe), dtype=torch.float64, device=dev)
            offset_of_row = torch.from_numpy(sharding.packed_gather_offsets(n, world, chunks)).to(dev)

        def compute(c, out_ptr, local, on=None):
            if not len(parts[c]):
                return
            if dist_on:
                ctx.gram_rows_packed(parts[c], out_ptr, row_off[c], stream if on is None else on)
            else:
                ctx.gram_rows(parts[c], out_ptr, n, None, 0, local, stream if on is None else on)

        # ONE side stream for the chunks' kernels (gkm_multi.hip compute_streams(): two launches on two streams run
        # concurrently and complete together, so chunk c's all-gather would hide behind nothing); the collective runs on
        # RCCL's own stream and overlaps the next chunk's kernel.  GKM_MULTI_STREAMS=two: rounds 2-5, for measurements.
        nside = 2 if os.environ.get("GKM_MULTI_STREAMS") == "two" else 1
        side = [torch.cuda.Stream(dev) for _ in range(nside)] if dist_on else None

        def step(probe=None):
            """probe: a dict that receives torch events around the collectives and the assembly (an extra,
            untimed step after the wall-clock region: the timed steps record nothing)."""
            if not dist_on:
                compute(0, full.data_ptr(), False)
                ctx.normalize(full.data_ptr(), n, sq.data_ptr(), False, stream)
                return
            main_s = torch.cuda.current_stream()
            pending = []
            for c in range(chunks):
                st = side[c % nside]
                if c < nside:
                    st.wait_stream(main_s)      # the previous step has finished reading slab / gathered
                ctx.set_scratch_slot(c & 1)     # chunks c and c+2 share a slot and a stream
                with torch.cuda.stream(st):
                    compute(c, slab[c].data_ptr(), True, st.cuda_stream)
                    if probe is not None:
                        ev = torch.cuda.Event(enable_timing=True)
                        ev.record(st)
                        probe.setdefault("ag0", []).append(ev)
                    if backend == "nccl":   # RCCL over xGMI, asynchronous: overlaps the next chunk's kernel
                        pending.append(dist.all_gather_into_tensor(gathered[c], slab[c], async_op=True))
                    else:                   # rehearsal through host memory
                        host = torch.empty(gathered[c].shape, dtype=gathered.dtype)
                        dist.all_gather_into_tensor(host, slab[c].cpu())
                        gathered[c].copy_(host)
                    if probe is not None:
                        if backend == "nccl":
                            pending[-1].wait()  # (the probe step only: makes this stream wait for the collective)
                        ev = torch.cuda.Event(enable_timing=True)
                        ev.record(st)
                        probe.setdefault("ag1", []).append(ev)
            ctx.set_scratch_slot(0)
            for w in pending:
                w.wait()
            for st in side:
                main_s.wait_stream(st)
            if probe is not None:
                probe["as0"] = torch.cuda.Event(enable_timing=True)
                probe["as0"].record(main_s)
            # un-permutation + normalisation in one pass over the gathered (packed: lds = 1, element offsets) slabs
            ctx.assemble_normalize(gathered.data_ptr(), 1, offset_of_row.data_ptr(), full.data_ptr(), n, sq.data_ptr(),
                                   False, stream)
            if probe is not None:
                probe["as1"] = torch.cuda.Event(enable_timing=True)
                probe["as1"].record(main_s)

    def barrier():
        if dist_on:
            dist.barrier()
        for dv in range(torch.cuda.device_count()) if cabi else [dev]:
            torch.cuda.synchronize(dv)

    for _ in range(args.warmup):
        step()
    barrier()
    allocs0 = device.load().gkmhip_allgather_alloc_count() if cabi else 0
    # N = 1: the Gram kernel's HIP events are kept per launch INSIDE the timed loop (gkmhip_kernel_timeline: no host
    # wait between the steps) and read after it, and every step is bracketed by events on the launch stream: the
    # line's kernel_ms and small_kernels_ms are parts of the very steps that ms_per_step times
    timeline = not dist_on and not cabi
    step_ev = []
    if timeline:
        ctx.kernel_timeline(True)
        step_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        if timeline:
            step_ev[i][0].record()
        step()
        if timeline:
            step_ev[i][1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    timed_kernel_ms = timed_span_ms = None
    if timeline:
        ksum, klaunches = ctx.kernel_timeline_ms()
        ctx.kernel_timeline(False)
        if ksum >= 0 and klaunches == args.steps:
            timed_kernel_ms = ksum / args.steps
            timed_span_ms = sum(a.elapsed_time(b) for a, b in step_ev) / args.steps
    allocs_timed = (device.load().gkmhip_allgather_alloc_count() - allocs0) if cabi else 0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # PARITY GATE: the matrix the last timed step left behind -- every device's copy (cabi), this rank's copy (torch)
    if cabi:
        parity_mine = [parity_of_device_matrix(args, K) for K in Ks]
    else:
        parity_mine = [parity_of_device_matrix(args, full)]
    if dist_on:
        gathered_par = [None] * world
        dist.all_gather_object(gathered_par, parity_mine[0])
        parity_mine = gathered_par
    parity = merge_parity(parity_mine)

    # dominant kernel: extra launches bracketed by HIP events on the launch stream (recorded inside
    # gkmhip_gram_rows), outside the wall-clock region
    durs, comparisons = [], 0.0
    per_rank = None     # N > 1: what every rank measured on its own stream (kernel, collective, assembly)
    if cabi:
        st = device.allgather_stats()   # HIP events of the last timed step, per host thread / device
        if st:
            per_rank = [{"kernel_ms": st["kernel_ms"][g], "comparisons": st["comparisons"][g],
                         "allgather_ms": st["transfer_ms"][g], "assemble_ms": st["assemble_ms"][g]} for g in range(st["ranks"])]
        rows0 = np.concatenate(sharding.chunked_layout(n, args.gpus, 0, 1)[0])
        buf = torch.zeros((len(rows0), n), dtype=torch.float64, device=dev)
        for _ in range(3):
            ctx.gram_rows(rows0, buf.data_ptr(), n, None, 0, True, stream)
            torch.cuda.synchronize(dev)
            durs.append(ctx.last_kernel_ms())
            comparisons = ctx.last_comparisons()
        del buf
    elif timed_kernel_ms is not None:
        durs.append(timed_kernel_ms)
        comparisons = ctx.last_comparisons()    # 2 n_a n_j summed over the (a, j<=a) pairs: one launch per step
    else:
        for _ in range(max(3, min(args.steps, 5))):
            ms, comparisons = 0.0, 0.0
            for c in range(chunks):
                if not len(parts[c]):
                    continue
                compute(c, slab[c].data_ptr() if dist_on else full.data_ptr(), dist_on)
                torch.cuda.synchronize(dev)
                ms += ctx.last_kernel_ms()
                comparisons += ctx.last_comparisons()   # 2 n_a n_j summed over this rank's (a, j<=a) pairs
            durs.append(ms)
    kern_ms = float(np.mean(durs))
    kname = ctx.last_kernel_name()
    if dist_on and not cabi:
        # one more, untimed, step with events around every collective and around the assembly
        barrier()
        probe = {}
        step(probe)
        barrier()
        mine = {"kernel_ms": kern_ms, "comparisons": comparisons,
                "allgather_ms": float(sum(a0.elapsed_time(a1) for a0, a1 in zip(probe["ag0"], probe["ag1"]))),
                "assemble_ms": float(probe["as0"].elapsed_time(probe["as1"]))}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if args.check:  # every rank recomputes the whole matrix alone and compares bit for bit
        step()
        barrier()
        ref = torch.zeros((n, n), dtype=torch.float64, device=dev)
        sq2 = torch.zeros(n, dtype=torch.float64, device=dev)
        ctx.gram_rows(np.arange(n), ref.data_ptr(), n, None, 0, False, stream)
        ctx.normalize(ref.data_ptr(), n, sq2.data_ptr(), False, stream)
        torch.cuda.synchronize(dev)
        same = bool((torch.tril(ref) == torch.tril(full)).all().item())
        print("rank %d: assembled matrix identical to single-GPU matrix: %s" % (rank, same), file=sys.stderr, flush=True)
        assert same

    n_gpus = args.gpus if cabi else world
    pairs = n * (n - 1) / 2
    sec_per_step = elapsed / args.steps
    if cabi:
        sharding_note = ("one process, %d host threads; folded row blocks in %d chunks, %s all-gather overlapped with "
                         "the next chunk (gkmhip_gram_allgather)" % (args.gpus, chunks, device.load().gkmhip_last_transport().decode()))
    elif dist_on:
        sharding_note = ("one process per GPU; folded row blocks in %d interleaved chunks, %s all-gather overlapped with the "
                         "next chunk" % (chunks, "RCCL" if backend == "nccl" else backend))
    else:
        sharding_note = "single GPU"
    desc = ("%d pos + %d neg x %s bp %s, kernel type %d, L=%d k=%d d=%d, M=50 H=50; full lower-triangular Gram matrix + "
            "normalisation" % (args.n_pos, args.n_neg,
                               ("%d-%d" % tuple(args.length_range)) if args.length_range else str(args.length),
                               "peak-like synthetic DNA (gkmqc_amd.synth.make_peak_sequences seeds 11/12)"
                               if args.generator == "peaks" else "iid ACGT (splitmix64 seeds 1/2)",
                               args.kernel_type, args.L, args.k, args.d))
    out = {
        "metric": "gkm kernel pairs/sec (N=%dk, %d bp, L=%d,k=%d,d=%d)" % (n // 1000, args.length, args.L, args.k, args.d)
        if n % 1000 == 0 else "gkm kernel pairs/sec (N=%d, %d bp, L=%d,k=%d,d=%d)" % (n, args.length, args.L, args.k, args.d),
        "value": pairs / sec_per_step,
        "unit": "pairs/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": sec_per_step * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": args.label + ": " + desc, "n_sequences": n, "row_sharding": sharding_note, "kernel": kname,
                   # ranks = communicator size actually created; transport = what moved the slabs
                   "ranks": len(per_rank) if per_rank else (args.gpus if cabi else dist.get_world_size() if dist_on else 1),
                   "transport": (device.load().gkmhip_last_transport().decode() if cabi else
                                 ("rccl" if backend == "nccl" else backend) if dist_on else "none"),
                   "chunks": chunks,
                   "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("GKM_")}},
    }
    # the parity gate: a matrix that is not the reference's, on any copy, sets the number aside
    out["parity"] = parity
    parity_failed = parity is not None and parity.get("ok") is False
    if parity_failed:
        out["parity_failed"] = True
        out["value_set_aside"] = out["value"]
        out["value"] = None
    if rank == 0:
        kern_s = kern_ms * 1e-3
        algorithmic = comparisons * OPS_PER_COMPARISON / kern_s / 1e9
        pmc, pmc_src = (None, "custom problem or forced kernel: no PMC summary applies")
        if not args.custom and args.kernel == "auto":
            pmc, pmc_src = pmc_summary(args.workload)     # taken on ONE GPU: the whole triangle in one launch
        insts = pmc["per_launch"].get("SQ_INSTS_VALU") if pmc else None
        traffic = pmc.get("hbm_bytes_per_launch") if pmc else None
        multi = None
        if n_gpus == 1:
            executed = (insts * 64 / kern_s / 1e9) if insts else None
            ipc = (insts * 64 / comparisons) if insts else None
        else:
            multi = multi_gpu_roofline(
                insts, per_rank, sec_per_step, n_gpus,
                bytes_per_rank=(device.load().gkmhip_allgather_bytes_per_rank() if cabi else
                                sharding.allgather_bytes_per_rank(n, n_gpus, chunks)),
                bytes_full_width=sharding.allgather_bytes_per_rank(n, n_gpus, chunks, packed=False))
            executed, ipc, traffic = multi["achieved"], multi["executed_insts_per_comparison"], None
        peak_meas, peak_src = measured_valu_peak()
        out["roofline"] = {
            "bound": "valu",
            # EXECUTED lane-ops of the dominant kernel (rocprofv3 SQ_INSTS_VALU x 64 lanes, per launch, from the
            # committed summary taken on this very kernel source and workload) over its live duration
            "achieved": executed, "peak": PEAK_INT32_GOPS, "unit": "Gop/s",
            "frac": (executed / (PEAK_INT32_GOPS * n_gpus)) if executed else None,
            "peak_measured": peak_meas, "peak_measured_source": peak_src,
            "frac_of_measured_peak": (executed / (peak_meas * n_gpus)) if executed and peak_meas else None,
            "traffic": traffic, "pmc_source": pmc_src,
            "hbm_achieved_GBps": (traffic / kern_s / 1e9) if traffic else None,
            "kernel": kname, "kernel_ms": kern_ms,
            # N = 1: kernel_ms is the mean over the TIMED steps' own launches; small_kernels_ms is the rest of a step's
            # span on the launch stream (row tables, untile, self norms, normalise, the gaps between them)
            "kernel_ms_source": ("HIP events of the timed steps' launches (gkmhip_kernel_timeline)" if timed_kernel_ms is not None
                                 else "extra launches after the timed region, one device sync each"),
            "small_kernels_ms": (timed_span_ms - timed_kernel_ms) if timed_kernel_ms is not None else None,
            "step_span_ms": timed_span_ms,
            "comparisons_per_launch": comparisons,
            "comparisons_per_s": comparisons / kern_s,
            "executed_insts_per_comparison": ipc,
            # the op model of SURVEY.md §8(d) (6 int32 ops per l-mer comparison): what a comparison-by-comparison
            # kernel would have to execute; the bit-sliced kernel executes ~10x fewer, so this "fraction" exceeds 1
            "algorithmic_ops_per_comparison": OPS_PER_COMPARISON,
            "algorithmic_Gops": algorithmic, "algorithmic_frac": algorithmic / PEAK_INT32_GOPS,
            "note": "Integer-VALU bound, neither HBM nor MFMA (SURVEY.md §8(d)). frac = executed VALU lane-ops / peak "
                    "(256 CU x 4 SIMD x 32 lanes x 2.4 GHz); null when the committed PMC summary was not taken on this "
                    "kernel source + workload. kernel_ms: HIP events on the launch stream. HBM traffic is incidental.",
        }
        if n_gpus == 1 and not args.custom and args.kernel == "auto":
            im = issue_model(args.workload)
            if im and im.get("profiled_run_clock_GHz") and im.get("profiled_run_kernel_ms") and kern_ms > 0:
                # same instructions, same issue cycles: the clock THIS run held, by proportion
                im["implied_clock_GHz_this_run"] = im["profiled_run_clock_GHz"] * im["profiled_run_kernel_ms"] / kern_ms
            out["roofline"]["issue_model"] = im
        if n_gpus > 1:
            rf = out["roofline"]
            rf.update(multi)
            rf["note"] += (" N > 1: achieved = the one-GPU launch's executed lane-ops / the max-over-ranks step time (kernels, "
                           "collectives, assembly), frac = that / (N x peak); kernel_ms, comparisons_per_launch and "
                           "comparisons_per_s are rank 0's own kernel; allgather_ms = sum over the %d chunks of the "
                           "collective's time on its stream (HIP events), which overlaps the next chunk's kernel; "
                           "allgather_bytes_per_rank = what every rank receives from its peers per matrix (packed row "
                           "slabs, row a = a + 1 doubles)." % chunks)
            if cabi:
                rf["allgather_hipmalloc_calls_in_timed_region"] = allocs_timed
        # the side measurements must never cost the line itself: a failure is reported in their place
        if n_gpus == 1 and args.workload == "c2" and not args.custom and not args.no_also:
            # gkmQC's own shape (bin/gkmqc.py:150-154,181-185: 600 bp, wgkm L=10 k=6 d=3) and config 5 (ragged,
            # L=12 d=4), three steps each, so that their numbers are timed by whoever runs the headline line
            try:
                ctx.close()
                full = None
                torch.cuda.empty_cache()
                out["also"] = {name: side_workload(name, dev) for name in ("peaks", "c5")}
            except Exception as e:   # noqa: BLE001
                out["also"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if n_gpus == 1 and not args.no_end_to_end:
            ctx.close()
            full = None
            torch.cuda.empty_cache()
            try:
                out["end_to_end"] = end_to_end(args, dev)
            except Exception as e:   # noqa: BLE001
                out["end_to_end"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if n_gpus == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args)
                cb = out["cpu_baseline"]
                if cb.get("headline_workload") and cb.get("value") and out["value"]:
                    # BASELINE.md holds no published number for this metric (the reference ships none); the one
                    # baseline there is: its own CPU path on this box's host cores, same files, same N
                    out["vs_baseline"] = out["value"] / cb["value"]
                    out["vs_baseline_is"] = ("value / cpu_baseline.value: the reference's CPU path (%s, %d cores) on the "
                                             "same box and the same workload; BASELINE.md has no published number"
                                             % (cb["kind"], cb["cores"]))
            except Exception as e:   # noqa: BLE001
                out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if emit is not None:
            emit(out)
        else:
            print(json.dumps(out), file=real_stdout, flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if cabi:
        for c in ctxs:
            c.close()
    else:
        ctx.close()
    # descriptor 1 is the real stdout again: with `emit` the caller prints the (merged) line itself
    sys.stdout.flush()
    os.dup2(real_stdout.fileno(), 1)
    real_stdout.close()
    return 3 if parity_failed else 0


if __name__ == "__main__":
    sys.exit(main())
