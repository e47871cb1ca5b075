#!/usr/bin/env python3
"""bench.py -- gkm kernel pairs/sec (N=10k, 300 bp, L=11, k=7, d=3) on N GPUs of one node.

One "step" = one complete pass of the hot path over the synthetic problem with the
sequences already resident in HBM: build the per-call row tables, run the Gram kernel for
this rank's rows, all-gather the row slabs over RCCL (world_size > 1), normalise the
assembled matrix (division by the self norms, unit diagonal) on every rank.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `value` = N(N-1)/2 pairs / (max-over-ranks seconds per step).
The total work is fixed as the GPU count grows (the N x N matrix is sharded by row block),
so `scaling` is "strong".
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Integer-VALU peak of MI355X: 256 CUs x 4 SIMDs x 32 lanes/clk x 2.4 GHz = 78 643 Gop/s.
# (tools/valu_peak.hip measures 65-72 Tera lane-ops/s for full-rate VOP2 ops on the box, i.e. a
# wave64 v_xor_b32 every ~2.2 cycles per SIMD; SURVEY.md §8(d) assumed 16 lanes/clk = 39 321.)
PEAK_INT32_GOPS = 256 * 4 * 32 * 2.4
OPS_PER_COMPARISON = 6            # op model of SURVEY.md §8(d): xor, shift, or, and, popcount, compare


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


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary
    (separate FETCH_SIZE / WRITE_SIZE passes, tools/collect_profiles.sh); None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        return d.get("hbm_bytes_per_launch"), os.path.basename(files[-1])
    except Exception:
        return None, None


def measured_valu_insts():
    """SQ_INSTS_VALU (wave instructions) per launch of the dominant kernel from the same summary."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    try:
        return json.load(open(files[-1]))["per_launch"]["SQ_INSTS_VALU"]
    except Exception:
        return None


def cpu_baseline(args, L, k, d, kernel_type):
    """The reference's own CPU path (oracle/_ref, unmodified sources built by oracle/Makefile)
    timed on this box's host cores on a bounded sample of the same workload; falls back to
    the C restatement (kind "port") if the reference build did not travel."""
    from gkmqc_amd import synth
    from oracle import oracle as O
    cores = host_cores()
    npos = nneg = args.cpu_sample
    tmp = tempfile.mkdtemp(prefix="gkm_bench_")
    pf, nf = os.path.join(tmp, "p.fa"), os.path.join(tmp, "n.fa")
    synth.write_problem(pf, nf, npos, nneg, args.length)
    n = npos + nneg
    if O.have_ref():
        kind, fn = "reference", O.ref_pywrapper
    else:
        kind, fn = "port", O.oracle_pywrapper
        npos = nneg = min(args.cpu_sample, 150)   # brute-force port: keep it to seconds
        synth.write_problem(pf, nf, npos, nneg, args.length)
        n = npos + nneg
    opt = O.make_opt(kernel_type, L, k, d, 50, 50.0, 1.0, pf, nf, nthreads=cores, verbosity=0)
    t0 = time.time()
    rc, _, _, _ = fn(opt, n)
    wall = time.time() - t0
    assert rc == 0
    return {"value": (n * (n - 1) / 2) / wall, "unit": "pairs/s", "cores": cores, "kind": kind,
            "sample": "%d+%d x %d bp synthetic, same parameters, whole gkm_main_pywrapper call "
                      "(FASTA read + tree + rows), %d row threads, %.1f s wall; the reference's "
                      "pairs/s rises with N (N=10k on 8 cores: 218 s = 229 k pairs/s, BASELINE.md)"
                      % (npos, nneg, args.length, cores, wall)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-pos", type=int, default=5000)
    ap.add_argument("--n-neg", type=int, default=5000)
    ap.add_argument("--length", type=int, default=300)
    ap.add_argument("--length-range", type=int, nargs=2, default=None, help="uniform random lengths (config 5: 150 600)")
    ap.add_argument("--kernel-type", type=int, default=4)
    ap.add_argument("-L", type=int, default=11)
    ap.add_argument("-k", type=int, default=7)
    ap.add_argument("-d", type=int, default=3)
    ap.add_argument("--kernel", default="auto", choices=["auto", "direct", "bitslice"])
    ap.add_argument("--cpu-sample", type=int, default=1500, help="pos (=neg) sequences of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true", help="compare the assembled matrix with a 1-GPU run (debug)")
    args = ap.parse_args()

    # stdout carries exactly one JSON line.  Libraries write there too (RCCL prints its version
    # banner on file descriptor 1 when stderr is not a file), so descriptor 1 is pointed at stderr
    # for the duration of the run and the line goes to a private copy of the real stdout.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from gkmqc_amd import device, sharding
    from gkmqc_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    assert args.gpus == world, "--gpus must equal WORLD_SIZE (launch with torch.distributed.run)"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    # Rehearsal knobs (not used by the driver): GKM_BENCH_BACKEND=gloo runs the collective through
    # host memory, GKM_BENCH_SHARE_GPU=1 puts every rank on GPU 0 -- lets the N>1 path be exercised
    # on a one-GPU box.
    backend = os.environ.get("GKM_BENCH_BACKEND", "nccl")
    if os.environ.get("GKM_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist_on:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # synthetic problem (identical on every rank), resident in HBM before timing starts
    lr = tuple(args.length_range) if args.length_range else None
    seqs = [device.encode(s) for s in synth.make_sequences(1, args.n_pos, args.length, lr) +
            synth.make_sequences(2, args.n_neg, args.length, lr)]
    n = len(seqs)
    ctx = device.GramContext(args.kernel_type, args.L, args.k, args.d, 50, 50.0, 1.0, local_rank)
    ctx.set_kernel({"auto": 0, "direct": 1, "bitslice": 2}[args.kernel])
    stream = torch.cuda.current_stream().cuda_stream
    ctx.set_sequences(seqs, stream)

    # Row sharding: folded row blocks per rank; with more than one rank the rank's rows are cut
    # into interleaved chunks so that the RCCL all-gather of one chunk overlaps the kernel of the next.
    chunks = max(1, int(os.environ.get("GKM_BENCH_CHUNKS", "4"))) if dist_on else 1
    parts, pc = sharding.chunked_layout(n, world, rank, chunks)
    full = torch.zeros((n, n), dtype=torch.float64, device=dev)
    sq = torch.zeros(n, dtype=torch.float64, device=dev)
    if dist_on:
        slab = torch.zeros((chunks, pc, n), dtype=torch.float64, device=dev)
        gathered = torch.zeros((chunks, world * pc, n), dtype=torch.float64, device=dev)
        slot_of_row = torch.from_numpy(sharding.chunked_gather_index(n, world, chunks)).to(dev)

    def compute(c, out_ptr, local, on=None):
        if len(parts[c]):
            ctx.gram_rows(parts[c], out_ptr, n, None, 0, local, stream if on is None else on)

    # Two side streams, chunks alternate between them: the kernel of chunk c+1 fills the CUs that the
    # drain of chunk c leaves idle, and its all-gather overlaps as before.
    side = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)] if dist_on else None

    def step():
        if not dist_on:
            compute(0, full.data_ptr(), False)
        else:
            main = torch.cuda.current_stream()
            pending = []
            for c in range(chunks):
                st = side[c & 1]
                if c < 2:
                    st.wait_stream(main)        # the previous step has finished reading slab / gathered
                ctx.set_scratch_slot(c & 1)     # chunks c and c+2 share a slot and a stream
                with torch.cuda.stream(st):
                    compute(c, slab[c].data_ptr(), True, st.cuda_stream)
                    if backend == "nccl":   # RCCL over xGMI, asynchronous: overlaps the next chunk's kernel
                        pending.append(dist.all_gather_into_tensor(gathered[c], slab[c], async_op=True))
                    else:                   # rehearsal through host memory
                        host = torch.empty(gathered[c].shape, dtype=gathered.dtype)
                        dist.all_gather_into_tensor(host, slab[c].cpu())
                        gathered[c].copy_(host)
            ctx.set_scratch_slot(0)
            for w in pending:
                w.wait()
            main.wait_stream(side[0])
            main.wait_stream(side[1])
            torch.index_select(gathered.view(chunks * world * pc, n), 0, slot_of_row, out=full)
        ctx.normalize(full.data_ptr(), n, sq.data_ptr(), False, stream)

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel: one extra launch bracketed by HIP events on the launch stream
    # (recorded inside gkmhip_gram_rows), outside the wall-clock region
    durs, comparisons = [], 0.0
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

    if args.check:  # every rank recomputes the whole matrix alone and compares bit for bit
        step()
        torch.cuda.synchronize(dev)
        ref = torch.zeros((n, n), dtype=torch.float64, device=dev)
        ctx.gram_rows(np.arange(n), ref.data_ptr(), n, None, 0, False, stream)
        ctx.normalize(ref.data_ptr(), n, sq.data_ptr(), False, stream)
        torch.cuda.synchronize(dev)
        same = bool((torch.tril(ref) == torch.tril(full)).all().item())
        print("rank %d: assembled matrix identical to single-GPU matrix: %s" % (rank, same), file=sys.stderr, flush=True)
        assert same

    shape = (args.n_pos, args.n_neg, args.length, lr, args.L, args.k, args.d)
    WORKLOAD_LABEL = {(5000, 5000, 300, None, 11, 7, 3): "configs[1]", (200, 200, 300, None, 10, 6, 3): "configs[0]",
                      (10000, 10000, 300, None, 11, 7, 3): "configs[2]"}.get(shape, "custom")
    pairs = n * (n - 1) / 2
    sec_per_step = elapsed / args.steps
    out = {
        "metric": "gkm kernel pairs/sec (N=%dk, %d bp, L=%d,k=%d,d=%d)" % (n // 1000, args.length, args.L, args.k, args.d)
        if n % 1000 == 0 else "gkm kernel pairs/sec (N=%d, %d bp, L=%d,k=%d,d=%d)" % (n, args.length, args.L, args.k, args.d),
        "value": pairs / sec_per_step,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": sec_per_step * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": WORKLOAD_LABEL + ": %d pos + %d neg x %d bp iid ACGT (splitmix64 seeds 1/2), kernel type %d, "
                               "L=%d k=%d d=%d, M=50 H=50; full lower-triangular Gram matrix + normalisation"
                               % (args.n_pos, args.n_neg, args.length, args.kernel_type, args.L, args.k, args.d),
                   "n_sequences": n, "row_sharding": ("folded row blocks in %d interleaved chunks, RCCL all-gather overlapped with the next chunk" % chunks) if dist_on else "single GPU",
                   "kernel": kname},
    }
    if rank == 0:
        achieved = comparisons * OPS_PER_COMPARISON / (kern_ms * 1e-3) / 1e9
        traffic, traffic_src = (None, None)
        insts = None
        if world == 1 and (args.n_pos, args.n_neg, args.length, args.L, args.k, args.d, args.kernel_type) == (5000, 5000, 300, 11, 7, 3, 4):
            traffic, traffic_src = measured_traffic()   # PMC numbers were taken on exactly this workload
            insts = measured_valu_insts()
        out["roofline"] = {
            "bound": "valu",
            "achieved": achieved, "peak": PEAK_INT32_GOPS, "unit": "Gop/s", "frac": achieved / PEAK_INT32_GOPS,
            "traffic": traffic, "traffic_source": traffic_src,
            "kernel": kname, "kernel_ms": kern_ms, "comparisons_per_launch": comparisons,
            "hbm_achieved_GBps": (traffic / (kern_ms * 1e-3) / 1e9) if traffic else None,
            # executed (not algorithmic) VALU work: rocprofv3 SQ_INSTS_VALU x 64 lanes over the live kernel time
            "executed_valu_Gops": (insts * 64 / (kern_ms * 1e-3) / 1e9) if insts else None,
            "executed_frac_of_peak": (insts * 64 / (kern_ms * 1e-3) / 1e9 / PEAK_INT32_GOPS) if insts else None,
            "executed_insts_per_comparison": (insts * 64 / comparisons) if insts else None,
            "note": "This path is integer-VALU bound, neither HBM nor MFMA (SURVEY.md §8(d)); bound says so. "
                    "achieved = ALGORITHMIC ops: 6 int32 ops per l-mer comparison (SURVEY op model) x "
                    "comparisons_per_launch (2 n_a n_j per pair, this rank's pairs) / kernel_ms (HIP events on the "
                    "launch stream). peak = 256 CU x 4 SIMD x 32 lanes x 2.4 GHz. The bit-sliced kernel EXECUTES "
                    "about 0.56 VALU instructions per comparison instead of 6, which is why frac exceeds 1: see DESIGN.md for "
                    "the executed-instruction utilisation from rocprofv3 (profiles/). HBM traffic is incidental.",
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.L, args.k, args.d, args.kernel_type)
        print(json.dumps(out), file=real_stdout, flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
