#!/usr/bin/env python3
"""What ONE rank's step of the G-GPU Gram matrix costs, measured on ONE GPU (no round has had a node):

    python3 tools/rank_alone.py [--workload c2] [--ranks 2 4 8] [--reps 5] [--chunks 0] [--json out.json]

For every G: G contexts on device 0 run the product's gkmhip_gram_allgather once (peer copies: fills every rank's
gathered slabs; the assembled matrix is digest-checked against the reference's), then each rank g runs its step ALONE on
the device through gkmhip_gram_rank_alone (gkm_multi.hip: the very rank_thread of the product -- same chunks, streams,
packed slabs, assembly; only the transfer is a device copy of its own slab) `reps` times.  Reported per (G, g):
  wall       host clock, first enqueue -> all streams complete (what a rank contributes to the step, transfer excluded)
  kernels    HIP events around each chunk's launch group (tables, row planes, Gram kernel, untile), summed
  gram       the Gram kernels alone (gkmhip_kernel_timeline), summed over the chunks
  assemble   un-permute + normalise of the whole matrix
  ideal      the one-GPU kernel time of the same run / G
and the matrix every alone-step leaves in K is digest-checked too.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--ranks", type=int, nargs="*", default=[2, 4, 8])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--chunks", type=int, default=0)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    import torch
    import bench
    from gkmqc_amd import device, sharding
    a = bench.parse_args(["--workload", args.workload])
    seqs = [device.encode(s) for s in bench.make_problem(a)]
    n = len(seqs)
    lib = device.load()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    res = {"workload": args.workload, "n": n, "per_G": {}}

    # the one-GPU reference point of THIS run (same box, same clock)
    ctx = device.GramContext(a.kernel_type, a.L, a.k, a.d, 50, 50.0, 1.0, 0)
    ctx.set_sequences(seqs, stream)
    G1 = torch.zeros((n, n), dtype=torch.float64, device=dev)
    one = []
    for i in range(4):
        t0 = time.perf_counter()
        ctx.gram_rows(np.arange(n), G1.data_ptr(), n, None, 0, False, stream)
        ctx.normalize(G1.data_ptr(), n, None, False, stream)
        torch.cuda.synchronize(dev)
        if i:
            one.append(((time.perf_counter() - t0) * 1e3, ctx.last_kernel_ms()))
    ctx.close()
    del G1
    step1 = min(x[0] for x in one)
    kern1 = min(x[1] for x in one)
    res["one_gpu"] = {"step_ms": step1, "gram_kernel_ms": kern1}
    print("one GPU: step %.2f ms, Gram kernel %.2f ms" % (step1, kern1), flush=True)

    for G in args.ranks:
        ctxs, Ks = [], []
        for g in range(G):
            c = device.GramContext(a.kernel_type, a.L, a.k, a.d, 50, 50.0, 1.0, 0)
            c.set_sequences(seqs, stream)
            ctxs.append(c)
            Ks.append(torch.zeros((n, n), dtype=torch.float64, device=dev))
        torch.cuda.synchronize(dev)
        handles = (ctypes.c_void_p * G)(*[c.handle for c in ctxs])
        outs = (ctypes.c_void_p * G)(*[K.data_ptr() for K in Ks])
        rc = lib.gkmhip_gram_allgather(handles, G, outs, n, 0, args.chunks)
        if rc:
            raise SystemExit("gkmhip_gram_allgather failed: " + lib.gkmhip_last_error().decode())
        par = bench.parity_of_device_matrix(a, Ks[0])
        chunks = args.chunks or sharding.auto_chunks(n, G)
        per = {"chunks": chunks, "transport_of_the_fill": lib.gkmhip_last_transport().decode(),
               "allgather_parity_ok": par.get("ok"), "ranks": {}}
        out6 = np.zeros(6)
        for g in range(G):
            rows, times, spans = [], [], []
            for rep in range(args.reps + 1):
                Ks[g].zero_()
                torch.cuda.synchronize(dev)
                ctxs[g].kernel_timeline(True)
                rc = lib.gkmhip_gram_rank_alone(ctxs[g].handle, g, G, args.chunks, Ks[g].data_ptr(), n, 0, out6.ctypes.data)
                if rc:
                    raise SystemExit("gkmhip_gram_rank_alone failed: " + lib.gkmhip_last_error().decode())
                gram, launches = ctxs[g].kernel_timeline_ms()
                sp = np.zeros(2 * 16)
                nsp = lib.gkmhip_kernel_timeline_spans(ctxs[g].handle, sp.ctypes.data, len(sp)) if hasattr(lib, "gkmhip_kernel_timeline_spans") else 0
                ctxs[g].kernel_timeline(False)
                ct = np.zeros(4 * 16)
                nct = lib.gkmhip_allgather_chunk_times(g, ct.ctypes.data, len(ct)) if hasattr(lib, "gkmhip_allgather_chunk_times") else 0
                if rep:
                    rows.append((out6[0], out6[1], gram, out6[2], out6[3], launches))
                    times.append(ct[:nct].reshape(-1, 4).copy())
                    spans.append(sp[:nsp].reshape(-1, 2).copy())
            ok = bench.parity_of_device_matrix(a, Ks[g]).get("ok")
            arr = np.array(rows)
            best = arr[arr[:, 0].argmin()]
            per["ranks"][g] = {"wall_ms_min": float(arr[:, 0].min()), "wall_ms_median": float(np.median(arr[:, 0])),
                               "kernels_ms": float(best[1]), "gram_ms": float(best[2]), "copy_in_ms": float(best[3]),
                               "assemble_ms": float(best[4]), "launches": int(best[5]), "comparisons": float(out6[4]),
                               "parity_ok": ok,
                               "chunk_launch_group_start_end_ms": times[int(arr[:, 0].argmin())][:, :2].tolist(),
                               "chunk_slab_copy_start_end_ms": times[int(arr[:, 0].argmin())][:, 2:].tolist(),
                               "gram_kernel_start_end_ms": spans[int(arr[:, 0].argmin())].tolist()}
            print("G=%d rank %d: wall %.2f ms (median %.2f), launch groups %.2f, Gram kernels %.2f (%d launches), copy-in %.2f, "
                  "assemble %.2f; chunks ran %s ms (their Gram kernels %s); ideal %.2f (one-GPU kernel / G); matrix %s"
                  % (G, g, arr[:, 0].min(), np.median(arr[:, 0]), best[1], best[2], best[5], best[3], best[4],
                     ", ".join("%.2f-%.2f (its slab copied %.2f-%.2f)" % tuple(x) for x in times[int(arr[:, 0].argmin())]),
                     ", ".join("%.2f-%.2f" % (x[0], x[1]) for x in spans[int(arr[:, 0].argmin())]), kern1 / G,
                     "identical to the reference's" if ok else "NOT CHECKED" if ok is None else "WRONG"), flush=True)
        walls = [v["wall_ms_min"] for v in per["ranks"].values()]
        per["max_wall_ms"] = max(walls)
        per["ideal_ms"] = kern1 / G
        per["over_ideal"] = max(walls) / (kern1 / G)
        res["per_G"][G] = per
        print("G=%d: slowest rank alone %.2f ms = %.3f x (one-GPU kernel / G = %.2f ms), chunks %d"
              % (G, max(walls), per["over_ideal"], kern1 / G, chunks), flush=True)
        for c in ctxs:
            c.close()
        del Ks, ctxs
        lib.gkmhip_release_comms()
        torch.cuda.empty_cache()
    if args.json:
        json.dump(res, open(args.json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
