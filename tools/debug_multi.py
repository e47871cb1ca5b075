#!/usr/bin/env python3
"""Stress of the one-process multi-context assembly (GPU box): 2, 3 and 5 contexts on device 0 against the
single-context matrix, full-size workloads, several trials in ONE process (contexts created and destroyed each
time, so pooled host buffers and scratch are reused across calls)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from gkmqc_amd import device  # noqa: E402

bad_total = 0
for wl in ("c2", "peaks", "c5"):
    a = bench.parse_args(["--workload", wl])
    seqs = [device.encode(s) for s in bench.make_problem(a)]
    one = device.gram_matrix(seqs, a.kernel_type, a.L, a.k, a.d)["K"]
    for nctx in (2, 5, 3):
        for trial in range(2):
            res = device.gram_matrix_multi(seqs, a.kernel_type, a.L, a.k, a.d, devices=[0] * nctx, chunks=4 if trial else 3)
            for g, K in enumerate(res["K"]):
                bad = (K != one) | torch.isnan(K)
                nb = int(bad.sum())
                bad_total += nb
                print(wl, "contexts", nctx, "trial", trial, "copy", g, "bad cells", nb, "%.0f ms" % res["ms"], flush=True)
            del res
    del one
print("TOTAL BAD CELLS", bad_total)
sys.exit(1 if bad_total else 0)
