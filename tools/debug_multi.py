#!/usr/bin/env python3
"""Debug helper: one-process multi-context assembly vs the single-context matrix at full size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gkmqc_amd import device
for wl, extra in (("c2", []), ("peaks", [])):
    a = bench.parse_args(["--workload", wl] + extra)
    seqs = [device.encode(s) for s in bench.make_problem(a)]
    one = device.gram_matrix(seqs, a.kernel_type, a.L, a.k, a.d)["K"]
    for trial in range(2):
        res = device.gram_matrix_multi(seqs, a.kernel_type, a.L, a.k, a.d, devices=[0, 0])
        for g, K in enumerate(res["K"]):
            bad = (K != one) | torch.isnan(K)
            rows = torch.nonzero(bad.any(dim=1)).flatten().cpu().numpy()
            print(wl, "trial", trial, "copy", g, "nan", int(torch.isnan(K).sum()), "bad cells", int(bad.sum()),
                  "bad rows", len(rows), rows[:10], rows[-5:] if len(rows) else "", flush=True)
    del one
