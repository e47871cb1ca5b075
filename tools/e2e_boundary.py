#!/usr/bin/env python3
"""End-to-end timing THROUGH THE DROP-IN BOUNDARY, driven exactly like the reference's caller
(scripts/gkmsvm.py:67-99): FASTA files on disk -> gkm_main_pywrapper -> a 15000x15000 zeroed
numpy matrix addressed through row pointers -> crop + np.maximum(K, K.T).

    python tools/e2e_boundary.py [--n-pos 5000 --n-neg 5000 --length 300 -t 4 -L 11 -k 7 -d 3]
    python tools/e2e_boundary.py --config c3|c5
"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gkmqc_amd import device, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n-pos", type=int, default=5000)
ap.add_argument("--n-neg", type=int, default=5000)
ap.add_argument("--length", type=int, default=300)
ap.add_argument("--length-range", type=int, nargs=2, default=None)
ap.add_argument("-t", type=int, default=4)
ap.add_argument("-L", type=int, default=11)
ap.add_argument("-k", type=int, default=7)
ap.add_argument("-d", type=int, default=3)
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--config", default=None)
args = ap.parse_args()
if args.config == "c3":
    args.n_pos = args.n_neg = 10000
elif args.config == "c5":
    args.n_pos = args.n_neg = 5000
    args.length_range, args.L, args.k, args.d = (150, 600), 12, 8, 4

tmp = tempfile.mkdtemp(prefix="gkm_e2e_")
pf, nf = os.path.join(tmp, "p.fa"), os.path.join(tmp, "n.fa")
synth.write_problem(pf, nf, args.n_pos, args.n_neg, args.length, tuple(args.length_range) if args.length_range else None)
n = args.n_pos + args.n_neg
rows_alloc = max(15000, n)
lib = device.load()
res = {}
for rep in range(2):   # first call pays HIP context creation
    t0 = time.time()
    kmat = np.zeros((rows_alloc, rows_alloc))
    rowp = (kmat.ctypes.data + np.arange(rows_alloc) * kmat.strides[0]).astype(np.uintp)
    sizes = np.ones(2, dtype=np.int32)
    t1 = time.time()
    opt = device.gkmOpt(args.t, args.L, args.k, args.d, 50, 50.0, 1.0, pf.encode(), nf.encode(), args.threads, int(os.environ.get("GKM_E2E_VERBOSITY", "0")))
    rc = lib.gkm_main_pywrapper(ctypes.byref(opt), rowp.ctypes.data, sizes.ctypes.data)
    t2 = time.time()
    assert rc == 0
    K = kmat[:n, :n]
    K = np.maximum(K, K.T)
    t3 = time.time()
    res = {"n": n, "pairs": n * (n - 1) // 2, "caller_alloc_s": t1 - t0, "boundary_call_s": t2 - t1,
           "caller_symmetrise_s": t3 - t2, "pairs_per_s_boundary": n * (n - 1) / 2 / (t2 - t1),
           "call": rep, "diag_ok": bool((np.diag(K) == 1).all()), "checksum": float(K[n - 1, : min(n, 64)].sum())}
    print(json.dumps(res), flush=True)
