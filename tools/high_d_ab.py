#!/usr/bin/env python3
"""Bit-sliced kernel against the general one (k_gram_direct), kernel ms and bit-identity of the matrices, per (L, d) (GPU box):
how round 4 found the break-even that `auto` now goes by (gkm_gram.hip auto_takes_bitslice; profiles/r4_high_d_bitslice_vs_direct.txt).

    python3 tools/high_d_ab.py [--n 2000] [--length 300] [--pairs "10,5 8,4 7,3"]     # pairs of the product's table
    python3 tools/high_d_ab.py --all                                                   # every pair with 5 <= d < L <= 12

Pairs the product does not instantiate need a build whose table holds them:

    tools/build_variant.sh bsx "" HEAD       # then extend pick_bitslice in build_variants/src_bsx/gkmqc_amd/csrc/gkm_gram_bitslice.hip
                                             # (GKM_BS(L, d) ...) and run that directory's make again
    GKM_LIB_PATH=build_variants/lib_bsx.so python3 tools/high_d_ab.py --all --n 8000

Measure at a realistic size: on 2 000 sequences the general kernel's grid once did not fill the GPU and the comparison misled.
iid ACGT, kernel type 4, whole lower triangle, second launch of each kernel.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--length", type=int, default=300)
    ap.add_argument("--all", action="store_true", help="every (L, d) with 6 <= L <= 12, 5 <= d < L (k = L - d)")
    ap.add_argument("--pairs", default="", help='e.g. "10,6 11,7": these (L, d) only')
    ap.add_argument("--peaks", action="store_true", help="peak-like sequences (synth.make_peak_sequences: GC drift, repeats, shared elements) instead of iid ACGT")
    a = ap.parse_args()
    import torch
    from gkmqc_amd import device, synth
    raw = (synth.make_peak_sequences(11, a.n // 2, a.length, True) + synth.make_peak_sequences(12, a.n - a.n // 2, a.length, False)
           if a.peaks else synth.make_sequences(1, a.n, a.length))
    seqs = [np.frombuffer(s, np.uint8) for s in raw]
    table = np.zeros(256, np.uint8)
    for i, ch in enumerate(b"ACGT"):
        table[ch] = i
    seqs = [table[s] for s in seqs]
    print("%d x %d bp, %s, type 4; kernel ms (second launch)" % (a.n, a.length, "peak-like" if a.peaks else "iid ACGT"))
    pairs = [(L, d) for L in range(6, 13) for d in range(5, L)] if a.all else [(11, 5), (12, 5), (12, 6), (10, 5), (11, 6), (12, 7)]
    if a.pairs:
        pairs = [tuple(int(x) for x in q.split(",")) for q in a.pairs.split()]
    for L, k, d in [(L, L - d, d) for L, d in pairs]:
        res = {}
        for name, which in (("bitslice", device.KERNEL_BITSLICE), ("direct", device.KERNEL_DIRECT)):
            try:
                for _ in range(2):
                    r = device.gram_matrix(seqs, 4, L, k, d, kernel=which)
                res[name] = (r["ms"], r["kernel"], r["K"].cpu().numpy())
            except device.GkmError as e:
                res[name] = (None, str(e)[:60], None)
        b, g = res["bitslice"], res["direct"]
        same = b[2] is not None and g[2] is not None and np.array_equal(np.tril(b[2]), np.tril(g[2]))
        print("L=%2d k=%d d=%d   bit-sliced %s   general %s   %s" % (
            L, k, d, "%8.1f ms (%s)" % (b[0], b[1]) if b[0] else "not built (%s)" % b[1],
            "%8.1f ms" % g[0] if g[0] else "failed", "bit-identical" if same else "NOT COMPARED" if b[2] is None else "DIFFERENT"), flush=True)


if __name__ == "__main__":
    main()
