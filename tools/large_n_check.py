#!/usr/bin/env python3
"""Spot check beyond the fixtures' sizes (GPU box): a Gram matrix of --n sequences (default 30 000 x 300 bp: 7.2 GB of
matrix, 450 M work items) from the bit-sliced kernel, a handful of its rows recomputed by the general kernel
(gkmhip_gram_rows on those rows alone) -- raw values must be identical.  Exercises the 64-bit offsets of the work-item
table, the tile-transposed scratch and the matrix at a size no fixture reaches."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=30000)
    ap.add_argument("--length", type=int, default=300)
    a = ap.parse_args()
    import torch
    from gkmqc_amd import device, synth
    seqs = [device.encode(s) for s in synth.make_sequences(7, a.n, a.length, None)]
    n = len(seqs)
    stream = torch.cuda.current_stream().cuda_stream
    ctx = device.GramContext(4, 11, 7, 3)
    ctx.set_sequences(seqs, stream)
    G = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    t0 = time.time()
    ctx.gram_rows(np.arange(n), G.data_ptr(), n, None, 0, False, stream)
    torch.cuda.synchronize()
    print("n = %d: %s, %.1f ms (%.3g pairs/s), wall %.2f s" % (n, ctx.last_kernel_name(), ctx.last_kernel_ms(),
                                                               n * (n - 1) / 2 / (ctx.last_kernel_ms() * 1e-3), time.time() - t0))
    rows = np.array(sorted({0, 1, 63, 64, n // 3, n // 2, n - 65, n - 2, n - 1}), dtype=np.int32)
    ref = device.GramContext(4, 11, 7, 3)
    ref.set_kernel(device.KERNEL_DIRECT)
    ref.set_sequences(seqs, stream)
    R = torch.zeros((len(rows), n), dtype=torch.float64, device="cuda")
    ref.gram_rows(rows, R.data_ptr(), n, None, 0, True, stream)
    torch.cuda.synchronize()
    ok = True
    for i, r in enumerate(rows):
        same = bool(torch.equal(G[r, : r + 1], R[i, : r + 1]))
        ok = ok and same
        print("row %6d: %s (general kernel %s)" % (r, "identical" if same else "DIFFERS", ref.last_kernel_name()))
    above = float(G[0, 1:].abs().sum().item()) + float(G[n // 2, n // 2 + 1:].abs().sum().item())
    print("cells above the diagonal untouched: %s" % (above == 0.0))
    ctx.close()
    ref.close()
    sys.exit(0 if ok and above == 0.0 else 1)


if __name__ == "__main__":
    main()
