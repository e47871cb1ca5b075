#!/usr/bin/env python3
"""What a record of the bit-sliced kernel's hit list holds (CPU, numpy; no GPU): for random (row, column) pairs of a
synthetic workload, every window pair with at most d mismatches on both strands, grouped the way the kernel groups them
-- a record = one lane (320 row positions at W = 10), one cyclic shift, BS_GRP = 5 of the lane's 10 words, i.e. 32 runs
of 5 consecutive positions of one diagonal.  Prints hits per record and how many record visits a trip that resolved TWO
hits per record would save: a visit resolves one hit, a record with c hits is visited c times.

    python3 tools/hits_per_record.py peaks 60      # gkmQC's own shape: 600 bp, L=10 d=3
    python3 tools/hits_per_record.py c2 100        # config 2: 300 bp, L=11 d=3
"""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gkmqc_amd import synth
def codes(s):
    t = np.zeros(256, np.uint8); t[ord('C')] = 1; t[ord('G')] = 2; t[ord('T')] = 3
    return t[np.frombuffer(s, np.uint8)]
def stats(seqs, L, d, W, npairs, seed=1):
    rng = np.random.default_rng(seed)
    hist = np.zeros(64, np.int64); adj = 0; hits_total = 0; rec_adjpair = 0
    pair_visits = [0, 0]
    for _ in range(npairs):
        a = codes(seqs[rng.integers(len(seqs))]); b0 = codes(seqs[rng.integers(len(seqs))])
        for strand in range(2):
            b = b0 if strand == 0 else (3 - b0)[::-1]
            M = (a[:, None] != b[None, :]).astype(np.int16)
            na, nb = len(a) - L + 1, len(b) - L + 1
            S = np.zeros((na, nb), np.int16)
            for t in range(L): S += M[t:t + na, t:t + nb]
            p, q = np.nonzero(S <= d)
            piece = p // (32 * W); i0 = p % (32 * W)
            delta = (q - i0) % len(b)
            w = i0 % W; bit = i0 // W
            key = ((piece * 4096 + delta) * 2 + (w // 5))
            u, c = np.unique(key, return_counts=True)
            hist += np.bincount(np.minimum(c, 63), minlength=64)
            hits_total += len(p)
            # pairs: a visit takes the first hit (lowest word of the group, then lowest bit) AND, if set, the same bit of the
            # next word of the group = the next position of the same diagonal (shares the record, the piece, both packed
            # windows: two shifts instead of four loads)
            order = np.lexsort((bit, w % 5, key))
            ks, gs, bs = key[order], (w % 5)[order], bit[order]
            taken = np.zeros(len(ks), bool)
            pos = {}
            for n_, (k_, g_, b_) in enumerate(zip(ks.tolist(), gs.tolist(), bs.tolist())):
                pos[(k_, g_, b_)] = n_
            for n_, (k_, g_, b_) in enumerate(zip(ks.tolist(), gs.tolist(), bs.tolist())):
                if taken[n_]:
                    continue
                taken[n_] = True
                pair_visits[0] += 1
                m_ = pos.get((k_, g_ + 1, b_)) if g_ < 4 else None
                if m_ is not None and not taken[m_]:
                    taken[m_] = True
                    pair_visits[1] += 1
    recs = hist.sum()
    print("records with a hit %d, hits %d, hits/record %.2f" % (recs, hits_total, hits_total / recs))
    print("total=1: %.3f  2: %.3f  3: %.3f  4: %.3f  >=5: %.3f" % tuple(list(hist[1:5] / recs) + [hist[5:].sum() / recs]))
    # trips needed per record = total hits (each trip resolves one) -> record-trips = hits; with 2 per trip: sum ceil(c/2)
    print("visits that also take the next position of the diagonal: %d visits for %d hits (%.3f), %.3f of the visits carry a pair" % (
        pair_visits[0], hits_total, pair_visits[0] / hits_total, pair_visits[1] / pair_visits[0]))
    cs = np.arange(64)
    print("record-visits now %d; with two hits per visit %d (%.3f)" % ((hist * cs).sum(), (hist * ((cs + 1) // 2)).sum(), (hist * ((cs + 1) // 2)).sum() / (hist * cs).sum()))
if sys.argv[1] == "peaks":
    seqs = synth.make_peak_sequences(11, 300, 600, True) + synth.make_peak_sequences(12, 300, 600, False)
    stats(seqs, 10, 3, 10, int(sys.argv[2]))
elif sys.argv[1] == "c2":
    seqs = synth.make_sequences(1, 300, 300) + synth.make_sequences(2, 300, 300)
    stats(seqs, 11, 3, 10, int(sys.argv[2]))
