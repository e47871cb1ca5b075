#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the COMPILED REFERENCE.

Run in the development container only (needs /root/reference):

    make -C oracle all ref
    python tests/golden/make_golden.py            # everything except the full C2 run
    python tests/golden/make_golden.py --c2-full  # + full 10 000 x 10 000 reference run (minutes)

The reference has no tests or known-answer vectors of its own, so
every expected value here is an OUTPUT OF THE REFERENCE ITSELF:
  * oracle/_ref/gkmkern_pylib_ref.so  -- unmodified gkm_main_pywrapper  -> K matrices
  * oracle/_ref/ref_probe.so          -- unmodified libgkm.c internals   -> c_m, positional
                                         weights, sqnorm, integer mismatch profiles
Fixtures are data only: FASTA inputs we wrote/generated ourselves + expected outputs.
"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from gkmqc_amd import synth  # noqa: E402

# (kernel_type, L, k, d, M, H, gamma)
QUIRK_PARAMS = [
    (4, 11, 7, 3, 50, 50.0, 1.0),
    (2, 10, 6, 3, 50, 50.0, 1.0),
    (4, 12, 8, 4, 50, 50.0, 1.0),
    (0, 8, 4, 4, 50, 50.0, 1.0),
    (3, 10, 6, 3, 50, 50.0, 2.0),
    (5, 11, 7, 3, 50, 50.0, 2.0),
    (1, 11, 7, 3, 50, 50.0, 1.0),
    (4, 10, 6, 3, 255, 20.0, 1.0),
    (2, 4, 2, 2, 50, 50.0, 1.0),
    (4, 9, 5, 4, 30, 7.0, 1.0),
    (0, 12, 6, 6, 50, 50.0, 1.0),
]

WEIGHT_TUPLES = [(t, L, k, d) for t in range(6) for (L, k, d) in
                 [(10, 6, 3), (11, 7, 3), (12, 8, 4), (8, 4, 4), (6, 3, 3), (12, 6, 6), (2, 1, 1), (9, 9, 0)]]


def rc(s):
    return s[::-1].translate(bytes.maketrans(b"ACGTacgt", b"TGCAtgca"))


def write_quirks():
    """A small FASTA pair exercising the reader's and encoder's corner cases."""
    rng = np.random.default_rng(20241003)

    def rnd(n):
        return synth._BASES[rng.integers(0, 4, n)].tobytes()

    pos, neg = [], []
    lens = [30, 45, 64, 100, 150, 151, 299, 300, 301, 320, 321, 333, 450, 600, 640, 700]
    for i, ln in enumerate(lens):
        pos.append((b"p%d" % i, [rnd(ln)]))
        neg.append((b"n%d some description here" % i, [rnd(lens[-1 - i])]))
    base = pos[7][1][0]
    pos.append((b"lower", [base[:120].lower()]))                      # case folding
    pos.append((b"dup_of_p7", [base]))                                # duplicate sequence
    pos.append((b"rc_of_p7", [rc(base)]))                             # reverse-complement pair
    pos.append((b"polyA", [b"A" * 100]))                              # dense hits
    pos.append((b"polyAC", [b"AC" * 60]))
    s = bytearray(rnd(200)); s[10] = ord("N"); s[50:54] = b"NNNN"; s[120] = ord("x"); s[199] = ord("-")
    pos.append((b"withN", [bytes(s)]))                                # non-ACGT -> A
    w = rnd(305)
    pos.append((b"wrapped", [w[i:i + 60] for i in range(0, len(w), 60)]))  # multi-line record
    long = rnd(2500)
    pos.append((b"toolong", [long[i:i + 500] for i in range(0, 2500, 500)]))  # truncated to 2047
    neg.append((b"crlf", [rnd(80) + b"\r", rnd(70) + b"\r"]))          # CRLF line ends
    neg.append((b"exact2047", [rnd(1000), rnd(1000), rnd(47)]))
    neg.append((b"short_as_L", [rnd(12)]))                             # exactly one 12-mer
    neg.append((b"mixedcase", [b"acgtACGTacgtNNNNacgtacgtacgtTTTTGGGGCCCCAAAAcgcgcgatatat"]))
    neg.append((b"tabs\tin header", [rnd(90)]))

    def dump(path, recs, crlf_names=(b"crlf",)):
        with open(path, "wb") as f:
            for name, lines in recs:
                f.write(b">" + name + b"\n")
                for ln in lines:
                    f.write(ln + b"\n")
                f.write(b"\n")

    pp, pn = os.path.join(HERE, "quirks_pos.fa"), os.path.join(HERE, "quirks_neg.fa")
    dump(pp, pos)
    dump(pn, neg)
    return pp, pn


def tril_pack(K):
    i, j = np.tril_indices(K.shape[0], -1)
    return K[i, j].copy()


def ref_K(opt, n):
    rc_, kmat, npos, nneg = O.ref_pywrapper(opt, n)
    assert rc_ == 0 and npos + nneg == n
    return kmat, npos


# name: n_pos, n_neg, length, length range, kernel type, L, k, d  (BASELINE.json configs[1], [2], [4])
FULL_CONFIGS = {"c2": (5000, 5000, 300, None, 4, 11, 7, 3),
                "c3": (10000, 10000, 300, None, 4, 11, 7, 3),
                "c5": (5000, 5000, None, (150, 600), 4, 12, 8, 4),
                # configs[3] stand-in: ONE `gkmqc.py evaluate` subset at its real size and parameters
                # (reference bin/gkmqc.py:150-154,181-185: 5 000 peaks + 5 000 nulls x 600 bp, wgkm L=10 k=6 d=3)
                # on the peak-like generator gkmqc_amd.synth.make_peak_sequences (no genome here)
                "c4": (5000, 5000, 600, None, 4, 10, 6, 3)}
PEAK_CONFIGS = ("c4",)


# (vii) gkmkernel_kernelfunc_batch (src/libgkm.c:1115-1153), the reference's scoring of one sequence against a set
# (SURVEY.md section 8 row f4): "query" rows scored against every sequence of the problem, support set first.
# name: (n_support, n_query, length, length_range, [(kernel_type, L, k, d, M, H, gamma), ...])
BATCH_CASES = {
    "ragged": (200, 40, None, (60, 700), [(2, 10, 6, 3, 50, 50.0, 1.0), (4, 11, 7, 3, 50, 50.0, 1.0),
                                          (5, 11, 7, 3, 50, 50.0, 2.0), (4, 12, 8, 4, 50, 50.0, 1.0),
                                          (0, 8, 4, 4, 50, 50.0, 1.0)]),
    "fixed300": (120, 24, 300, None, [(4, 11, 7, 3, 50, 50.0, 1.0), (3, 10, 6, 3, 50, 50.0, 2.0)]),
}


def write_batch_rows(tmp):
    out = {}
    for name, (nsup, nq, ln, lr, params) in BATCH_CASES.items():
        pf, nf = os.path.join(tmp, "batch_%s_p.fa" % name), os.path.join(tmp, "batch_%s_n.fa" % name)
        # the "positive" file is the support set, the "negative" file the queries: rows nsup .. nsup + nq - 1
        synth.write_problem(pf, nf, nsup, nq, ln or 300, lr)
        rows = np.arange(nsup, nsup + nq, dtype=np.int32)
        out[name + "_cfg"] = np.array([nsup, nq, ln or 0, lr[0] if lr else 0, lr[1] if lr else 0])
        for idx, (t, L, k, d, M, H, g) in enumerate(params):
            opt = O.make_opt(t, L, k, d, M, H, g, pf, nf, nthreads=1)
            out["%s_p%d_params" % (name, idx)] = np.array([t, L, k, d, M, H, g], dtype=np.float64)
            out["%s_p%d_K" % (name, idx)] = O.ref_batch_rows(opt, rows)
            print("batch rows", name, (t, L, k, d, M, H, g), out["%s_p%d_K" % (name, idx)].shape)
    np.savez_compressed(os.path.join(HERE, "batch_rows_expected.npz"), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c2-full", action="store_true")
    ap.add_argument("--full", action="append", choices=sorted(FULL_CONFIGS),
                    help="full-size digest of a BASELINE configuration through the reference (minutes to an hour)")
    ap.add_argument("--only-full", action="store_true", help="skip the small fixtures")
    ap.add_argument("--only-batch", action="store_true",
                    help="only batch_rows_expected.npz (the reference's batch-vs-set scoring entry)")
    ap.add_argument("--threads", type=int, default=os.cpu_count(), help="row threads of the reference for --full")
    args = ap.parse_args()
    if not O.have_ref():
        raise SystemExit("oracle/_ref is not built: run `make -C oracle ref` where /root/reference exists")

    tmp = os.path.join(ROOT, "gpurun_out", "golden_tmp")
    os.makedirs(tmp, exist_ok=True)
    if args.only_batch or not args.only_full:
        write_batch_rows(tmp)
    if args.only_batch:
        return
    if not args.only_full:
        # (i) mismatch weights c_m, bit patterns as hex
        wt = {}
        for (t, L, k, d) in WEIGHT_TUPLES:
            if O.lib().gkmo_check_params(t, L, k, d):
                continue
            wt["%d,%d,%d,%d" % (t, L, k, d)] = [float(x).hex() for x in O.ref_weights(t, L, k, d)]
        json.dump(wt, open(os.path.join(HERE, "mismatch_weights.json"), "w"), indent=0, sort_keys=True)

        # (ii)+(iii) quirks FASTA: lengths, positional weights, sqnorm, int profiles, K
        pp, pn = write_quirks()
        out = {}
        for idx, (t, L, k, d, M, H, g) in enumerate(QUIRK_PARAMS):
            opt = O.make_opt(t, L, k, d, M, H, g, pp, pn, nthreads=4)
            r = O.ref_profiles(opt)
            n = r["n"]
            K, npos = ref_K(opt, n)
            tag = "q%d" % idx
            out[tag + "_params"] = np.array([t, L, k, d, M, H, g], dtype=np.float64)
            out[tag + "_P"] = r["P"]
            out[tag + "_sqnorm"] = r["sqnorm"]
            out[tag + "_K"] = tril_pack(K)
            out[tag + "_wt"] = r["wt"][:, : int(r["lens"].max())]
            out["lens"] = r["lens"]
            out["n_pos"] = np.array(npos)
            print("quirks", (t, L, k, d, M, H, g), "N", n, "npos", npos)
        np.savez_compressed(os.path.join(HERE, "quirks_expected.npz"), **out)

        # (iv)/(v) synthetic configs (inputs are regenerated from gkmqc_amd.synth, not stored)
        cfgs = {
            # name: (n_pos, n_neg, length, length_range, t, L, k, d)
            "c1_full": (200, 200, 300, None, 2, 10, 6, 3),
            "c2_cut192": (192, 192, 300, None, 4, 11, 7, 3),
            "c5_cut64": (64, 64, None, (150, 600), 4, 12, 8, 4),
        }
        syn = {}
        for name, (npos, nneg, ln, lr, t, L, k, d) in cfgs.items():
            pf, nf = os.path.join(tmp, name + "_p.fa"), os.path.join(tmp, name + "_n.fa")
            synth.write_problem(pf, nf, npos, nneg, ln or 300, lr)
            opt = O.make_opt(t, L, k, d, 50, 50.0, 1.0, pf, nf, nthreads=8)
            K, _ = ref_K(opt, npos + nneg)
            syn[name + "_K"] = tril_pack(K)
            syn[name + "_cfg"] = np.array([npos, nneg, ln or 0, lr[0] if lr else 0, lr[1] if lr else 0, t, L, k, d])
            if name != "c1_full":
                r = O.ref_profiles(opt)
                syn[name + "_P"] = r["P"]
                syn[name + "_sqnorm"] = r["sqnorm"]
            print(name, "done")
        np.savez_compressed(os.path.join(HERE, "synthetic_expected.npz"), **syn)

    fulls = list(args.full or []) + (["c2"] if args.c2_full else [])
    for name in fulls:
        # (vi) a BASELINE configuration at FULL size through the reference: digest + sampled entries + row sums
        import time
        npos, nneg, ln, lr, t, L, k, d = FULL_CONFIGS[name]
        pf, nf = os.path.join(tmp, name + "_p.fa"), os.path.join(tmp, name + "_n.fa")
        if name in PEAK_CONFIGS:
            synth.write_peak_problem(pf, nf, npos, nneg, ln)
        else:
            synth.write_problem(pf, nf, npos, nneg, ln or 300, lr)
        opt = O.make_opt(t, L, k, d, 50, 50.0, 1.0, pf, nf, nthreads=args.threads)
        t0 = time.time()
        K, _ = ref_K(opt, npos + nneg)
        wall = time.time() - t0
        tri = tril_pack(K)
        rng = np.random.default_rng(7)
        sel = rng.choice(tri.size, 4000, replace=False)
        np.savez_compressed(os.path.join(HERE, name + "_full_digest.npz"),
                            sha256=np.frombuffer(hashlib.sha256(tri.tobytes()).digest(), dtype=np.uint8),
                            sample_idx=sel, sample_val=tri[sel],
                            row_sums=np.tril(K, -1).sum(axis=1), total=np.array(tri.sum()),
                            ref_wall_s=np.array(wall), ref_threads=np.array(args.threads),
                            cfg=np.array([npos, nneg, ln or 0, lr[0] if lr else 0, lr[1] if lr else 0, t, L, k, d]))
        print("%s full: reference wall %.1f s on %d threads" % (name, wall, args.threads), flush=True)
        del K, tri


if __name__ == "__main__":
    main()
