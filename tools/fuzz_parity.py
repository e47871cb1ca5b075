"""Randomised parity sweep on the GPU box: random (kernel type, L, k, d, M, H), random length
distributions (fixed, ragged, minimal, duplicates, poly-A), random row subsets -- integer
mismatch profiles and K of the device layer against the CPU oracle (test infrastructure).
python tools/fuzz_parity.py [--seconds 240] [--seed 1]"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


BITSLICED = [(L, d) for L in range(5, 13) for d in range(0, 5)]   # all have a bit-sliced kernel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import torch  # noqa: F401
    from gkmqc_amd import device, synth
    from oracle import oracle as O
    rng = np.random.default_rng(a.seed)
    tmp = tempfile.mkdtemp()
    pf, nf = os.path.join(tmp, "p.fa"), os.path.join(tmp, "n.fa")
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    t_end = time.time() + a.seconds
    cases = 0
    kernels = {}
    while time.time() < t_end:
        t = int(rng.integers(0, 6))
        if rng.random() < 0.75:   # (L, d) pairs with a bit-sliced instantiation
            L, d = BITSLICED[int(rng.integers(len(BITSLICED)))]
            k = int(rng.integers(1, L - d + 1))
        else:
            L = int(rng.integers(3, 13))
            k = int(rng.integers(1, L + 1))
            d = int(rng.integers(0, min(4, L - k) + 1))
        if device.check_parameters(t, L, k, d):
            continue
        M, H = int(rng.integers(1, 256)), float(rng.integers(1, 200))
        gamma = float(rng.choice([0.5, 1.0, 2.0]))
        n = int(rng.integers(2, 90))
        mode = int(rng.choice([0, 0, 5, 5, 5, 1, 2, 3, 4]))
        if mode == 0:
            lens = np.full(n, int(rng.integers(L, 700)))
        elif mode == 5:   # fixed lengths that fill one lane piece each: the one-piece-per-lane kernel variant
            lens = np.full(n, int(rng.choice([int(rng.integers(170, 321)), int(rng.integers(500, 641)), 300, 600])))
        elif mode == 1:
            lens = rng.integers(L, 700, n)
        elif mode == 2:
            lens = rng.integers(L, L + 12, n)
        elif mode == 3:
            lens = rng.choice([150, 300, 320, 321, 640, 2047], n)
        else:
            lens = rng.integers(L, 2048, n)
        seqs = [letters[rng.integers(0, 4, int(ln))].tobytes() for ln in lens]
        if n > 4:
            seqs[1] = seqs[0]                                  # duplicate
            seqs[2] = b"A" * len(seqs[2])                      # poly-A
            seqs[3] = bytes(reversed(seqs[0].translate(bytes.maketrans(b"ACGT", b"TGCA"))))  # reverse complement
        n_pos = max(1, n // 2)
        synth.write_fasta(pf, seqs[:n_pos], "p")
        synth.write_fasta(nf, seqs[n_pos:], "n")
        opt = O.make_opt(t, L, k, d, M, H, gamma, pf, nf)
        ref = O.gram(opt, want_profiles=True, nthreads=16)
        codes = [device.encode(s) for s in seqs]
        for kern in (device.KERNEL_AUTO, device.KERNEL_DIRECT):
            res = device.gram_matrix(codes, t, L, k, d, M, H, gamma, want_profiles=True, kernel=kern)
            K = res["K"].cpu().numpy()
            P = res["P"].cpu().numpy()
            il = np.tril_indices(n)
            if not (P[il] == ref["P"][il]).all():
                raise SystemExit("PROFILE MISMATCH t=%d L=%d k=%d d=%d n=%d mode=%d kernel=%s seed=%d case=%d" %
                                 (t, L, k, d, n, mode, res["kernel"], a.seed, cases))
            # (a poly-A row with large weights wraps its int32 self profile -- like the reference -- and
            #  its square root is NaN on both sides: NaNs must coincide, the rest must agree)
            kd, kr = K[il], ref["K"][il]
            fin = np.isfinite(kr)
            same_nan = np.array_equal(np.isnan(kd), np.isnan(kr))
            err = (np.abs(kd[fin] - kr[fin]).max() / max(1e-300, np.abs(kr[fin]).max())) if fin.any() else 0.0
            tol = 1e-9 if t in (3, 5) else 1e-12
            if not (same_nan and err < tol):
                raise SystemExit("K MISMATCH %g t=%d L=%d k=%d d=%d n=%d mode=%d kernel=%s" % (err, t, L, k, d, n, mode, res["kernel"]))
            kernels[res["kernel"]] = kernels.get(res["kernel"], 0) + 1
        cases += 1
        if cases % 50 == 0:
            print("%d cases ok %s" % (cases, kernels), flush=True)
    print("fuzz ok: %d cases, kernels used: %s" % (cases, kernels))


if __name__ == "__main__":
    main()
