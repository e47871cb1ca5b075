"""Randomised check of the drop-in boundary on the GPU box: random FASTA pairs with the quirks the
reference's reader accepts (lower case, N and other letters, CRLF, multi-line records, blank
lines) through `gkm_main_pywrapper` of this build and of the
COMPILED REFERENCE (oracle/_ref, test infrastructure), compared entry by entry.
python tools/fuzz_boundary.py [--seconds 120] [--seed 1]"""
import argparse
import ctypes
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def write_quirky(path, seqs, rng, prefix):
    with open(path, "wb") as f:
        for i, s in enumerate(seqs):
            eol = b"\r\n" if rng.random() < 0.2 else b"\n"
            f.write(b">" + prefix + str(i).encode() + (b" some description" if rng.random() < 0.3 else b"") + eol)
            if rng.random() < 0.4:      # multi-line record
                w = int(rng.integers(20, 90))
                for o in range(0, len(s), w):
                    f.write(s[o:o + w] + eol)
            else:
                f.write(s + eol)
            if rng.random() < 0.5:
                f.write(eol)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import torch  # noqa: F401
    from gkmqc_amd import device
    from oracle import oracle as O
    if not O.have_ref():
        raise SystemExit("oracle/_ref is missing (built where /root/reference exists)")
    rng = np.random.default_rng(a.seed)
    tmp = tempfile.mkdtemp()
    pf, nf = os.path.join(tmp, "p.fa"), os.path.join(tmp, "n.fa")
    alphabet = np.frombuffer(b"ACGTacgtNnRyKmsW", dtype=np.uint8)
    probs = np.array([20, 20, 20, 20, 3, 3, 3, 3, 1, 1, .3, .3, .3, .3, .3, .3])
    probs = probs / probs.sum()
    lib = device.load()
    t_end = time.time() + a.seconds
    cases = 0
    while time.time() < t_end:
        t = int(rng.integers(0, 6))
        L = int(rng.integers(4, 13))
        k = int(rng.integers(1, L + 1))
        d = int(rng.integers(0, min(4, L - k) + 1))
        if device.check_parameters(t, L, k, d):
            continue
        M, H, gamma = int(rng.integers(1, 256)), float(rng.integers(1, 200)), float(rng.choice([0.5, 1.0, 2.0]))
        n = int(rng.integers(2, 60))
        # (the compiled reference itself segfaults in this container once a sequence reaches ~1 400 nt --
        #  its DFS keeps 32 KB arrays per recursion level on the thread stack -- so the records stay
        #  short here; long and truncated records are compared with the oracle in fuzz_parity.py and
        #  in the golden "quirks" fixture)
        lens = rng.integers(L, 900, n)
        seqs = [alphabet[rng.choice(len(alphabet), int(ln), p=probs)].tobytes() for ln in lens]
        n_pos = int(rng.integers(1, n))
        write_quirky(pf, seqs[:n_pos], rng, b"p")
        write_quirky(nf, seqs[n_pos:], rng, b"n")
        opt = O.make_opt(t, L, k, d, M, H, gamma, pf, nf, nthreads=int(rng.integers(1, 9)), verbosity=0)
        rc_ref, k_ref, rp, rn = O.ref_pywrapper(opt, n + 3)
        popt = device.gkmOpt(t, L, k, d, M, H, gamma, pf.encode(), nf.encode(), int(rng.integers(1, 9)), 0)
        kmat = np.zeros((n + 3, n + 3))
        rows = (kmat.ctypes.data + np.arange(n + 3) * kmat.strides[0]).astype(np.uintp)
        sizes = np.ones(2, dtype=np.int32)
        rc = lib.gkm_main_pywrapper(ctypes.byref(popt), rows.ctypes.data, sizes.ctypes.data)
        tag = "t=%d L=%d k=%d d=%d M=%d H=%g n=%d seed=%d case=%d" % (t, L, k, d, M, H, n, a.seed, cases)
        if rc != 0 or rc_ref != 0 or (int(sizes[0]), int(sizes[1])) != (rp, rn):
            raise SystemExit("RETURN/SIZES differ: %s rc=%d ref=%d sizes=%s ref=%s" % (tag, rc, rc_ref, sizes, (rp, rn)))
        if not ((np.triu(kmat, 1) == 0).all() and (kmat[n:] == 0).all() and (kmat[:, n:] == 0).all()):
            raise SystemExit("wrote outside the contract: " + tag)
        il = np.tril_indices(n)
        kd, kr = kmat[il], k_ref[il]
        fin = np.isfinite(kr)
        err = (np.abs(kd[fin] - kr[fin]).max() / max(1e-300, np.abs(kr[fin]).max())) if fin.any() else 0.0
        if not (np.array_equal(np.isnan(kd), np.isnan(kr)) and err < (1e-9 if t in (3, 5) else 1e-12)):
            raise SystemExit("K differs from the reference: %s err=%g" % (tag, err))
        cases += 1
        if cases % 25 == 0:
            print("%d cases ok" % cases, flush=True)
    print("boundary fuzz ok: %d cases" % cases)


if __name__ == "__main__":
    main()
