"""Shared helpers for the test-suite (test infrastructure; may use the oracle)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

QUIRK_POS = os.path.join(GOLDEN, "quirks_pos.fa")
QUIRK_NEG = os.path.join(GOLDEN, "quirks_neg.fa")


def golden_weights():
    raw = json.load(open(os.path.join(GOLDEN, "mismatch_weights.json")))
    return {tuple(int(x) for x in k.split(",")): np.array([float.fromhex(v) for v in vals])
            for k, vals in raw.items()}


def quirks_expected():
    z = np.load(os.path.join(GOLDEN, "quirks_expected.npz"))
    cases = []
    i = 0
    while "q%d_params" % i in z:
        t, L, k, d, M, H, g = z["q%d_params" % i]
        cases.append(dict(idx=i, kernel_type=int(t), L=int(L), k=int(k), d=int(d), M=int(M), H=float(H),
                          gamma=float(g), P=z["q%d_P" % i], sqnorm=z["q%d_sqnorm" % i], K=z["q%d_K" % i],
                          wt=z["q%d_wt" % i]))
        i += 1
    return cases, z["lens"], int(z["n_pos"])


def synthetic_expected():
    return np.load(os.path.join(GOLDEN, "synthetic_expected.npz"))


def tril_unpack(vec, n):
    K = np.zeros((n, n))
    i, j = np.tril_indices(n, -1)
    K[i, j] = vec
    return K


def tril_pack(K):
    i, j = np.tril_indices(K.shape[0], -1)
    return K[i, j]


def max_rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = np.maximum(np.abs(b), 1e-300)
    return float(np.max(np.abs(a - b) / denom)) if a.size else 0.0


def synth_codes(n_pos, n_neg, length=300, length_range=None):
    """The synthetic problem of bench.py / the golden generator as base-code arrays."""
    from gkmqc_amd import synth
    from gkmqc_amd.device import encode
    pos = synth.make_sequences(1, n_pos, length, length_range)
    neg = synth.make_sequences(2, n_neg, length, length_range)
    return [encode(s) for s in pos + neg]


def batch_rows_expected():
    """tests/golden/batch_rows_expected.npz: the reference's gkmkernel_kernelfunc_batch (src/libgkm.c:1115-1153) driven
    by tests/golden/make_golden.py --only-batch.  -> list of dict(name, n_support, n_query, length, length_range,
    kernel_type, L, k, d, M, H, gamma, K[n_query, n])."""
    z = np.load(os.path.join(GOLDEN, "batch_rows_expected.npz"))
    cases = []
    for key in z.files:
        if not key.endswith("_cfg"):
            continue
        name = key[:-4]
        nsup, nq, ln, lo, hi = (int(x) for x in z[key])
        i = 0
        while "%s_p%d_params" % (name, i) in z:
            t, L, k, d, M, H, g = z["%s_p%d_params" % (name, i)]
            cases.append(dict(name="%s_p%d" % (name, i), n_support=nsup, n_query=nq, length=ln or 300,
                              length_range=(lo, hi) if hi else None, kernel_type=int(t), L=int(L), k=int(k), d=int(d),
                              M=int(M), H=float(H), gamma=float(g), K=z["%s_p%d_K" % (name, i)]))
            i += 1
    return cases
