#!/usr/bin/env python3
"""Capture golden input/output of the reference's OWN Python caller (scripts/gkmsvm.py) for the
host-side mirror gkmqc_amd/gkmsvm.py.  Development container only.

The reference module is imported from a scratch tree under /tmp (copies of bin/ scripts/ data/ with
the compiled reference .so in bin/); nothing of it is copied into this
repository.  Committed here: the two FASTA inputs we generate and the numbers it returned."""
import hashlib
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

TREE = "/tmp/gkmqc_tree"


def make_inputs():
    rng = np.random.default_rng(424242)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    motifs = [b"TGACTCAGCA", b"GGGCGGGGC"]

    def seq(n, plant):
        s = bytearray(bases[rng.integers(0, 4, n)].tobytes())
        if plant:
            m = motifs[int(rng.integers(0, len(motifs)))]
            for _ in range(int(rng.integers(1, 3))):
                p = int(rng.integers(20, n - 30))
                s[p:p + len(m)] = m
        return bytes(s)

    pos = [seq(200, rng.random() < 0.8) for _ in range(150)]
    neg = [seq(200, False) for _ in range(160)]
    pf, nf = os.path.join(HERE, "motif_pos.fa"), os.path.join(HERE, "motif_neg.fa")
    for path, seqs, tag in ((pf, pos, "p"), (nf, neg, "n")):
        with open(path, "wb") as f:
            for i, s in enumerate(seqs):
                f.write(b">chr1:%d-%d_%s%d\n" % (1000 * i, 1000 * i + 200, tag.encode(), i) + s + b"\n\n")
    return pf, nf


def c4_fixture(ref):
    """BASELINE configs[3] stand-in at its real size: ONE `gkmqc.py evaluate` subset (5 000 peak-like
    positives + 5 000 matched nulls x 600 bp, wgkm L=10 k=6 d=3, 5-fold CV; reference
    bin/gkmqc.py:150-154,181-185,213-215) through the reference's OWN module: kernel matrix by the
    compiled reference, cross-validation by its scikit-learn harness."""
    import time
    from gkmqc_amd import synth
    tmp = os.path.join(ROOT, "gpurun_out", "golden_tmp")
    os.makedirs(tmp, exist_ok=True)
    pf, nf = os.path.join(tmp, "c4_p.fa"), os.path.join(tmp, "c4_n.fa")
    synth.write_peak_problem(pf, nf, 5000, 5000, 600)
    threads = int(os.environ.get("GKM_GOLDEN_THREADS", "6"))
    t0 = time.time()
    kmat, n_pos, n_neg = ref.computeGkmKernel([4, 10, 6, 3, 50, 50, 1.0, pf, nf, threads, 0])
    t1 = time.time()
    svm = [1.0, 0.001, 0, 512, 5, 1, 0, 1, 5]
    auc, std = ref.crossValidate(list(svm), kmat, n_pos, n_neg)
    t2 = time.time()
    rng = np.random.default_rng(5)
    ii, jj = rng.integers(0, kmat.shape[0], 300), rng.integers(0, kmat.shape[0], 300)
    out = {"c4_peaks": {
        "args_gkm": [4, 10, 6, 3, 50, 50, 1.0, "<synth.write_peak_problem 5000+5000 x 600>", "", threads, 0],
        "args_svm": svm, "n_pos": int(n_pos), "n_neg": int(n_neg), "auc_mean": float(auc), "auc_std": float(std),
        "kmat_sha256": hashlib.sha256(np.ascontiguousarray(kmat).tobytes()).hexdigest(),
        "kmat_min": float(kmat.min()), "kmat_sum": float(kmat.sum()),
        "sample_i": ii.tolist(), "sample_j": jj.tolist(), "sample_v": [float(kmat[a, b]) for a, b in zip(ii, jj)],
        "ref_kernel_wall_s": t1 - t0, "ref_cv_wall_s": t2 - t1, "ref_threads": threads}}
    # the same subset as `gkmqc.py evaluate` really runs it: 5-fold x 10 repeats (bin/gkmqc.py:213-216: `-x 5 -r 10`)
    svm10 = [1.0, 0.001, 0, 512, 5, 10, 0, 1, 5]
    auc10, std10 = ref.crossValidate(list(svm10), kmat, n_pos, n_neg)
    t3 = time.time()
    out["c4_peaks_r10"] = {"args_svm": svm10, "auc_mean": float(auc10), "auc_std": float(std10),
                           "kmat_sha256": out["c4_peaks"]["kmat_sha256"], "ref_cv_wall_s": t3 - t2}
    json.dump(out, open(os.path.join(HERE, "gkmsvm_expected_c4.json"), "w"), indent=1)
    print("c4_peaks", n_pos, n_neg, auc, std, "kernel %.0f s, cv %.0f s" % (t1 - t0, t2 - t1))
    print("c4_peaks 5-fold x 10 repeats", auc10, std10, "cv %.0f s" % (t3 - t2))


def main():
    assert O.have_ref(), "run `make -C oracle ref` first"
    if os.path.isdir(TREE):
        shutil.rmtree(TREE)
    os.makedirs(TREE)
    for d in ("bin", "scripts", "data"):
        shutil.copytree(os.path.join("/root/reference", d), os.path.join(TREE, d))
    shutil.copy(os.path.join(ROOT, "oracle", "_ref", "gkmkern_pylib_ref.so"), os.path.join(TREE, "bin", "gkmkern_pylib.so"))
    sys.path.insert(0, os.path.join(TREE, "scripts"))
    import gkmsvm as ref  # the reference's module

    if "--c4" in sys.argv:
        return c4_fixture(ref)
    pf, nf = make_inputs()
    out = {}
    for name, gkm, svm in (
        ("wgkm_L10", [4, 10, 6, 3, 50, 50, 1.0, pf, nf, 2, 0], [1.0, 0.001, 0, 512, 5, 2, 0, 7, 2]),
        ("gkmrbf_L11", [3, 11, 7, 3, 50, 50, 2.0, pf, nf, 2, 0], [10.0, 0.001, 1, 512, 4, 1, 0, 11, 1]),
        ("full_L8", [1, 8, 5, 3, 50, 50, 1.0, pf, nf, 1, 0], [1.0, 0.001, 0, 512, 3, 1, 0, 3, 1]),
    ):
        kmat, n_pos, n_neg = ref.computeGkmKernel(list(gkm))
        auc, std = ref.crossValidate(list(svm), kmat, n_pos, n_neg)
        rng = np.random.default_rng(5)
        ii, jj = rng.integers(0, kmat.shape[0], 300), rng.integers(0, kmat.shape[0], 300)
        out[name] = {
            "args_gkm": gkm[:7] + ["motif_pos.fa", "motif_neg.fa"] + gkm[9:], "args_svm": svm,
            "n_pos": int(n_pos), "n_neg": int(n_neg), "auc_mean": float(auc), "auc_std": float(std),
            "kmat_sha256": hashlib.sha256(np.ascontiguousarray(kmat).tobytes()).hexdigest(),
            "kmat_min": float(kmat.min()), "kmat_sum": float(kmat.sum()),
            "sample_i": ii.tolist(), "sample_j": jj.tolist(), "sample_v": [float(kmat[a, b]) for a, b in zip(ii, jj)],
        }
        print(name, n_pos, n_neg, auc, std, kmat.min())
    json.dump(out, open(os.path.join(HERE, "gkmsvm_expected.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
