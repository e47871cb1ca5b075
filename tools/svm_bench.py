"""Time the GPU-resident cross-validation (gkmqc_amd/svmcv.py) against the reference's sklearn
path on the same matrix.  python tools/svm_bench.py [--n-pos 5000 --n-neg 5000 --length 300] [--sklearn]"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-pos", type=int, default=5000)
    ap.add_argument("--n-neg", type=int, default=5000)
    ap.add_argument("--length", type=int, default=300)
    ap.add_argument("-L", type=int, default=11)
    ap.add_argument("-k", type=int, default=7)
    ap.add_argument("-d", type=int, default=3)
    ap.add_argument("--ncv", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=1)
    ap.add_argument("--motif", type=float, default=0.5, help="fraction of positives carrying a planted motif")
    ap.add_argument("--sklearn", action="store_true", help="also time the reference's sklearn CV (host)")
    ap.add_argument("--procs", type=int, default=5)
    ap.add_argument("--shrinking", type=int, default=0, help="LIBSVM's shrinking heuristic (the general GPU solver)")
    ap.add_argument("--general", action="store_true", help="force the general GPU solver (k_smo_general) without shrinking")
    a = ap.parse_args()
    import torch
    from gkmqc_amd import gkmsvm, svmcv, synth
    tmp = tempfile.mkdtemp()
    pos, neg = os.path.join(tmp, "p.fa"), os.path.join(tmp, "n.fa")
    ps = synth.make_sequences(1, a.n_pos, a.length)
    ns = synth.make_sequences(2, a.n_neg, a.length)
    rng = np.random.default_rng(5)
    motifs = [b"TGACTCAGCA", b"GGGCGGGGCC", b"CACGTGACCA"]
    for i in range(a.n_pos):                     # a weak, realistic signal: one of three motifs in some positives
        if rng.random() < a.motif:
            m = motifs[int(rng.integers(3))]
            at = int(rng.integers(0, a.length - len(m)))
            ps[i] = ps[i][:at] + m + ps[i][at + len(m):]
    synth.write_fasta(pos, ps, "p")
    synth.write_fasta(neg, ns, "n")
    args_gkm = [4, a.L, a.k, a.d, 50, 50, 1.0, pos, neg, 16, 0]
    args_svm = [1.0, 0.001, a.shrinking, 512, a.ncv, a.repeats, 0, 7, a.procs]
    if a.general:
        svmcv.FAST_FOLD_SAMPLES = 0
    t0 = time.time()
    K, n_pos, n_neg = gkmsvm.computeGkmKernel(args_gkm, resident=True)
    torch.cuda.synchronize()
    t1 = time.time()
    auc = svmcv.crossValidate(args_svm, K, n_pos, n_neg)
    torch.cuda.synchronize()
    t2 = time.time()
    auc2 = svmcv.crossValidate(args_svm, K, n_pos, n_neg)
    t3 = time.time()
    print("matrix %.3f s   gpu cv %.3f s (second call %.3f s)   auc %.6f +- %.6f" % (t1 - t0, t2 - t1, t3 - t2, *auc))
    if a.sklearn:
        Kh = K.cpu().numpy()
        t4 = time.time()
        auc3 = gkmsvm.crossValidate(args_svm, Kh, n_pos, n_neg)
        t5 = time.time()
        print("sklearn cv %.3f s on %d processes   auc %.6f +- %.6f   identical: %s" % (t5 - t4, a.procs, *auc3, auc3 == auc))


if __name__ == "__main__":
    import logging
    logging.basicConfig(stream=sys.stdout, level=logging.INFO, format="%(message)s")
    main()
