"""gkmqc_amd.gkmsvm.init (one subset after the other) against init_many (cross-validation of one subset on a
second stream beside the Gram kernel of the next) on synthetic subsets.
python tools/many_subsets.py [--subsets 6] [--length 300]
python tools/many_subsets.py --workload peaks --subsets 20 --repeats 10     # `gkmqc.py evaluate` as the pipeline runs it:
    up to 20 subsets x (5 000 peaks of 600 bp + 5 000 matched nulls), wgkm L=10 k=6 d=3, 5-fold x 10 repeats
    (reference bin/gkmqc.py:150-154,181-185,213-216,338-343; its README puts the run at "1 ~ 2 hrs with 10 threads")"""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--subsets", type=int, default=6)
    ap.add_argument("--n", type=int, default=5000)
    ap.add_argument("--length", type=int, default=300)
    ap.add_argument("--repeats", type=int, default=1)
    ap.add_argument("-L", type=int, default=11)
    ap.add_argument("-k", type=int, default=7)
    ap.add_argument("-d", type=int, default=3)
    ap.add_argument("--gpus", type=int, nargs="*", default=None, help="device ordinals for init_many(gpus=...)")
    ap.add_argument("--workload", choices=("iid", "peaks"), default="iid",
                    help="peaks: peak-like 600-bp subsets with gkmQC's default parameters (L=10 k=6 d=3)")
    ap.add_argument("--skip-sequential", action="store_true", help="only time init_many")
    a = ap.parse_args()
    if a.workload == "peaks":
        a.length, a.L, a.k, a.d = 600, 10, 6, 3
    from gkmqc_amd import gkmsvm, synth
    tmp = tempfile.mkdtemp()
    pairs = []
    for s in range(a.subsets):
        pf, nf = os.path.join(tmp, "p%d.fa" % s), os.path.join(tmp, "n%d.fa" % s)
        if a.workload == "peaks":
            synth.write_peak_problem(pf, nf, a.n, a.n, a.length, seed_pos=100 + 2 * s, seed_neg=101 + 2 * s)
        else:
            synth.write_problem(pf, nf, a.n, a.n, a.length, seed_pos=10 + 2 * s, seed_neg=11 + 2 * s)
        pairs.append((pf, nf))
    base = ["-p", "x", "-n", "y", "-s", "7", "-v", "0", "-t", "4", "-L", str(a.L), "-k", str(a.k), "-d", str(a.d),
            "-r", str(a.repeats)]
    a1 = gkmsvm.build_parser().parse_args(base + ["-w", os.path.join(tmp, "seq")])
    a2 = gkmsvm.build_parser().parse_args(base + ["-w", os.path.join(tmp, "ovl")])
    gkmsvm.init(*pairs[0], a1)                       # warm-up
    t0 = time.perf_counter()
    r1 = None if a.skip_sequential else [gkmsvm.init(p, n, a1) for p, n in pairs]
    t1 = time.perf_counter()
    r2 = gkmsvm.init_many(pairs, a2, gpus=a.gpus)
    t2 = time.perf_counter()
    print("whole evaluate run (%d subsets, FASTA on disk -> one AUC line per subset): %.2f s with init_many%s; AUCs %s"
          % (a.subsets, t2 - t1, "" if r1 is None else ", %.2f s one subset after the other" % (t1 - t0),
             " ".join("%.4f" % r[0] for r in r2)))
    if r1 is None:
        r1 = r2
    print("%d subsets of %d + %d x %d bp (L=%d k=%d d=%d), 5-fold x %d: init %.1f ms per subset, init_many %.1f ms "
          "per subset, same results: %s" % (a.subsets, a.n, a.n, a.length, a.L, a.k, a.d, a.repeats,
                                            (t1 - t0) / a.subsets * 1e3, (t2 - t1) / a.subsets * 1e3, r1 == r2))


if __name__ == "__main__":
    main()
