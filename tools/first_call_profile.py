#!/usr/bin/env python3
"""Where the FIRST call of a process goes (GPU box): a fresh `python -m gkmqc_amd.gkmsvm` pays imports, HIP start-up, code
object loading and first-touch allocations once -- what a one-subset-per-process run (`bin/gkmqc.py -P`, one SLURM job per
subset) pays every time.  Stage by stage, first call and second call.     python3 tools/first_call_profile.py [--workload c2|peaks]
"""
import argparse
import os
import sys
import tempfile
import time

T0 = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    a = ap.parse_args()
    marks = [("interpreter + argparse", time.perf_counter())]

    def mark(name):
        marks.append((name, time.perf_counter()))

    import numpy as np  # noqa: F401
    mark("import numpy")
    import torch
    mark("import torch")
    from gkmqc_amd import device, gkmsvm, svmcv  # noqa: F401
    mark("import gkmqc_amd (device, gkmsvm, svmcv)")
    import bench
    w = bench.parse_args(["--workload", a.workload])
    tmp = tempfile.mkdtemp(prefix="gkm_first_")
    pf, nf = bench.write_problem_files(w, w.n_pos, w.n_neg, tmp)
    mark("write the FASTA files (not part of a real run)")
    device.load()
    mark("load gkmkern_pylib.so")
    torch.zeros(1, device="cuda")
    torch.cuda.synchronize()
    mark("HIP start-up (first torch allocation)")
    args_gkm = [w.kernel_type, w.L, w.k, w.d, 50, 50.0, 1.0, pf, nf, 16, 0]
    args_svm = [1.0, 0.001, 0, 512, 5, 1, 0, 1, 1]
    for rep in ("first", "second"):
        K, n_pos, n_neg = gkmsvm.computeGkmKernel(args_gkm, resident=True)
        torch.cuda.synchronize()
        mark("%s computeGkmKernel (FASTA -> matrix in HBM)" % rep)
        auc, std = gkmsvm.crossValidate(args_svm, K, n_pos, n_neg)
        mark("%s crossValidate (5 folds on the GPU)" % rep)
        del K
    prev = T0
    for name, t in marks:
        print("%-60s %8.1f ms   (at %7.1f ms)" % (name, (t - prev) * 1e3, (t - T0) * 1e3))
        prev = t


if __name__ == "__main__":
    main()
