"""Where the time of the GPU-resident pipeline (gkmqc_amd.gkmsvm: FASTA -> matrix -> CV) goes.
python tools/pipeline_profile.py [--n-pos 5000 --n-neg 5000]
python tools/pipeline_profile.py --workload peaks      # one `gkmqc.py evaluate` subset: 600 bp, L=10 k=6 d=3, 5-fold x 10"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def peaks(a):
    """One evaluate subset, piece by piece, on the context init_many keeps (reference bin/gkmqc.py:213-216: -x 5 -r 10)."""
    import torch
    from gkmqc_amd import device, svmcv, synth
    tmp = tempfile.mkdtemp()
    pos, neg = os.path.join(tmp, "p.fa"), os.path.join(tmp, "n.fa")
    synth.write_peak_problem(pos, neg, a.n_pos, a.n_neg, 600)
    torch.zeros(1, device="cuda")
    for rep in range(3):
        t = [time.perf_counter()]
        seqs, n_pos, _, _ = device.read_problem(pos, neg)
        t.append(time.perf_counter())
        ctx = device.cached_context(4, 10, 6, 3, 50, 50.0, 1.0, 0, 0)
        stream = torch.cuda.current_stream().cuda_stream
        ctx.set_sequences(seqs, stream)
        torch.cuda.synchronize()
        t.append(time.perf_counter())
        n = len(seqs)
        G = torch.zeros((n, n), dtype=torch.float64, device="cuda")
        sq = torch.zeros(n, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        t.append(time.perf_counter())
        ctx.gram_rows(np.arange(n), G.data_ptr(), n, None, n, False, stream)
        t.append(time.perf_counter())
        torch.cuda.synchronize()
        t.append(time.perf_counter())
        ctx.normalize(G.data_ptr(), n, sq.data_ptr(), True, stream)
        torch.cuda.synchronize()
        t.append(time.perf_counter())
        svmcv.crossValidate([1.0, 0.001, 0, 512, 5, 10, 0, 1, 1], G, n_pos, n - n_pos)
        t.append(time.perf_counter())
        names = ["read_problem", "set_sequences", "alloc", "gram enqueue", "gram wait", "normalize (symmetric)", "cv 5 x 10"]
        print("rep %d: " % rep + "  ".join("%s %.1f ms" % (nm, (t[i + 1] - t[i]) * 1e3) for i, nm in enumerate(names))
              + "  total %.1f ms (hot kernel %.1f ms)" % ((t[-1] - t[0]) * 1e3, ctx.last_kernel_ms()))
        del G
    device.release_cached_contexts()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-pos", type=int, default=5000)
    ap.add_argument("--n-neg", type=int, default=5000)
    ap.add_argument("--length", type=int, default=300)
    ap.add_argument("--workload", choices=("iid", "peaks"), default="iid")
    a = ap.parse_args()
    if a.workload == "peaks":
        return peaks(a)
    import torch
    from gkmqc_amd import device, gkmsvm, svmcv, synth
    tmp = tempfile.mkdtemp()
    pos, neg = os.path.join(tmp, "p.fa"), os.path.join(tmp, "n.fa")
    synth.write_problem(pos, neg, a.n_pos, a.n_neg, a.length)
    torch.zeros(1, device="cuda")
    for rep in range(3):
        t = [time.perf_counter()]
        seqs, n_pos, _, _ = device.read_problem(pos, neg)
        t.append(time.perf_counter())
        ctx = device.GramContext(4, 11, 7, 3, 50, 50.0, 1.0, 0)
        stream = torch.cuda.current_stream().cuda_stream
        ctx.set_sequences(seqs, stream)
        torch.cuda.synchronize()
        t.append(time.perf_counter())
        n = len(seqs)
        G = torch.zeros((n, n), dtype=torch.float64, device="cuda")
        sq = torch.zeros(n, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        t.append(time.perf_counter())
        ctx.gram_rows(np.arange(n), G.data_ptr(), n, None, n, False, stream)
        torch.cuda.synchronize()
        t.append(time.perf_counter())
        ctx.normalize(G.data_ptr(), n, sq.data_ptr(), False, stream)
        torch.cuda.synchronize()
        t.append(time.perf_counter())
        K = torch.maximum(G, G.T)
        torch.cuda.synchronize()
        t.append(time.perf_counter())
        ctx.close()
        t.append(time.perf_counter())
        auc = svmcv.crossValidate([1.0, 0.001, 0, 512, 5, 1, 0, 7, 1], K, n_pos, n - n_pos)
        t.append(time.perf_counter())
        names = ["read_problem", "ctx+set_sequences", "alloc", "gram_rows", "normalize", "maximum(K,K.T)", "close", "cv"]
        print("rep %d: " % rep + "  ".join("%s %.1f ms" % (nm, (t[i + 1] - t[i]) * 1e3) for i, nm in enumerate(names))
              + "  total %.1f ms" % ((t[-1] - t[0]) * 1e3))
        # the cross-validation in pieces
        from sklearn.metrics import roc_auc_score
        from sklearn.model_selection import StratifiedKFold
        y = np.concatenate((np.repeat(1, n_pos), np.repeat(0, n - n_pos)))
        c0 = time.perf_counter()
        folds = list(StratifiedKFold(n_splits=5, shuffle=True, random_state=7).split(np.zeros(n), y))
        c1 = time.perf_counter()
        sol, h = svmcv.train_folds(K, [tr for tr, _ in folds], y, 1.0, 1e-3)
        c2 = time.perf_counter()
        sc = svmcv.decision_values(K, h, [te for _, te in folds])
        c3 = time.perf_counter()
        aucs = [roc_auc_score(y[te], s_) for (_, te), s_ in zip(folds, sc)]
        c4 = time.perf_counter()
        print("        cv pieces: folds %.1f ms  train %.1f ms (%d..%d iterations)  decision %.1f ms  auc %.1f ms" %
              ((c1 - c0) * 1e3, (c2 - c1) * 1e3, int(sol.iters.min()), int(sol.iters.max()), (c3 - c2) * 1e3, (c4 - c3) * 1e3))
        t0 = time.perf_counter()
        gkmsvm.main(["-p", pos, "-n", neg, "-w", os.path.join(tmp, "out"), "-s", "7", "-v", "0", "-t", "4", "-L", "11",
                     "-k", "7", "-d", "3"])
        print("        gkmsvm.main end to end %.1f ms" % ((time.perf_counter() - t0) * 1e3))


if __name__ == "__main__":
    main()
