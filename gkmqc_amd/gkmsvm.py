"""Host-side mirror of the reference's kernel + SVM harness (scripts/gkmsvm.py) for the
MI355X build -- SURVEY.md §8(f1).

Same names, argument lists and results as the reference so the pipeline can call either:

    computeGkmKernel(args_gkm)            -> (kmat, n_pseqs, n_nseqs)     scripts/gkmsvm.py:67-99
    crossValidate(args_svm, kmat, np, nn) -> (auc_mean, auc_std)          scripts/gkmsvm.py:127-176
    init(pos_fa, neg_fa, args)            appends one line to <name>.gkmqc.eval.out   :181-220
    main()                                same command line               :224-303

Differences (all above the C ABI, none changes a number):
  * the matrix is produced on the GPU through the device layer (include/gkm_hip.h) and sized to
    the problem: no 15 000 x 15 000 host buffer to zero-fill, no 15 000-sequence cap;
  * `np.maximum(kmat, kmat.T)` of the reference (lower triangle against a zero upper triangle,
    i.e. negative entries clamp to 0) is reproduced exactly, on the device;
  * `--fast-estimation 1` raises NotImplementedError (the reference hits a NameError there,
    SURVEY.md App. B #11).
`computeGkmKernel(..., backend="boundary")` instead drives `gkm_main_pywrapper` exactly like the
reference does (row pointers into a zeroed matrix) -- useful to validate a drop-in install.
`computeGkmKernel(..., resident=True)` leaves the symmetrised matrix in HBM (torch CUDA tensor);
`crossValidate` then runs every fold on the GPU (gkmqc_amd/svmcv.py) with
results bit-identical to scikit-learn's solver.  `--svm-solver sklearn` keeps the reference's.
"""
import argparse
import ctypes
import logging
import os
import sys
from multiprocessing import Pool

import numpy as np

from . import device

_KMAT = None  # shared with forked CV workers, like the reference's module global


def computeGkmKernel(args_gkm, backend="device", gpu=0, resident=False, gpus=1, keep_context=False, context_slot=0,
                     wait=True):
    """args_gkm = [kernel_type, L, k, d, M, H, gamma, pos_fa, neg_fa, n_processes, verbosity].

    wait=False (resident, one GPU, keep_context): the matrix is returned as soon as its kernels are enqueued on torch's
    current stream; whatever the caller enqueues there next (the GPU cross-validation) is ordered behind them.

    gpus > 1 (or a list of device ordinals): the matrix is computed on that many GPUs of the node by
    this one process (row-block sharding + RCCL all-gather behind the C ABI, include/gkm_hip.h
    gkmhip_gram_allgather) and the copy on device `gpus[0]` is returned -- bit-identical to gpus=1."""
    kernel_type, L, k, d, M, H, gamma, pos_fa, neg_fa, nproc, verbosity = args_gkm
    if backend == "boundary":
        seqs, n_pos, _, _ = device.read_problem(pos_fa, neg_fa)
        n = len(seqs)
        kmat = np.zeros((n, n))
        rows = (kmat.ctypes.data + np.arange(n) * kmat.strides[0]).astype(np.uintp)
        narr = np.ones(2, dtype=np.int32)
        opts = device.gkmOpt(kernel_type, L, k, d, int(M), float(H), float(gamma), os.fsencode(pos_fa),
                             os.fsencode(neg_fa), int(nproc), int(verbosity))
        ret = device.load().gkm_main_pywrapper(ctypes.byref(opts), rows.ctypes.data, narr.ctypes.data)
        if ret:
            logging.error("error on kernel construction")
            sys.exit()
        n_pseqs, n_nseqs = int(narr[0]), int(narr[1])
        return np.maximum(kmat, kmat.T), n_pseqs, n_nseqs

    import torch
    seqs, n_pos, n_invalid, n_trunc = device.read_problem(pos_fa, neg_fa)
    if n_pos == 0 or n_pos == len(seqs):
        logging.error("error on kernel construction")
        sys.exit()
    devices = list(gpus) if isinstance(gpus, (list, tuple)) else list(range(gpu, gpu + int(gpus)))
    if len(devices) > 1:
        res = device.gram_matrix_multi(seqs, kernel_type, L, k, d, int(M), float(H), float(gamma), devices=devices)
        K = res["K"][0]
        del res
    else:
        K = device.gram_matrix(seqs, kernel_type, L, k, d, int(M), float(H), float(gamma), device=devices[0],
                               keep_context=keep_context, context_slot=context_slot,
                               wait=wait or not (resident and keep_context))["K"]
    K = torch.maximum(K, K.T)                  # scripts/gkmsvm.py:97 (lower triangle + unit diagonal, zeros above)
    if resident:
        return K, n_pos, len(seqs) - n_pos
    return K.cpu().numpy(), n_pos, len(seqs) - n_pos


def _svm_fold(job):
    """One fold on the host with scikit-learn's C-SVC on the precomputed kernel -- the solver, and the arguments, the
    reference hands each fold to (scripts/gkmsvm.py:104-122); used for `--svm-solver sklearn` and as the fallback of the
    GPU solver.  -> the fold's AUC."""
    from sklearn.metrics import roc_auc_score
    from sklearn.svm import SVC
    (C, tol, shrinking, cache_mb), y, train, test = job[0][:4], job[1], job[2], job[3]
    model = SVC(kernel="precomputed", C=C, tol=tol, shrinking=bool(shrinking), gamma=1.0, cache_size=cache_mb)
    model.fit(_KMAT[np.ix_(train, train)], y[train])
    auc = roc_auc_score(y[test], model.decision_function(_KMAT[np.ix_(test, train)]))
    logging.info("fold of %d + %d samples on the host: nu %.3f, AUC %.3f", len(train), len(test),
                 np.abs(model.dual_coef_[0]).sum() / len(train), auc)
    return auc


def crossValidate(args_svm, _kmat, n_pseqs, n_nseqs, about_to_launch=None, plan=None):
    """args_svm = [C, tol, shrinking, cache_size, ncv, repeats, fast_estimation, random_seeds, p].
    about_to_launch, plan: see svmcv.crossValidate (only the GPU-resident path uses them)."""
    if not isinstance(_kmat, np.ndarray):      # torch CUDA tensor: the matrix stays in HBM
        from . import svmcv
        ncv_ = max(2, int(args_svm[4]))
        if (n_pseqs + n_nseqs) * (ncv_ - 1) / ncv_ <= svmcv.MAX_FOLD_SAMPLES:
            return svmcv.crossValidate(args_svm, _kmat, n_pseqs, n_nseqs, about_to_launch, plan)
        logging.warning("folds of more than %d samples: cross-validation with scikit-learn on the host",
                        svmcv.MAX_FOLD_SAMPLES)
        _kmat = _kmat.cpu().numpy()
    from sklearn.model_selection import StratifiedKFold
    global _KMAT
    ncv, repeats, fast_estimation, random_seeds, p = args_svm[4:9]
    if fast_estimation != 0:
        raise NotImplementedError("fast AUC estimation is dead code in the reference (its regressor is never loaded)")
    if random_seeds < 0:
        random_seeds = None
    seqids = ["p%4d" % i for i in range(n_pseqs)] + ["n%4d" % i for i in range(n_nseqs)]
    y = np.concatenate((np.repeat(1, n_pseqs), np.repeat(0, n_nseqs)))
    jobs = []
    for _ in range(repeats):
        folds = StratifiedKFold(n_splits=ncv, shuffle=True, random_state=random_seeds)
        for train, test in folds.split(seqids, y):
            jobs.append((args_svm, y, train, test))
    _KMAT = _kmat
    try:
        if p > 1:
            with Pool(p) as pool:  # forked workers see _KMAT without copying it
                aucs = pool.map(_svm_fold, jobs)
        else:
            aucs = [_svm_fold(j) for j in jobs]
    finally:
        _KMAT = None
    logging.info("done cross-validation.")
    return (np.mean(aucs), np.std(aucs))


def init(pos_fa, neg_fa, args):
    args_gkm = [args.kernel_type, args.full_word_length, args.non_gap_length, args.max_num_gaps, args.init_decay,
                args.half_life_decay, args.rbf_gamma, pos_fa, neg_fa, args.n_processes, args.verbosity]
    logging.info("%s: building up kernel matrix", pos_fa)
    # GPU-resident path on one GPU: the process keeps ONE context for all its subsets (bin/gkmqc.py:341-343 calls init
    # once per subset) and does not wait for the matrix here -- the folds of the cross-validation are drawn (~10 ms of
    # scikit-learn) while the Gram kernel runs, and the solver is enqueued behind it on the same stream.
    on_gpu = getattr(args, "svm_solver", "gpu") == "gpu"
    one_gpu = not isinstance(getattr(args, "gpus", 1), (list, tuple)) and int(getattr(args, "gpus", 1)) == 1
    kmat, n_pseqs, n_nseqs = computeGkmKernel(args_gkm, resident=on_gpu, gpus=getattr(args, "gpus", 1),
                                              keep_context=on_gpu and one_gpu, wait=not (on_gpu and one_gpu))
    args_svm = [args.regularization, args.precision, args.shrinking, args.cache_size, args.ncv, args.repeats,
                args.fast_estimation, args.random_seeds, args.n_processes]
    logging.info("%s: svm training", pos_fa)
    auc_score, auc_std = crossValidate(args_svm, kmat, n_pseqs, n_nseqs)
    logging.info("%s: writing result to output file", pos_fa)
    with open(args.name + ".gkmqc.eval.out", "a") as fa:
        fa.write("\t".join(map(str, [pos_fa, neg_fa, n_pseqs, auc_score, auc_std])) + "\n")
    return auc_score, auc_std


def _init_many_on(pairs, args, gpu, slot):
    """The subsets of `pairs`, one after the other on one GPU, the cross-validation of subset s on a second HIP
    stream while the Gram kernel of subset s+1 has the rest of the GPU.  Returns [(auc, std, n_pos)]."""
    import queue
    import threading
    import torch
    dev = torch.device("cuda", gpu)
    gram_stream = torch.cuda.Stream(dev)
    cv_stream = torch.cuda.Stream(dev, priority=-1)
    args_svm = [args.regularization, args.precision, args.shrinking, args.cache_size, args.ncv, args.repeats,
                args.fast_estimation, args.random_seeds, args.n_processes]
    results = [None] * len(pairs)
    errors = []
    handoff = queue.Queue(maxsize=1)     # at most one finished matrix waits while the next one is computed
    # Launch order: the solver's workgroups are few and big (1024 threads, ~100 KB of LDS each), the Gram kernel's many
    # and small; once the Gram kernel of the next subset owns every CU the solver waits for room and the two barely
    # overlap (measured: 512 ms per subset against 522 one after the other).  So the next Gram kernel is held back
    # until the solver of this subset is on its way: it then takes the CUs it needs first and the Gram kernel the rest.
    solver_out = [threading.Event() for _ in pairs]

    def consumer():
        # Whatever goes wrong in here -- the import included -- is recorded for the calling thread, and every event the
        # producer may be waiting for is set on the way out: the producer never waits for a consumer that has died.
        try:
            from . import svmcv
            consume(svmcv)
        except BaseException as e:          # re-raised by the calling thread
            errors.append(e)
        finally:
            for ev in solver_out:
                ev.set()

    def consume(svmcv):
        plan = None
        while True:
            item = handoff.get()
            if item is None:
                return
            s, K, n_pos, n_neg = item
            if errors:
                solver_out[s].set()
                continue
            try:
                with torch.cuda.stream(cv_stream):
                    K.record_stream(cv_stream)
                    logging.info("%s: svm training", pairs[s][0])
                    auc, std = crossValidate(args_svm, K, n_pos, n_neg, about_to_launch=solver_out[s].set, plan=plan)
                    cv_stream.synchronize()
                results[s] = (auc, std, n_pos)
                # the next subset's folds, drawn while its matrix is being computed (the subsets of an evaluate run have
                # the same sizes; other sizes are planned when they arrive)
                plan = svmcv.plan_folds(args_svm, n_pos, n_neg) if s + 1 < len(pairs) else None
            except BaseException as e:      # re-raised by the calling thread
                errors.append(e)
            finally:
                solver_out[s].set()

    th = threading.Thread(target=consumer)
    th.start()
    try:
        for s, (pos_fa, neg_fa) in enumerate(pairs):
            if errors:
                break
            logging.info("%s: building up kernel matrix", pos_fa)
            args_gkm = [args.kernel_type, args.full_word_length, args.non_gap_length, args.max_num_gaps,
                        args.init_decay, args.half_life_decay, args.rbf_gamma, pos_fa, neg_fa, args.n_processes,
                        args.verbosity]
            with torch.cuda.stream(gram_stream):
                # (one context for all subsets: closing one would wait for the other stream's solver, hipFree)
                K, n_pos, n_neg = computeGkmKernel(args_gkm, gpu=gpu, resident=True, keep_context=True,
                                                   context_slot=slot)
                gram_stream.synchronize()
            while th.is_alive():         # (a consumer that died no longer takes anything: do not block on the queue)
                try:
                    handoff.put((s, K, n_pos, n_neg), timeout=1.0)
                    break
                except queue.Full:
                    continue
            del K
            while not solver_out[s].wait(timeout=1.0):   # (the FASTA files of the next subset are read after this)
                if not th.is_alive():
                    break
    finally:
        while th.is_alive():
            try:
                handoff.put(None, timeout=1.0)
                break
            except queue.Full:
                continue
        th.join()
        # the worker's context was kept from subset to subset; the run is over (the solver too: nothing left on the
        # device that its hipFree could wait for), so its scratch goes back before the caller allocates anything else
        device.release_cached_contexts(gpu, slot)
    if errors:
        raise errors[0]
    return results


def init_many(pairs, args, gpu=0, gpus=None):
    """`init` for several (pos_fa, neg_fa) subsets in a row -- what `bin/gkmqc.py evaluate` does with its peak
    subsets -- with the two GPU stages overlapped: the cross-validation of subset s (a handful of workgroups, one
    CU each) runs on a second HIP stream while the Gram kernel of subset s+1 has the rest of the GPU.  Same
    numbers and the same lines in <name>.gkmqc.eval.out, in order, as one `init` per subset; measured per subset
    at 5 000 + 5 000 sequences: 150 -> 117 ms (300 bp), 384 -> 343 ms (600 bp).

    gpus (a list of device ordinals): the subsets are independent, so they are dealt round-robin to one worker per
    entry -- each with its own context, streams and share of the subsets, no exchange between GPUs (the 20 subsets
    of an `evaluate` run on 8 GPUs: 3 rounds instead of 20).  The results and the eval lines keep the order of
    `pairs`."""
    import threading
    if getattr(args, "svm_solver", "gpu") != "gpu" or len(pairs) < 2:
        return [init(p, n, args) for p, n in pairs]
    devices = list(gpus) if gpus else [gpu]
    devices = devices[:len(pairs)]
    if len(devices) == 1:
        results = _init_many_on(pairs, args, devices[0], 0)
    else:
        results = [None] * len(pairs)
        errors = []

        def work(w):
            mine = list(range(w, len(pairs), len(devices)))
            try:
                for s, r in zip(mine, _init_many_on([pairs[s] for s in mine], args, devices[w], w)):
                    results[s] = r
            except BaseException as e:      # re-raised below
                errors.append(e)

        workers = [threading.Thread(target=work, args=(w,)) for w in range(len(devices))]
        for t in workers:
            t.start()
        for t in workers:
            t.join()
        if errors:
            raise errors[0]
    with open(args.name + ".gkmqc.eval.out", "a") as fa:
        for (pos_fa, neg_fa), (auc, std, n_pos) in zip(pairs, results):
            fa.write("\t".join(map(str, [pos_fa, neg_fa, n_pos, auc, std])) + "\n")
    return [(r[0], r[1]) for r in results]


def build_parser():
    parser = argparse.ArgumentParser(description="gkm-SVM cross-validation on an MI355X-computed gkm kernel matrix",
                                     formatter_class=argparse.RawTextHelpFormatter)
    parser.add_argument("-p", "--pos-fa", type=str, required=True, help="positive fa file. REQUIRED")
    parser.add_argument("-n", "--neg-fa", type=str, required=True, help="negative fa file. REQUIRED")
    parser.add_argument("-w", "--name", type=str, required=True, help="prefix of output file to write AUC score. REQUIRED")
    parser.add_argument("-s", "--random-seeds", type=int, default=-1, help="random seed (default: no seed)")
    parser.add_argument("-@", "--n-processes", type=int, default=1, help="number of processes (default: 1)")
    parser.add_argument("-v", "--verbosity", type=int, default=1, help="verbosity (default: 1), 0: silent")
    g = parser.add_argument_group("gkm-kernel")
    g.add_argument("-t", "--kernel-type", type=int, default=4,
                   help="0 gapped-kmer, 1 estimated l-mer full filter, 2 truncated filter (gkm), 3 gkm+RBF,\n"
                        "4 gkm + centre weighted (wgkm, default), 5 wgkm+RBF")
    g.add_argument("-L", "--full-word-length", type=int, default=10, help="full word length, 3<=L<=12 (default: 10)")
    g.add_argument("-k", "--non-gap-length", type=int, default=6, help="non-gap positions, k<=L (default: 6)")
    g.add_argument("-d", "--max-num-gaps", type=int, default=3, help="max gaps, d<=min(4, L-k) (default: 3)")
    g.add_argument("-M", "--init-decay", type=int, default=50, help="initial value of the decay, -t 4/5 (default: 50)")
    g.add_argument("-H", "--half-life-decay", type=int, default=50, help="half life of the decay, -t 4/5 (default: 50)")
    g.add_argument("-G", "--rbf-gamma", type=float, default=1.0, help="gamma for RBF kernels, -t 3/5 (default: 1.0)")
    v = parser.add_argument_group("SVM training")
    v.add_argument("-C", "--regularization", type=float, default=1.0, help="regularization parameter C (default: 1.0)")
    v.add_argument("-e", "--precision", type=float, default=0.001, help="precision parameter epsilon (default: 0.001)")
    v.add_argument("-u", "--shrinking", type=int, default=0, help="use the shrinking heuristics (default: 0)")
    v.add_argument("-c", "--cache-size", type=int, default=512, help="cache memory size in MB (default: 512)")
    v.add_argument("-x", "--ncv", type=int, default=5, help="x-fold cross validation (default: 5)")
    v.add_argument("-r", "--repeats", type=int, default=1, help="repeats of CV training (default: 1)")
    v.add_argument("--svm-solver", choices=("gpu", "sklearn"), default="gpu",
                   help="gpu: all folds on the GPU-resident matrix (default); sklearn: the reference's solver")
    parser.add_argument("--gpus", type=int, default=1,
                        help="GPUs of this node that share the kernel matrix (row blocks + RCCL all-gather; default: 1)")
    v.add_argument("-f", "--fast-estimation", type=int, default=0, help="not supported (dead code in the reference)")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    return init(args.pos_fa, args.neg_fa, args)


if __name__ == "__main__":
    logging.basicConfig(stream=sys.stdout, format="%(levelname)s %(asctime)s: %(message)s",
                        datefmt="%Y-%m-%d %H:%M:%S", level=logging.INFO)
    main()
