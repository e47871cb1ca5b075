"""Cross-validation of a C-SVC on a gkm kernel matrix that stays in HBM -- SURVEY.md §8(f4).

The reference trains `sklearn.svm.SVC(kernel="precomputed")` once per fold on host copies of the
matrix (`scripts/gkmsvm.py:104-125`, fancy-indexed sub-matrices, one process per fold).  Here the
matrix never leaves the GPU: every fold of every repeat is one workgroup of `k_smo`
(gkmqc_amd/csrc/gkm_svm.hip, include/gkm_svm.h) reading the resident matrix through index lists,
and the decision values come from `k_decision`.  Only the index lists go up and the per-fold
decision values (a few thousand doubles) come back; fold generation (StratifiedKFold) and the AUC
stay scikit-learn's so that the folds and the metric are the reference's own.

Parity (tests/test_svm_gpu.py): dual coefficients, intercept and decision values bit-identical to
scikit-learn's LIBSVM on the same matrix, hence identical AUC.
"""
import ctypes
import logging

import numpy as np

from . import device


SHAPE_REFUSED = 5        # GKMSVM_RC_SHAPE_REFUSED (include/gkm_svm.h)
FAST_FOLD_SAMPLES = 16384  # k_smo: one workgroup per fold, 1024 threads x 16 samples in registers (gkm_svm.hip)
MAX_FOLD_SAMPLES = 60000   # k_smo_general: state in global memory (scanned part in LDS up to 8 192 samples), also the solver with LIBSVM's shrinking


class SvmError(RuntimeError):
    pass


def _lib():
    L = device.load()
    if not hasattr(L, "_svm_bound"):
        vp, i32, i64, dbl = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double
        L.gkmsvm_train_batch.restype = i32
        L.gkmsvm_train_batch.argtypes = (i32, vp, i64, i32, i32, vp, vp, vp, dbl, dbl, vp, vp, vp, vp, vp)
        L.gkmsvm_train_batch_general.restype = i32
        L.gkmsvm_train_batch_general.argtypes = (i32, vp, i64, i32, i32, vp, vp, vp, dbl, dbl, i32, vp, vp, vp, vp, vp)
        L.gkmsvm_decision_batch.restype = i32
        L.gkmsvm_decision_batch.argtypes = (i32, vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp)
        L.gkmsvm_last_error.restype = ctypes.c_char_p
        L.gkmsvm_release_cache.restype = None
        L._svm_bound = True
    return L


def libsvm_order(train, y):
    """Training indices in LIBSVM's internal order (sklearn's svm_group_classes: classes sorted by
    label, the samples of a class in their original order) and the size of class 0."""
    train = np.asarray(train)
    lab = np.asarray(y)[train]
    classes = np.unique(lab)
    if len(classes) != 2:
        raise SvmError("a fold needs samples of exactly two classes")
    first = train[lab == classes[0]]
    second = train[lab == classes[1]]
    return np.concatenate((first, second)).astype(np.int32), int(len(first))


class FoldSolutions:
    """Result of `train_folds`: per-fold alpha (LIBSVM order), rho, iteration count."""

    def __init__(self, idx, n0, alpha, grad, rho, iters):
        self.idx, self.n0, self.alpha, self.grad, self.rho, self.iters = idx, n0, alpha, grad, rho, iters

    def dual_coef(self, f):
        """sklearn's `dual_coef_[0]` and `support_` of fold f (positions in the fold's `train`)."""
        a = self.alpha[f]
        ysign = np.where(np.arange(len(a)) < self.n0[f], 1.0, -1.0)
        sv = np.nonzero(a > 0)[0]
        return -(a[sv] * ysign[sv]), self.idx[f][sv]


def train_folds(K, trains, y, C=1.0, tol=1e-3, shrinking=False, about_to_launch=None):
    """Solve one C-SVC per entry of `trains` (index arrays into the symmetric torch CUDA fp64
    matrix K) concurrently.  Returns (FoldSolutions, device handles for `decision_values`).
    `shrinking`: LIBSVM's shrinking heuristic (scikit-learn `SVC(shrinking=True)`); it and folds of more
    than 16 384 samples run on the general solver (k_smo_general), everything else on k_smo."""
    import torch
    if not (K.is_cuda and K.dtype == torch.float64 and K.dim() == 2 and K.shape[0] == K.shape[1]
            and K.stride(1) == 1):
        raise SvmError("K must be a square fp64 CUDA tensor with unit column stride")
    L = _lib()
    dev = K.device
    orders = [libsvm_order(t, y) for t in trains]
    idx = [o[0] for o in orders]
    n0 = np.array([o[1] for o in orders], dtype=np.int32)
    off = np.zeros(len(idx) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(i) for i in idx])
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream().cuda_stream
        d_idx = torch.from_numpy(np.concatenate(idx)).to(dev)
        d_alpha = torch.empty(int(off[-1]), dtype=torch.float64, device=dev)
        d_grad = torch.empty_like(d_alpha)
        d_rho = torch.empty(len(idx), dtype=torch.float64, device=dev)
        d_it = torch.empty(len(idx), dtype=torch.int32, device=dev)
        if about_to_launch is not None:
            about_to_launch()     # everything on the host is done: the solver kernel goes out within microseconds
        if shrinking or max(len(i) for i in idx) > FAST_FOLD_SAMPLES:
            rc = L.gkmsvm_train_batch_general(dev.index or 0, K.data_ptr(), K.stride(0), K.shape[0], len(idx),
                                              d_idx.data_ptr(), off.ctypes.data, n0.ctypes.data, float(C), float(tol),
                                              1 if shrinking else 0, d_alpha.data_ptr(), d_grad.data_ptr(),
                                              d_rho.data_ptr(), d_it.data_ptr(), stream)
        else:
            rc = L.gkmsvm_train_batch(dev.index or 0, K.data_ptr(), K.stride(0), K.shape[0], len(idx), d_idx.data_ptr(),
                                      off.ctypes.data, n0.ctypes.data, float(C), float(tol), d_alpha.data_ptr(),
                                      d_grad.data_ptr(), d_rho.data_ptr(), d_it.data_ptr(), stream)
            if rc == SHAPE_REFUSED:
                # ONLY a launch shape the device refuses (the big shapes take up to 144 KB of LDS; include/gkm_svm.h):
                # the general solver needs none of that and returns the same bits without shrinking.  Bad arguments
                # or a device fault are not retried -- the retry would fail too and hide the cause.
                first = L.gkmsvm_last_error().decode()
                logging.warning("k_smo: %s: solving with the general GPU solver", first)
                rc = L.gkmsvm_train_batch_general(dev.index or 0, K.data_ptr(), K.stride(0), K.shape[0], len(idx),
                                                  d_idx.data_ptr(), off.ctypes.data, n0.ctypes.data, float(C), float(tol),
                                                  0, d_alpha.data_ptr(), d_grad.data_ptr(), d_rho.data_ptr(),
                                                  d_it.data_ptr(), stream)
                if rc:
                    raise SvmError("gkmsvm_train_batch: %s; then gkmsvm_train_batch_general: %s"
                                   % (first, L.gkmsvm_last_error().decode()))
        if rc:
            raise SvmError("gkmsvm_train_batch: %s" % L.gkmsvm_last_error().decode())
        alpha = d_alpha.cpu().numpy()
        grad = d_grad.cpu().numpy()
        rho = d_rho.cpu().numpy()
        iters = d_it.cpu().numpy()
    if (iters < 0).any():
        # scikit-learn (max_iter=-1) would have kept iterating: these folds are not the reference's solution
        logging.warning("SMO stopped at the iteration cap in %d fold(s)", int((iters < 0).sum()))
    sol = FoldSolutions(idx, n0, [alpha[off[f]:off[f + 1]] for f in range(len(idx))],
                        [grad[off[f]:off[f + 1]] for f in range(len(idx))], rho, iters)
    return sol, dict(idx=d_idx, off=off, n0=n0, alpha=d_alpha, rho=d_rho)


def decision_values(K, handles, tests):
    """scikit-learn's `decision_function` of every fold on its test samples (list of arrays)."""
    import torch
    L = _lib()
    dev = K.device
    toff = np.zeros(len(tests) + 1, dtype=np.int64)
    toff[1:] = np.cumsum([len(t) for t in tests])
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream().cuda_stream
        d_test = torch.from_numpy(np.concatenate(tests).astype(np.int32)).to(dev)
        d_dec = torch.empty(int(toff[-1]), dtype=torch.float64, device=dev)
        rc = L.gkmsvm_decision_batch(dev.index or 0, K.data_ptr(), K.stride(0), len(tests), handles["idx"].data_ptr(),
                                     handles["off"].ctypes.data, handles["n0"].ctypes.data,
                                     handles["alpha"].data_ptr(), handles["rho"].data_ptr(), d_test.data_ptr(),
                                     toff.ctypes.data, d_dec.data_ptr(), stream)
        if rc:
            raise SvmError("gkmsvm_decision_batch: %s" % L.gkmsvm_last_error().decode())
        dec = d_dec.cpu().numpy()
    # sklearn flips the sign of LIBSVM's decision value for a two-class problem
    return [-dec[toff[f]:toff[f + 1]] for f in range(len(tests))]


def device_block(K, rows, cols):
    """K[rows][:, cols] of the device matrix (torch tensor), still on the device."""
    import torch
    r = torch.as_tensor(np.asarray(rows), device=K.device, dtype=torch.long)
    c = torch.as_tensor(np.asarray(cols), device=K.device, dtype=torch.long)
    return K.index_select(0, r).index_select(1, c)


def plan_folds(args_svm, n_pseqs, n_nseqs):
    """The host-side half of `crossValidate` that does not need the matrix: labels and the ncv x repeats stratified folds
    (the reference's splitter and seeding, scripts/gkmsvm.py:134-150), equal folds solved once.  ~10 ms of scikit-learn
    at 10 000 sequences -- a pipeline draws the next subset's folds while the GPU is still busy with this one's."""
    from sklearn.model_selection import StratifiedKFold
    ncv, repeats, random_seeds = args_svm[4], args_svm[5], args_svm[7]
    if random_seeds is not None and random_seeds < 0:
        random_seeds = None
    seqids = ["p%4d" % i for i in range(n_pseqs)] + ["n%4d" % i for i in range(n_nseqs)]
    y = np.concatenate((np.repeat(1, n_pseqs), np.repeat(0, n_nseqs)))
    trains, tests = [], []
    for _ in range(repeats):
        folds = StratifiedKFold(n_splits=ncv, shuffle=True, random_state=random_seeds)
        for train, test in folds.split(seqids, y):
            trains.append(train)
            tests.append(test)
    # With a fixed seed the reference builds every repeat from the same random_state
    # (scripts/gkmsvm.py:148), i.e. the same folds again: solve each distinct fold once.
    first, which = {}, []
    for f, train in enumerate(trains):
        which.append(first.setdefault(train.tobytes(), len(first)))
    uniq = sorted(set(which))
    u_of = {w: [f for f in range(len(trains)) if which[f] == w][0] for w in uniq}
    return dict(sizes=(n_pseqs, n_nseqs), y=y, n_folds=len(trains), which=which,
                u_trains=[trains[u_of[w]] for w in uniq], u_tests=[tests[u_of[w]] for w in uniq])


def crossValidate(args_svm, K, n_pseqs, n_nseqs, about_to_launch=None, plan=None):
    """Same arguments and result as the reference's `crossValidate` (scripts/gkmsvm.py:127-176)
    with `K` a symmetric torch CUDA matrix: (mean AUC, std AUC) over ncv x repeats folds.
    about_to_launch: called once, right before the solver kernel is enqueued (init_many holds the next subset's Gram
    kernel back until then, so that the solver's few big workgroups find the CUs they need).
    plan: the folds drawn ahead of time by `plan_folds` for these sizes (otherwise drawn here)."""
    from sklearn.metrics import roc_auc_score
    regularization, precision, shrinking, _cache, _ncv, _repeats, fast_estimation = args_svm[:7]
    if fast_estimation != 0:
        raise NotImplementedError("fast AUC estimation is dead code in the reference (its regressor is never loaded)")
    if plan is None or plan["sizes"] != (n_pseqs, n_nseqs):
        plan = plan_folds(args_svm, n_pseqs, n_nseqs)
    y, which, u_trains, u_tests = plan["y"], plan["which"], plan["u_trains"], plan["u_tests"]
    logging.info("cross-validation on the GPU: %d folds (%d distinct)", plan["n_folds"], len(u_trains))
    sol, handles = train_folds(K, u_trains, y, regularization, precision, bool(shrinking), about_to_launch)
    scores = decision_values(K, handles, u_tests)
    capped = [f for f in range(len(u_trains)) if sol.iters[f] < 0]
    if capped:   # not converged within 10^7 iterations: the reference's solver has no cap, so use it for these folds
        from sklearn.svm import SVC
        logging.warning("%d fold(s) re-solved with scikit-learn (iteration cap of the GPU solver)", len(capped))
        for f in capped:
            tr, te = u_trains[f], u_tests[f]
            ktr = device_block(K, tr, tr).cpu().numpy()
            kte = device_block(K, te, tr).cpu().numpy()
            sv = SVC(kernel="precomputed", C=regularization, tol=precision, shrinking=bool(shrinking), cache_size=_cache)
            scores[f] = sv.fit(ktr, y[tr]).decision_function(kte)
            sol.alpha[f] = np.abs(sv.dual_coef_[0])     # (only its sum is used below)
    u_auc = []
    for f, (test, score) in enumerate(zip(u_tests, scores)):
        auc = roc_auc_score(y[test], score)
        nu = np.sum(sol.alpha[f]) / len(u_trains[f])
        logging.info("fold %d solved on the GPU: nu %.3f, AUC %.3f, %d iterations", f, nu, auc, abs(int(sol.iters[f])))
        u_auc.append(auc)
    aucs = [u_auc[w] for w in which]
    logging.info("cross-validation finished")
    return (np.mean(aucs), np.std(aucs))
