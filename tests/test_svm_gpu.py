"""GPU-resident C-SVC (gkmqc_amd/csrc/gkm_svm.hip, include/gkm_svm.h) against
scikit-learn's LIBSVM -- the solver the reference calls (scripts/gkmsvm.py:104-125).  The bar is
bit-identity of everything the solver returns (support set, dual coefficients, intercept, decision
values), which implies identical AUC; the cross-validation is also checked against the AUCs the
reference's own module produced (tests/golden/gkmsvm_expected.json)."""
import json
import os

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

EXPECTED = json.load(open(os.path.join(helpers.GOLDEN, "gkmsvm_expected.json")))
POS = os.path.join(helpers.GOLDEN, "motif_pos.fa")
NEG = os.path.join(helpers.GOLDEN, "motif_neg.fa")


def _rbf_matrix(n, dim, seed, dup=0):
    """A positive-definite test kernel with class structure; `dup` duplicated points force exact
    ties in the working-set selection."""
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, dim))
    X[: n // 2] += 0.6
    if dup:
        X[n - dup:] = X[:dup]
    d2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)
    K = np.exp(-d2 / (2.0 * dim))
    return np.maximum(K, K.T)


def _compare_with_sklearn(K, n_first, trains, tests, C, tol, shrinking=False):
    import torch
    from sklearn.svm import SVC
    from gkmqc_amd import svmcv
    n = K.shape[0]
    y = np.concatenate((np.repeat(1, n_first), np.repeat(0, n - n_first)))
    Kd = torch.from_numpy(K).cuda()
    sol, handles = svmcv.train_folds(Kd, trains, y, C, tol, shrinking)
    scores = svmcv.decision_values(Kd, handles, tests)
    for f, (train, test) in enumerate(zip(trains, tests)):
        sv = SVC(kernel="precomputed", C=C, tol=tol, shrinking=shrinking, cache_size=512)
        sv.fit(K[train][:, train], y[train])
        coef, support = sol.dual_coef(f)
        # sklearn's support_ are positions in `train`
        pos_in_train = {g: p for p, g in enumerate(train)}
        got_support = np.array([pos_in_train[g] for g in support])
        assert sol.iters[f] > 0
        assert np.array_equal(got_support, sv.support_), "fold %d: support set differs" % f
        assert np.array_equal(coef, sv.dual_coef_[0]), "fold %d: max |diff| %g" % (
            f, np.abs(coef - sv.dual_coef_[0]).max())
        assert sol.rho[f] == sv.intercept_[0]
        want = sv.decision_function(K[test][:, train])
        assert np.array_equal(scores[f], want), "fold %d: decision values differ by %g" % (
            f, np.abs(scores[f] - want).max())


def _folds(n, n_first, ncv, seed):
    from sklearn.model_selection import StratifiedKFold
    y = np.concatenate((np.repeat(1, n_first), np.repeat(0, n - n_first)))
    sp = StratifiedKFold(n_splits=ncv, shuffle=True, random_state=seed).split(np.zeros(n), y)
    trains, tests = zip(*sp)
    return list(trains), list(tests)


@pytest.mark.parametrize("n,dim,C,tol,dup", [
    (200, 6, 1.0, 1e-3, 0),
    (600, 10, 1.0, 1e-3, 0),
    (600, 10, 0.05, 1e-3, 0),      # most alphas at the upper bound
    (400, 4, 100.0, 1e-4, 0),      # few bounded, many iterations
    (300, 5, 1.0, 1e-3, 40),       # duplicated samples: exact ties in the selection
    (2600, 12, 1.0, 1e-3, 0),      # more samples than threads (several per thread)
])
def test_solver_is_bit_identical_to_sklearn(built, n, dim, C, tol, dup):
    K = _rbf_matrix(n, dim, seed=n + dim, dup=dup)
    trains, tests = _folds(n, n // 2, 3, seed=1)
    _compare_with_sklearn(K, n // 2, trains, tests, C, tol)


@pytest.mark.parametrize("n,dim,C,tol,dup", [
    (200, 6, 1.0, 1e-3, 0),
    (600, 10, 1.0, 1e-3, 0),
    (600, 10, 0.05, 1e-3, 0),      # most alphas at the upper bound: G_bar in use
    (400, 4, 100.0, 1e-4, 0),      # few bounded, many iterations: several rounds of shrinking, unshrink
    (300, 5, 1.0, 1e-3, 40),       # duplicated samples: ties decided by the (permuted) position
    (2600, 12, 1.0, 1e-3, 0),      # more than 1000 samples (shrinking every 1000 iterations), several per thread
    (3000, 3, 10.0, 1e-5, 100),
])
def test_general_solver_with_shrinking_is_bit_identical_to_sklearn(built, n, dim, C, tol, dup):
    """LIBSVM's shrinking (`--shrinking 1`): active-set permutation, G_bar, gradient reconstruction -- the support
    set, dual coefficients, intercept and decision values of scikit-learn's SVC(shrinking=True), bit for bit."""
    K = _rbf_matrix(n, dim, seed=n + dim, dup=dup)
    trains, tests = _folds(n, n // 2, 3, seed=1)
    _compare_with_sklearn(K, n // 2, trains, tests, C, tol, shrinking=True)
    if n == 3000:   # tens of thousands of iterations: shrinking changes LIBSVM's own path, so the match above means something
        from sklearn.svm import SVC
        y = np.concatenate((np.repeat(1, n // 2), np.repeat(0, n - n // 2)))
        differs = False
        for tr in trains:
            a, b = (SVC(kernel="precomputed", C=C, tol=tol, shrinking=sh).fit(K[tr][:, tr], y[tr]) for sh in (True, False))
            differs = differs or int(a.n_iter_[0]) != int(b.n_iter_[0]) or not np.array_equal(a.dual_coef_, b.dual_coef_)
        assert differs


def test_general_solver_without_shrinking(built, monkeypatch):
    """The same kernel with shrinking off is the path of folds beyond k_smo's 16 384 samples; forced here on
    small folds (FAST_FOLD_SAMPLES lowered), plus unbalanced and two-sample problems with shrinking on."""
    from gkmqc_amd import svmcv
    monkeypatch.setattr(svmcv, "FAST_FOLD_SAMPLES", 10)
    K = _rbf_matrix(700, 8, seed=11, dup=10)
    trains, tests = _folds(700, 350, 2, seed=2)
    _compare_with_sklearn(K, 350, trains, tests, 1.0, 1e-3)
    K = _rbf_matrix(90, 3, seed=5)
    trains, tests = _folds(90, 12, 4, seed=3)
    _compare_with_sklearn(K, 12, trains, tests, 1.0, 1e-3, shrinking=True)
    _compare_with_sklearn(K, 12, [np.array([0, 50])], [np.array([1, 2, 60])], 1.0, 1e-3, shrinking=True)


@pytest.mark.parametrize("shrinking", [False, True])
def test_general_solver_variants_agree(built, monkeypatch, shrinking):
    """The general solver exists three times: scanned state in LDS (folds of at most 8 192 samples, what the tests above
    ran), the same 512 threads with the state in global memory (GKM_SVM_GEN_LDS=0), and 1 024 threads (larger folds;
    GKM_SVM_GEN_T=1024).  Same alpha, gradient, rho and iteration count from all three, bit for bit, on a problem with
    bounded and free alphas, ties and -- with shrinking -- several rounds of it."""
    from gkmqc_amd import svmcv
    monkeypatch.setattr(svmcv, "FAST_FOLD_SAMPLES", 0)
    K = _rbf_matrix(3000, 3, seed=21, dup=60)
    import torch
    Kd = torch.from_numpy(K).cuda()
    y = np.concatenate((np.repeat(1, 1500), np.repeat(0, 1500)))
    trains, _ = _folds(3000, 1500, 2, seed=4)
    got = []
    for env in ({}, {"GKM_SVM_GEN_LDS": "0"}, {"GKM_SVM_GEN_T": "1024"}):
        for k in ("GKM_SVM_GEN_LDS", "GKM_SVM_GEN_T"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sol, _ = svmcv.train_folds(Kd, trains, y, 10.0, 1e-5, shrinking)
        got.append(sol)
    a = got[0]
    assert min(a.iters) > 1000
    for b in got[1:]:
        assert list(a.iters) == list(b.iters)
        for f in range(len(trains)):
            assert np.array_equal(a.alpha[f], b.alpha[f]) and np.array_equal(a.grad[f], b.grad[f]) and a.rho[f] == b.rho[f]


def test_large_folds_both_solvers_agree(built, monkeypatch):
    """16 000-sample folds (config 3's size): k_smo's 16-samples-per-thread shape (alpha in LDS, indices and
    diagonal re-read) against the general solver, which keeps its state in global memory -- same alpha, gradient,
    rho and iteration count bit for bit; and a fold beyond k_smo's 16 384 samples converges on the general solver
    (KKT violation below the tolerance).  scikit-learn itself would take minutes here; both solvers are pinned
    to it at smaller sizes above."""
    import torch
    from gkmqc_amd import svmcv
    n, dim = 17600, 6
    g = torch.Generator(device="cpu").manual_seed(5)
    X = torch.randn(n, dim, generator=g, dtype=torch.float64)
    X[: n // 2] += 0.5
    X = X.cuda()
    sq = (X * X).sum(1)
    K = torch.exp(-(sq[:, None] + sq[None, :] - 2.0 * X @ X.T).clamp_min(0) / (2.0 * dim))
    K = torch.maximum(K, K.T).contiguous()
    y = np.concatenate((np.repeat(1, n // 2), np.repeat(0, n - n // 2)))
    idx = np.random.default_rng(3).permutation(n)
    train = np.sort(idx[:16000])
    a, _ = svmcv.train_folds(K, [train], y, 1.0, 1e-3)
    monkeypatch.setattr(svmcv, "FAST_FOLD_SAMPLES", 0)
    b, _ = svmcv.train_folds(K, [train], y, 1.0, 1e-3)
    assert a.iters[0] > 0 and a.iters[0] == b.iters[0]
    assert np.array_equal(a.alpha[0], b.alpha[0]) and np.array_equal(a.grad[0], b.grad[0]) and a.rho[0] == b.rho[0]
    monkeypatch.setattr(svmcv, "FAST_FOLD_SAMPLES", 16384)
    big = np.sort(idx[:17500])
    c, _ = svmcv.train_folds(K, [big], y, 1.0, 1e-3)
    assert c.iters[0] > 0
    al, gr = c.alpha[0], c.grad[0]
    ys = np.where(np.arange(len(al)) < c.n0[0], 1.0, -1.0)
    up = ((ys > 0) & (al < 1.0)) | ((ys < 0) & (al > 0))
    low = ((ys > 0) & (al > 0)) | ((ys < 0) & (al < 1.0))
    assert (-(ys * gr))[up].max() + (ys * gr)[low].max() < 1e-3      # LIBSVM's stopping criterion


def test_cross_validation_with_shrinking_matches_sklearn(built, tmp_path):
    """`--shrinking 1` end to end: the GPU cross-validation against the reference's scikit-learn harness on the
    same (gkm) matrix."""
    from gkmqc_amd import gkmsvm
    case = EXPECTED["wgkm_L10"]
    a = list(case["args_gkm"])
    a[7], a[8] = POS, NEG
    Kd, n_pos, n_neg = gkmsvm.computeGkmKernel(a, resident=True)
    args_svm = list(case["args_svm"])
    args_svm[2] = 1
    got = gkmsvm.crossValidate(args_svm, Kd, n_pos, n_neg)
    want = gkmsvm.crossValidate(args_svm, Kd.cpu().numpy(), n_pos, n_neg)
    assert got == want
    C, tol = args_svm[0], args_svm[1]
    trains, tests = _folds(n_pos + n_neg, n_pos, args_svm[4], seed=args_svm[7])
    _compare_with_sklearn(Kd.cpu().numpy(), n_pos, trains, tests, C, tol, shrinking=True)


@pytest.mark.parametrize("shape", ["256x8", "512x8", "1024x4", "1024x8", "1024x10", "1024x12", "512x16", "1024x16"])
def test_every_launch_shape(built, monkeypatch, shape):
    """The launcher picks threads x samples-per-thread from the fold size; force each instantiation
    (incl. the register-table variant used above 8192 samples) on the same problem."""
    monkeypatch.setenv("GKM_SVM_SHAPE", shape)
    K = _rbf_matrix(700, 8, seed=11, dup=10)
    trains, tests = _folds(700, 350, 2, seed=2)
    _compare_with_sklearn(K, 350, trains, tests, 1.0, 1e-3)


def test_unbalanced_and_tiny_folds(built):
    K = _rbf_matrix(90, 3, seed=5)
    trains, tests = _folds(90, 12, 4, seed=3)
    _compare_with_sklearn(K, 12, trains, tests, 1.0, 1e-3)
    # a two-sample problem
    _compare_with_sklearn(K, 12, [np.array([0, 50])], [np.array([1, 2, 60])], 1.0, 1e-3)


@pytest.mark.parametrize("name", sorted(EXPECTED))
def test_gkm_matrix_solver_and_cv_match_reference(built, name):
    """gkm matrix computed on the GPU, left in HBM, cross-validated there: per-fold bit-identity with
    sklearn and the AUC mean/std of the reference's module."""
    from gkmqc_amd import gkmsvm
    case = EXPECTED[name]
    a = list(case["args_gkm"])
    a[7], a[8] = POS, NEG
    Kd, n_pos, n_neg = gkmsvm.computeGkmKernel(a, resident=True)
    assert Kd.is_cuda
    auc, std = gkmsvm.crossValidate(list(case["args_svm"]), Kd, n_pos, n_neg)
    assert abs(auc - case["auc_mean"]) < 1e-12 and abs(std - case["auc_std"]) < 1e-12
    C, tol = case["args_svm"][0], case["args_svm"][1]
    trains, tests = _folds(n_pos + n_neg, n_pos, case["args_svm"][4], seed=case["args_svm"][7])
    _compare_with_sklearn(Kd.cpu().numpy(), n_pos, trains, tests, C, tol)


def test_main_uses_the_gpu_solver(built, tmp_path):
    from gkmqc_amd import gkmsvm
    case = EXPECTED["wgkm_L10"]
    out = str(tmp_path / "run_gpu")
    for solver in ("gpu", "sklearn"):
        auc, std = gkmsvm.main(["-p", POS, "-n", NEG, "-w", out, "-s", "7", "-v", "0", "-t", "4", "-L", "10", "-k", "6",
                                "-d", "3", "-r", "2", "--svm-solver", solver])
        assert abs(auc - case["auc_mean"]) < 1e-12 and abs(std - case["auc_std"]) < 1e-12


def test_iteration_cap_falls_back_to_sklearn(built, monkeypatch):
    """scikit-learn has no iteration cap, the GPU solver stops at 10^7 (LIBSVM's classic default) and says
    so; the cross-validation then re-solves that fold with scikit-learn.  Forced here with a tiny cap."""
    from gkmqc_amd import gkmsvm
    case = EXPECTED["wgkm_L10"]
    a = list(case["args_gkm"])
    a[7], a[8] = POS, NEG
    Kd, n_pos, n_neg = gkmsvm.computeGkmKernel(a, resident=True)
    monkeypatch.setenv("GKM_SVM_MAX_ITER", "7")
    auc, std = gkmsvm.crossValidate(list(case["args_svm"]), Kd, n_pos, n_neg)
    assert abs(auc - case["auc_mean"]) < 1e-12 and abs(std - case["auc_std"]) < 1e-12


def test_bad_arguments_fail_loudly(built):
    import torch
    from gkmqc_amd import svmcv
    K = torch.eye(8, dtype=torch.float64)
    with pytest.raises(svmcv.SvmError):
        svmcv.train_folds(K, [np.arange(8)], np.array([1] * 4 + [0] * 4))        # not on the GPU
    with pytest.raises(svmcv.SvmError):
        svmcv.train_folds(K.cuda(), [np.arange(4)], np.array([1] * 4 + [0] * 4))  # one class only


def test_refused_launch_shape_falls_back_to_the_general_solver(built, monkeypatch, caplog):
    """gkmsvm_train_batch reports a launch shape the device refuses with its own return code
    (GKMSVM_RC_SHAPE_REFUSED, include/gkm_svm.h) and svmcv retries THAT failure -- and only that one -- with the
    general solver: same bits as scikit-learn.  Bad arguments are not retried and raise with their own message."""
    import logging
    import torch
    from gkmqc_amd import svmcv
    K = _rbf_matrix(300, 6, seed=11)
    trains, tests = _folds(300, 150, 3, seed=2)
    monkeypatch.setenv("GKM_SVM_SHAPE", "2048x16")          # big enough for the folds, not a shape the library has
    with caplog.at_level(logging.WARNING):
        _compare_with_sklearn(K, 150, trains, tests, 1.0, 1e-3)
    assert any("general GPU solver" in r.getMessage() for r in caplog.records)
    monkeypatch.delenv("GKM_SVM_SHAPE")
    y = np.concatenate((np.repeat(1, 150), np.repeat(0, 150)))
    with pytest.raises(svmcv.SvmError) as e:                 # a fold without samples: an argument error, no retry
        svmcv.train_folds(torch.from_numpy(K).cuda(), [trains[0], np.zeros(0, dtype=np.int64)], y, 1.0, 1e-3)
    assert "general" not in str(e.value)
