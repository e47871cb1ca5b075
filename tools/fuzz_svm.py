"""Randomised check of the GPU C-SVC against scikit-learn's LIBSVM: random kernels (RBF on random
features, low-rank linear + ridge, near-duplicate columns), sizes 8..3000, class balance, C, tol,
duplicated samples; dual coefficients, support set, intercept and decision values must be
bit-identical.   python tools/fuzz_svm.py [--seconds 180] [--seed 1] [--shrinking]
(--shrinking: LIBSVM's shrinking heuristic on both sides -- the general solver k_smo_general)"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=180)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--shrinking", action="store_true")
    a = ap.parse_args()
    import torch
    from sklearn.svm import SVC
    from gkmqc_amd import svmcv
    rng = np.random.default_rng(a.seed)
    t_end = time.time() + a.seconds
    cases = capped = shrunk = 0
    while time.time() < t_end:
        n = int(rng.choice([8, 20, 60, 200, 500, 1200, 3000]))
        dim = int(rng.integers(1, 12))
        X = rng.normal(size=(n, dim))
        n1 = int(rng.integers(max(1, n // 10), n - max(1, n // 10) + 1))
        X[:n1] += rng.uniform(0.0, 1.5)
        kind = int(rng.integers(0, 3))
        if kind == 0:
            d2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1) if n <= 1200 else None
            if d2 is None:
                sq = (X * X).sum(1)
                d2 = np.maximum(sq[:, None] + sq[None, :] - 2 * X @ X.T, 0)
            K = np.exp(-d2 / (2.0 * dim * rng.uniform(0.3, 3)))
        elif kind == 1:
            Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
            K = Xn @ Xn.T + 1e-3 * np.eye(n)
        else:
            K = (1 + X @ X.T / dim) ** 2
            dg = np.sqrt(np.diag(K))
            K = K / dg[:, None] / dg[None, :]
        K = np.maximum(K, K.T)
        ndup = int(rng.integers(0, max(1, n // 8)))
        if ndup:                                    # exact duplicates: ties in the working-set selection
            src = rng.integers(0, n, ndup)
            dst = rng.integers(0, n, ndup)
            K[dst, :] = K[src, :]
            K[:, dst] = K[:, src]
            K = np.maximum(K, K.T)
        y = np.concatenate((np.repeat(1, n1), np.repeat(0, n - n1)))
        C = float(rng.choice([0.01, 0.1, 1.0, 10.0, 100.0]))
        tol = float(rng.choice([1e-2, 1e-3, 1e-4, 1e-5] if a.shrinking else [1e-2, 1e-3, 1e-4]))
        idx = rng.permutation(n)
        ntr = max(2, int(n * rng.uniform(0.5, 0.95)))
        train = np.sort(idx[:ntr])
        test = np.sort(idx[ntr:]) if ntr < n else np.array([0])
        if len(np.unique(y[train])) < 2:
            continue
        Kd = torch.from_numpy(K).cuda()
        sol, h = svmcv.train_folds(Kd, [train], y, C, tol, a.shrinking)
        if sol.iters[0] < 0:      # 10^7 iterations without convergence: scikit-learn (no cap) goes on, see svmcv.py
            capped += 1
            continue
        dec = svmcv.decision_values(Kd, h, [test])[0]
        sv = SVC(kernel="precomputed", C=C, tol=tol, shrinking=a.shrinking, cache_size=512).fit(K[train][:, train], y[train])
        coef, support = sol.dual_coef(0)
        pos = {g: p for p, g in enumerate(train)}
        ok = (np.array_equal(np.array([pos[g] for g in support]), sv.support_) and np.array_equal(coef, sv.dual_coef_[0])
              and sol.rho[0] == sv.intercept_[0] and np.array_equal(dec, sv.decision_function(K[test][:, train])))
        if not ok:
            raise SystemExit("SVM MISMATCH n=%d kind=%d C=%g tol=%g ndup=%d seed=%d case=%d iters=%d" %
                             (n, kind, C, tol, ndup, a.seed, cases, int(sol.iters[0])))
        cases += 1
        shrunk += int(a.shrinking and int(sv.n_iter_[0]) > min(len(train), 1000))
        if cases % 50 == 0:
            print("%d cases ok" % cases, flush=True)
    print("svm fuzz ok: %d cases (%d more stopped at the iteration cap and were skipped)%s" % (
        cases, capped, "; %d ran long enough to shrink" % shrunk if a.shrinking else ""))


if __name__ == "__main__":
    main()
