/*
 * gkm_svm.h -- GPU-resident consumer of the gkm kernel matrix: C-SVC
 * training and decision values on a PRECOMPUTED kernel that already sits in HBM, for the
 * cross-validation gkmQC runs on every matrix (reference scripts/gkmsvm.py:104-176:
 * StratifiedKFold x repeats -> sklearn.svm.SVC(kernel="precomputed") -> decision_function).
 *
 * What it replaces: the LIBSVM solver inside scikit-learn (sklearn/svm/src/libsvm/svm.cpp,
 * `Solver::Solve` with second-order working-set selection; gkmsvm_train_batch without the shrinking
 * heuristic -- gkmQC's default `--shrinking 0` --, gkmsvm_train_batch_general with or without it).  That code is a third-party dependency of the reference, not part of
 * /root/reference; the algorithm restated here is Fan, Chen, Lin (2005) "Working set selection
 * using second order information" as implemented by LIBSVM 3.x: same iteration sequence, same
 * tie breaking, kernel values rounded to float as LIBSVM's Qfloat, all other arithmetic fp64.
 * Parity is pinned against scikit-learn itself (tests/test_svm_gpu.py): dual coefficients and
 * intercept bit-identical, AUC identical.
 *
 * All folds of a cross-validation are solved concurrently, one workgroup per fold.
 * Plain pointers and sizes; `K` and every array below are DEVICE pointers unless noted.
 */
#ifndef GKM_SVM_H
#define GKM_SVM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Train nprob independent C-SVC problems on sub-matrices of one symmetric kernel matrix.
 *   K, ld, n     n x n fp64, row-major with leading dimension ld, BOTH triangles filled, device
 *   idx          concatenated training indices (into K) of all problems, each problem in LIBSVM's
 *                internal order: all samples of class 0 first (they get y = +1), then class 1 (y = -1)
 *   off[nprob+1] HOST array: offsets of the problems into idx / alpha / grad
 *   n0[nprob]    HOST array: number of class-0 samples of each problem
 *   C, eps       box constraint and stopping tolerance (sklearn `C`, `tol`)
 *   alpha, grad  out, device, same layout as idx: dual variables (0..C) and final gradient
 *   rho          out, device, nprob doubles
 *   iters        out, device, nprob ints; NEGATIVE when the fold stopped at the iteration cap (10^7, LIBSVM's
 *                classic default; scikit-learn has none): that fold has not converged and must be re-solved
 *                by the caller if the reference's result is wanted
 * Returns 0 on success (work enqueued on `stream`); GKMSVM_RC_SHAPE_REFUSED when the device refuses the launch shape
 * the largest fold needs (its dynamic LDS) -- the one failure after which gkmsvm_train_batch_general is worth trying:
 * it needs no state in LDS (it keeps some there when the device grants it) and returns the same bits without
 * shrinking; any other value is an error of the arguments or of the device.
 */
#define GKMSVM_RC_SHAPE_REFUSED 5
int gkmsvm_train_batch(int device, const double *K, int64_t ld, int n, int nprob, const int *idx, const int64_t *off,
                       const int *n0, double C, double eps, double *alpha, double *grad, double *rho, int *iters,
                       void *stream);

/*
 * The same with LIBSVM's shrinking heuristic (`shrinking` != 0; scikit-learn `SVC(shrinking=True)`, what the
 * reference's `--shrinking 1` selects, scripts/gkmsvm.py:110-118) and for folds of up to 60 000 samples:
 * Solver::Solve restated with do_shrinking / swap_index / reconstruct_gradient / G_bar, one workgroup per fold,
 * state in global memory -- for folds of at most 8 192 samples the part of it every iteration scans (gradient, Q_i,
 * matrix index, one state byte per sample) in LDS instead, 147 KB, if the device grants that.  Bit-identical to
 * scikit-learn with either setting (tests/test_svm_gpu.py); 1.4x (8 000-sample folds) to 2x slower per iteration than
 * gkmsvm_train_batch, which stays the path for gkmQC's default.
 * Environment, for measurements: GKM_SVM_GEN_LDS=0 (state in global memory always), GKM_SVM_GEN_T=512|1024 (threads).
 */
int gkmsvm_train_batch_general(int device, const double *K, int64_t ld, int n, int nprob, const int *idx,
                               const int64_t *off, const int *n0, double C, double eps, int shrinking, double *alpha,
                               double *grad, double *rho, int *iters, void *stream);

/*
 * Decision values of problem p for its test samples, in LIBSVM's summation order:
 *   dec[t] = sum_{k in training order, alpha_k > 0} alpha_k y_k K(test_t, train_k) - rho
 * (scikit-learn's decision_function returns the negative of this for a two-class problem).
 *   test_idx / test_off: concatenated test indices per problem (test_off on the HOST).
 */
int gkmsvm_decision_batch(int device, const double *K, int64_t ld, int nprob, const int *idx, const int64_t *off,
                          const int *n0, const double *alpha, const double *rho, const int *test_idx,
                          const int64_t *test_off, double *dec, void *stream);

const char *gkmsvm_last_error(void);

/* The calls above keep their device scratch (problem descriptors, the matrix diagonal, the general solver's state) in a
 * per-process pool instead of allocating and freeing it per call: hipMalloc / hipFree wait for the whole device, i.e. for
 * the Gram kernel of the next subset that a pipeline runs beside the solver (gkmqc_amd/gkmsvm.py init_many).  This frees
 * what is not in use (optional). */
void gkmsvm_release_cache(void);

#ifdef __cplusplus
}
#endif
#endif
