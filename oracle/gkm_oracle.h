/*
 * gkm_oracle.h -- CPU restatement of gkmQC's gkm kernel-matrix path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker.  The product (gkmqc_amd/) never links or calls it.
 *
 * Parity status: PINNED.  Every function here is checked in tests/ against
 * outputs of the reference itself (oracle/_ref, built from /root/reference/src by
 * oracle/Makefile) through the committed fixtures under tests/golden/.
 *
 * The algorithm restated is the one behind
 *   int gkm_main_pywrapper(gkmOpt*, double **kmat, int *kmat_size)
 * (reference src/gkmkern_pylib.c:92-246, src/libgkm.c).  It is written as a
 * brute-force all-pairs l-mer comparison (no k-mer tree): the mathematics is
 * SURVEY.md Appendix A.
 */
#ifndef GKM_ORACLE_H
#define GKM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GKMO_MAX_L 12
#define GKMO_MAX_SEQ 2047 /* reference MAX_SEQ_LENGTH-1, libgkm.h:32 */

/* same field order/layout as the reference gkmOpt (libgkm.h:149-161) */
typedef struct {
    int kernel_type;
    int L;
    int k;
    int d;
    uint8_t M;
    double H;
    double gamma;
    char *posfile;
    char *negfile;
    int nthreads;
    int verbosity;
} gkmo_opt;

/* parsed problem: sequences as base codes 0..3 (A,C,G,T) */
typedef struct {
    int n;          /* number of sequences (pos then neg) */
    int n_pos;
    int *len;       /* [n] */
    uint8_t **seq;  /* [n][len] codes 0..3 */
    long n_invalid; /* characters mapped to 'A' (libgkm.c:870-873) */
    long n_truncated;
} gkmo_problem;

/* returns NULL if ok, else the reference's error text (gkmkern_pylib.c:38-64) */
const char *gkmo_check_params(int kernel_type, int L, int k, int d);

/* mismatch weights c_0..c_L (libgkm.c:107-217, 997-1019). out has L+1 doubles */
int gkmo_mismatch_weights(int kernel_type, int L, int k, double *out);

/* positional weights of the n forward l-mers (libgkm.c:910-932) */
void gkmo_position_weights(int kernel_type, int n, uint8_t M, double H, uint8_t *wt);

/* FASTA reader with the reference's rules (libgkm.c:1207-1332). 0 = ok */
int gkmo_read_problem(const char *posfile, const char *negfile, gkmo_problem *out);
void gkmo_free_problem(gkmo_problem *p);

/* integer mismatch profile P_m(a,b), m=0..d (libgkm.c:315-387,553-575 semantics;
 * for a==b this is the self profile of libgkm.c:723-751). prof has d+1 int32 */
void gkmo_profile(const gkmo_opt *o, const uint8_t *a, int la, const uint8_t *b, int lb,
                  int32_t *prof);

/* Everything for a problem: sqnorm[n]; lower-triangle profiles P[(a*n+j)*(d+1)+m]
 * for j<=a (may be NULL); K[a*n+j] for j<a and K[a*n+a]=1.0 (may be NULL).
 * nthreads >= 1 row threads. */
int gkmo_gram(const gkmo_opt *o, const gkmo_problem *p, double *sqnorm, int32_t *P, double *K,
              int nthreads);

/* drop-in restatement of gkm_main_pywrapper (same argument meaning) */
int gkmo_main_pywrapper(gkmo_opt *opts, double **kmat, int *kmat_size);

#ifdef __cplusplus
}
#endif
#endif
