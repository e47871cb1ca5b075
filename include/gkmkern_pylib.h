/*
 * gkmkern_pylib.h -- the drop-in boundary of the gkm kernel-matrix path.
 *
 * The shared object built from this repository is named `gkmkern_pylib.so` and
 * exports exactly the entry point the reference's Python caller binds with ctypes
 * (reference scripts/gkmsvm.py:85-88):
 *
 *     int gkm_main_pywrapper(gkmOpt *opts, double **kmat, int *kmat_size);
 *
 * Replaces: reference src/libgkm.h:149-164 (struct + prototype) and the
 * implementation in src/gkmkern_pylib.c:92-246.  Same struct layout (x86-64:
 * offsets 0,4,8,12,16,24,32,40,48,56,60; sizeof 64), same argument meaning:
 *
 *   opts        kernel_type 0..5 (GKM, EST_FULL, EST_TRUNC, EST_TRUNC_RBF,
 *               EST_TRUNC_PW, EST_TRUNC_PW_RBF -- libgkm.h:51), L, k, d, M, H,
 *               gamma, NUL-terminated FASTA paths, nthreads (host helper threads;
 *               the row work runs on the GPU), verbosity 0..4.
 *   kmat        caller-owned array of row pointers; row a receives K(a,j) for
 *               j<a and kmat[a][a]=1.0; nothing else is written
 *               (gkmkern_pylib.c:83,218-221).
 *   kmat_size   caller-owned int[2]: {n_pos, n_neg} (gkmkern_pylib.c:223-224).
 *   returns     0 on success, non-zero on any error.  Unlike the reference this
 *               library never calls exit(): unreadable files, bad verbosity, empty
 *               FASTA, sequences shorter than L, missing GPU and HIP errors all
 *               return non-zero after an ERROR log line.
 *   on failure  Everything that is checked or set up before the computation -- the
 *               parameters, both files, the device list (GKM_DEVICE / GKM_DEVICES),
 *               every device's context, upload and n x n allocation, on ALL devices
 *               when several are used -- fails BEFORE a single cell of kmat or
 *               kmat_size has been written (the reference returns from its own
 *               checks before writing anything, gkmkern_pylib.c:157-161).  Only a HIP
 *               error during the computation itself (a failed launch or copy) can
 *               leave rows partly written; the return value is non-zero then too and
 *               the caller must discard the matrix (scripts/gkmsvm.py:90-92 does).
 *   ownership   The callee frees all HOST memory it allocated before it returns and keeps
 *               no pointer into the caller's memory (gkmkern_pylib.c:226-243).  DEVICE
 *               memory of a single-GPU call -- the context with its scratch and the
 *               n x n matrix -- is kept for the next call of the same parameters and
 *               size (bin/gkmqc.py:341-343 makes ~20 per run; allocating and freeing it
 *               cost every call ~25 device-wide synchronisations), until
 *               gkm_release_device_cache(), a call with other parameters, a failed call,
 *               or the end of the process.  GKM_KEEP_DEVICE=0 frees it on return.
 */
#ifndef GKMKERN_PYLIB_H
#define GKMKERN_PYLIB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { GKM = 0, EST_FULL, EST_TRUNC, EST_TRUNC_RBF, EST_TRUNC_PW, EST_TRUNC_PW_RBF };

typedef struct _gkmOpt {
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
} gkmOpt;

int gkm_main_pywrapper(gkmOpt *opts, double **kmat, int *kmat_size);

/* Frees the device memory gkm_main_pywrapper keeps between calls (see "ownership" above); optional.
 * No counterpart in the reference, which has no device. */
void gkm_release_device_cache(void);
/* calls of gkm_main_pywrapper so far that found their context and matrix already on the device */
long gkm_device_cache_hits(void);

/*
 * Host-side pieces of the same path, exported so that callers and tests can use
 * them without a GPU.  They replace the file-static routines named on each line.
 */

/* NULL if (kernel_type, L, k, d) is acceptable, else the reference's message.
 * Replaces gkm_check_parameter, src/gkmkern_pylib.c:38-64. */
const char *gkm_check_parameter_values(int kernel_type, int L, int k, int d);

/* Mismatch weights c_0..c_L into out[L+1]; 0 on success.
 * Replaces calc_gkm_kernel_wt / calc_gkm_kernel_lmerest_wt, src/libgkm.c:107-217,
 * as selected by gkmkernel_init, src/libgkm.c:997-1019. */
int gkm_mismatch_weights(int kernel_type, int L, int k, double *out);

/* Positional weights of the n forward l-mers of one sequence into wt[n].
 * Replaces the weight loop of gkmkernel_new_object, src/libgkm.c:910-932. */
void gkm_position_weights(int kernel_type, int n, uint8_t M, double H, uint8_t *wt);

/* FASTA reader with the reference's record rules (src/libgkm.c:1207-1332):
 * returns a handle (NULL on error) holding all sequences, positives first. */
typedef struct gkm_problem gkm_problem;
gkm_problem *gkm_problem_read(const char *posfile, const char *negfile);
void gkm_problem_free(gkm_problem *p);
int gkm_problem_size(const gkm_problem *p);     /* n_pos + n_neg */
int gkm_problem_npos(const gkm_problem *p);
int gkm_problem_seqlen(const gkm_problem *p, int i);
const uint8_t *gkm_problem_codes(const gkm_problem *p, int i); /* 0..3 = A,C,G,T */
const int64_t *gkm_problem_offsets(const gkm_problem *p);      /* [n+1] offsets into the concatenated codes */
const uint8_t *gkm_problem_all_codes(const gkm_problem *p);    /* all sequences back to back (gkmhip_set_sequences) */
long gkm_problem_invalid_chars(const gkm_problem *p);          /* mapped to 'A' */
long gkm_problem_truncated(const gkm_problem *p);              /* cut at 2047 nt */

#ifdef __cplusplus
}
#endif
#endif
