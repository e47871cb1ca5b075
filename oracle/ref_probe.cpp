/*
 * ref_probe.cpp -- harness around the UNMODIFIED reference sources, built only in
 * the development container (where /root/reference exists) into oracle/_ref/.
 *
 * TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it includes
 * the reference translation unit where it lies so that the file-static routines
 * (weights, self-norm, k-mer tree DFS) can be driven directly and their integer
 * mismatch profiles dumped for tests/golden/ (recipe: SURVEY.md App. C #4).
 */
#define CLOG_MAIN
#include "/root/reference/src/libgkm.c"

extern "C" {

/* c_m as computed by gkmkernel_init (libgkm.c:978-1033) */
int refp_weights(int kernel_type, int L, int k, int d, double *out)
{
    gkm_parameter param;
    param.kernel_type = kernel_type; param.L = L; param.k = k; param.d = d;
    param.M = 50; param.H = 50; param.gamma = 1.0; param.nthreads = 1;
    clog_init_fd(LOGGER_ID, 1);
    clog_set_level(LOGGER_ID, CLOG_ERROR);
    gkm_kernel *kernel = gkmkernel_init(&param);
    for (int i = 0; i <= d; i++) out[i] = kernel->weights[i];
    kernel->prob_svm_data = NULL; kernel->prob_gkmkernel_index = NULL;
    kernel->prob_libsvm_index = NULL; kernel->prob_kmertree = NULL;
    gkmkernel_destroy(kernel);
    clog_free(LOGGER_ID);
    return 0;
}

/* Reads both FASTA files with the reference reader and returns, for every
 * (a, j) with j <= a, the int mismatch profile the reference's DFS produces
 * (last_seqid = N so the diagonal is included), plus sqnorm, lengths and the
 * forward positional weights.  P is [N][N][d+1] (upper triangle untouched),
 * wt is [N][wt_stride].  Returns N, or <0 on error. */
int refp_profiles(int kernel_type, int L, int k, int d, int M, double H, double gamma,
                  const char *posfile, const char *negfile, int maxn,
                  int *P, double *sqnorm, int *lens, unsigned char *wt, int wt_stride, int *n_pos)
{
    gkm_parameter param;
    svm_problem prob;
    param.kernel_type = kernel_type; param.L = L; param.k = k; param.d = d;
    param.M = (u_int8_t)M; param.H = H; param.gamma = gamma; param.nthreads = 1;
    clog_init_fd(LOGGER_ID, 1);
    clog_set_level(LOGGER_ID, CLOG_ERROR);
    gkm_kernel *kernel = gkmkernel_init(&param);
    int npos = gkmkernel_read_problems(kernel, &prob, posfile, negfile);
    int N = prob.l;
    if (N > maxn) return -1;
    gkmkernel_build_tree(kernel, prob.x, prob.l);
    *n_pos = npos;

    int **mm = (int **)malloc(sizeof(int *) * (size_t)(d + 1));
    for (int m = 0; m <= d; m++) mm[m] = (int *)malloc(sizeof(int) * (size_t)N);
    static BaseMismatchCount mb[MAX_SEQ_LENGTH];
    for (int a = 0; a < N; a++) {
        const gkm_data *da = prob.x[a];
        int n = da->seqlen - L + 1;
        lens[a] = da->seqlen;
        sqnorm[a] = da->sqnorm;
        for (int i = 0; i < n && i < wt_stride; i++) wt[(size_t)a * wt_stride + i] = da->wt[i];
        for (int i = 0; i < n; i++) { mb[i].bid = da->seq + i; mb[i].wt = da->wt[i]; mb[i].mmcnt = 0; }
        for (int m = 0; m <= d; m++) for (int j = 0; j < N; j++) mm[m][j] = 0;
        kmertree_dfs(kernel->prob_kmertree, N, 0, 0, mb, n, mm);
        for (int j = 0; j <= a; j++)
            for (int m = 0; m <= d; m++) P[((size_t)a * N + j) * (d + 1) + m] = mm[m][j];
    }
    for (int m = 0; m <= d; m++) free(mm[m]);
    free(mm);
    for (int i = 0; i < N; i++) gkmkernel_delete_object(prob.x[i]);
    free(prob.y); free(prob.x);
    gkmkernel_destroy(kernel);
    clog_free(LOGGER_ID);
    return N;
}

/* The reference's batch-vs-set scoring entry, gkmkernel_kernelfunc_batch (libgkm.c:1115-1153; exported by its .so,
 * unused by gkm_main_pywrapper: SURVEY.md section 8 row a14 / f4), driven as a caller would: the problem is read and its
 * tree built, then row rows[i] is scored against ALL n sequences of the problem (db_array = the problem's own records;
 * the DFS it runs goes over the problem tree with last_seqid = n, libgkm.c:553-589).  out is [nrows][n]: normalised,
 * RBF applied for types 3 / 5, the diagonal entry as the reference computes it (G / sqnorm^2, not forced to 1.0).
 * Returns n, or < 0. */
int refp_batch_rows(int kernel_type, int L, int k, int d, int M, double H, double gamma,
                    const char *posfile, const char *negfile, const int *rows, int nrows, int maxn, double *out)
{
    gkm_parameter param;
    svm_problem prob;
    param.kernel_type = kernel_type; param.L = L; param.k = k; param.d = d;
    param.M = (u_int8_t)M; param.H = H; param.gamma = gamma; param.nthreads = 1;
    clog_init_fd(LOGGER_ID, 1);
    clog_set_level(LOGGER_ID, CLOG_ERROR);
    gkm_kernel *kernel = gkmkernel_init(&param);
    gkmkernel_read_problems(kernel, &prob, posfile, negfile);
    const int N = prob.l;
    if (N > maxn) return -1;
    gkmkernel_build_tree(kernel, prob.x, prob.l);
    for (int i = 0; i < nrows; i++) {
        if (rows[i] < 0 || rows[i] >= N) return -2;
        gkmkernel_kernelfunc_batch(kernel, rows[i], (const gkm_data **)prob.x, N, out + (size_t)i * N);
    }
    for (int i = 0; i < N; i++) gkmkernel_delete_object(prob.x[i]);
    free(prob.y); free(prob.x);
    gkmkernel_destroy(kernel);
    clog_free(LOGGER_ID);
    return N;
}

} /* extern "C" */
