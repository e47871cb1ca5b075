/*
 * gkm_oracle.c -- CPU restatement (plain C) of gkmQC's gkm kernel-matrix path.
 *
 * TEST INFRASTRUCTURE ONLY (see gkm_oracle.h).  Parity status: PINNED against the
 * compiled reference through tests/golden/ (generator: tests/golden/make_golden.py).
 *
 * Design: no k-mer tree.  Each sequence becomes arrays of 2-bit packed l-mers
 * (forward for the row side; forward + reverse complement for the column side)
 * and the mismatch profile is a brute-force all-pairs Hamming histogram.  This is
 * algebraically what the reference's tree DFS accumulates.
 */
#include "gkm_oracle.h"

#include <ctype.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* parameter check: same conditions and order as gkmkern_pylib.c:38-64        */
const char *gkmo_check_params(int kernel_type, int L, int k, int d)
{
    if (kernel_type < 0 || kernel_type > 5) return "unknown kernel type";
    if (L < 2) return "L < 2";
    if (L > 12) return "L > 12";
    if (k > L) return "k > L";
    if (d > L - k) return "d > L - k";
    return NULL;
}

/* ------------------------------------------------------------------------- */
/* binomial coefficient as a double, with the reference's extension to n<0
 * (libgkm.c:73-105: C(n,r) = (-1)^r C(r-n-1, r) for n<0).                    */
static double binom(int n, int r)
{
    double row[64];
    int i, j;
    if (r < 0) return 0.0;
    if (n < 0) {
        double v = binom(r - n - 1, r);
        return (r % 2 == 0) ? v : -v;
    }
    if (n < r) return 0.0;
    if (r >= 64) return 0.0; /* never reached for L<=12 */
    for (j = 0; j <= r; j++) row[j] = 0.0;
    row[0] = 1.0;
    for (i = 1; i <= n; i++)
        for (j = r; j >= 1; j--) row[j] += row[j - 1]; /* Pascal's rule, exact in fp64 */
    return row[r];
}

/* c_m for kernel types 1..5 (libgkm.c:107-202).  The order of floating-point
 * operations follows the reference so the doubles come out bit-identical.     */
static void lmer_estimate_weights(int L, int K, int truncated, double *res)
{
    double cur[GKMO_MAX_L + 1][GKMO_MAX_L + 1], prev[GKMO_MAX_L + 1][GKMO_MAX_L + 1];
    double (*pc)[GKMO_MAX_L + 1] = cur, (*pp)[GKMO_MAX_L + 1] = prev, (*tmp)[GKMO_MAX_L + 1];
    double wm[GKMO_MAX_L + 1], h[GKMO_MAX_L + 1], hT[GKMO_MAX_L + 1];
    int i, j, iL, iK, jM, m;

    for (i = 0; i <= K; i++)
        for (j = 0; j <= K; j++) pc[i][j] = pp[i][j] = 1.0;

    /* libgkm.c:133-143 -- note the jM>=1 entries read the buffer being written */
    for (iL = 1; iL <= L; iL++) {
        for (iK = 1; iK <= K; iK++) {
            pc[iK][0] = pp[iK][0] + 3 * pp[iK - 1][0];
            for (jM = 1; jM <= iK; jM++) pc[iK][jM] = (pc[iK - 1][jM - 1] * (iK - iL)) / iK;
        }
        tmp = pp; pp = pc; pc = tmp;
    }

    {
        double nnorm = binom(L, K) * pow(4, 1.0 * L); /* libgkm.c:145 */
        for (i = 0; i <= K; i++) wm[i] = pp[K][i] / nnorm;
    }

    for (m = 0; m <= L; m++) { /* libgkm.c:152-158 */
        int ub = (m < K) ? m : K;
        h[m] = 0;
        for (i = 0; i <= ub; i++) h[m] += wm[i] * binom(L - m, K - i) * binom(m, i);
    }
    {
        int keep = 1; /* libgkm.c:160-168 */
        for (i = 0; i <= L; i++) {
            if (h[i] < 1e-50) keep = 0;
            hT[i] = keep ? h[i] : 0.0;
        }
    }
    for (m = 0; m <= L; m++) { /* libgkm.c:171-191 */
        int m1, m2, t;
        double w = 0;
        for (m1 = 0; m1 <= L; m1++)
            for (m2 = 0; m2 <= L; m2++)
                for (t = 0; t <= L; t++) {
                    int r = m1 + m2 - 2 * t - L + m;
                    if (t <= m && (m1 - t) <= (L - m) && r <= (m1 - t) && r >= 0) {
                        double cc = binom(m, t) * binom(L - m, m1 - t) * binom(m1 - t, r) *
                                    pow(3, 1.0 * t) * pow(2, 1.0 * r);
                        if (truncated) w += cc * hT[m1] * hT[m2];
                        else w += cc * h[m1] * h[m2];
                    }
                }
        res[L - m] = w;
    }
}

int gkmo_mismatch_weights(int kernel_type, int L, int k, double *out)
{
    int i;
    if (L < 1 || L > GKMO_MAX_L || k < 0 || k > L) return 1;
    for (i = 0; i <= L; i++) out[i] = 0.0;
    if (kernel_type == 0) { /* libgkm.c:204-217 */
        for (i = 0; i <= L; i++)
            if (L - i >= k) out[i] = binom(L - i, k);
    } else {
        /* type 1 = full filter, everything else truncated (libgkm.c:997-1019) */
        lmer_estimate_weights(L, k, kernel_type != 1, out);
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
void gkmo_position_weights(int kernel_type, int n, uint8_t M, double H, uint8_t *wt)
{
    int i, center = n / 2;
    if (kernel_type == 4 || kernel_type == 5) { /* libgkm.c:914-925 */
        for (i = 0; i < n; i++) {
            double v = floor(M * exp((-1) * log(2) * abs(center - i) / H) + 1);
            /* the reference casts the double straight to u_int8_t; going through int
             * gives the x86-64 behaviour (256 -> 0) without undefined behaviour    */
            uint8_t w = (uint8_t)(int)v;
            if (w > M) w = M;
            wt[i] = w;
        }
    } else {
        for (i = 0; i < n; i++) wt[i] = 1;
    }
}

/* ------------------------------------------------------------------------- */
/* FASTA (libgkm.c:1207-1332)                                                 */
typedef struct {
    int n, cap;
    int *len;
    uint8_t **seq;
} seqlist;

static int seqlist_push(seqlist *s, const char *txt, int len, long *n_invalid)
{
    int i;
    uint8_t *codes;
    if (s->n == s->cap) {
        s->cap = s->cap ? 2 * s->cap : 256;
        s->len = (int *)realloc(s->len, sizeof(int) * (size_t)s->cap);
        s->seq = (uint8_t **)realloc(s->seq, sizeof(uint8_t *) * (size_t)s->cap);
    }
    codes = (uint8_t *)malloc((size_t)(len > 0 ? len : 1));
    for (i = 0; i < len; i++) { /* libgkm.c:864-875 */
        switch (toupper((unsigned char)txt[i])) {
        case 'A': codes[i] = 0; break;
        case 'C': codes[i] = 1; break;
        case 'G': codes[i] = 2; break;
        case 'T': codes[i] = 3; break;
        default: codes[i] = 0; (*n_invalid)++; break;
        }
    }
    s->len[s->n] = len;
    s->seq[s->n] = codes;
    s->n++;
    return 0;
}

static int read_fasta(const char *path, seqlist *out, long *n_invalid, long *n_trunc)
{
    FILE *fp = fopen(path, "rb");
    char *buf;
    long size, pos;
    char cur[GKMO_MAX_SEQ + 1];
    int curlen = 0, open = 0;
    if (!fp) return 1;
    fseek(fp, 0, SEEK_END);
    size = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    buf = (char *)malloc((size_t)size + 1);
    if (size > 0 && fread(buf, 1, (size_t)size, fp) != (size_t)size) { fclose(fp); free(buf); return 1; }
    fclose(fp);
    buf[size] = '\0';

    pos = 0;
    while (pos < size) {
        long eol = pos, end;
        int linelen;
        while (eol < size && buf[eol] != '\n') eol++;
        end = pos; /* logical line stops at the first CR or LF (libgkm.c:1222) */
        while (end < eol && buf[end] != '\r') end++;
        linelen = (int)(end - pos);
        if (linelen > 0 && buf[pos] == '>') {
            if (open) seqlist_push(out, cur, curlen, n_invalid);
            open = 1;
            curlen = 0;
        } else if (open && curlen < GKMO_MAX_SEQ) { /* libgkm.c:1294-1302 */
            int take = linelen;
            if (curlen + linelen >= GKMO_MAX_SEQ + 1) { take = GKMO_MAX_SEQ - curlen; (*n_trunc)++; }
            memcpy(cur + curlen, buf + pos, (size_t)take);
            curlen += take;
        }
        pos = eol + 1;
    }
    if (open) seqlist_push(out, cur, curlen, n_invalid);
    free(buf);
    return 0;
}

int gkmo_read_problem(const char *posfile, const char *negfile, gkmo_problem *out)
{
    seqlist s = {0, 0, NULL, NULL};
    int npos;
    memset(out, 0, sizeof(*out));
    if (read_fasta(posfile, &s, &out->n_invalid, &out->n_truncated)) return 1;
    npos = s.n;
    if (read_fasta(negfile, &s, &out->n_invalid, &out->n_truncated)) return 1;
    out->n = s.n;
    out->n_pos = npos;
    out->len = s.len;
    out->seq = s.seq;
    if (npos == 0 || s.n == npos) return 2; /* empty file: undefined in the reference */
    return 0;
}

void gkmo_free_problem(gkmo_problem *p)
{
    int i;
    for (i = 0; i < p->n; i++) free(p->seq[i]);
    free(p->seq);
    free(p->len);
    memset(p, 0, sizeof(*p));
}

/* ------------------------------------------------------------------------- */
/* l-mers as 2 bits per base                                                  */
static void pack_lmers(const uint8_t *s, int len, int L, uint32_t *out)
{
    uint32_t mask = (L == 16) ? 0xffffffffu : ((1u << (2 * L)) - 1u), v = 0;
    int i;
    for (i = 0; i < len; i++) {
        v = ((v << 2) | s[i]) & mask;
        if (i >= L - 1) out[i - L + 1] = v;
    }
}

static inline int lmer_mismatches(uint32_t x, uint32_t y)
{
    uint32_t t = x ^ y;
    t = (t | (t >> 1)) & 0x55555555u;
    return __builtin_popcount(t);
}

typedef struct {
    int n;          /* number of forward l-mers */
    uint32_t *fwd;  /* [n] */
    uint32_t *rc;   /* [n] l-mers of the reverse-complement strand */
    uint8_t *wt;    /* [n] */
    uint8_t *wt_rc; /* [n]  wt_rc[n-1-i] = wt[i]  (libgkm.c:924) */
} lmer_set;

static int build_lmer_set(const gkmo_opt *o, const uint8_t *s, int len, lmer_set *ls)
{
    int n = len - o->L + 1, i;
    uint8_t *rcs;
    if (n <= 0) return 1;
    ls->n = n;
    ls->fwd = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
    ls->rc = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
    ls->wt = (uint8_t *)malloc((size_t)n);
    ls->wt_rc = (uint8_t *)malloc((size_t)n);
    rcs = (uint8_t *)malloc((size_t)len);
    for (i = 0; i < len; i++) rcs[i] = (uint8_t)(3 - s[len - 1 - i]); /* libgkm.c:877-888 */
    pack_lmers(s, len, o->L, ls->fwd);
    pack_lmers(rcs, len, o->L, ls->rc);
    free(rcs);
    gkmo_position_weights(o->kernel_type, n, o->M, o->H, ls->wt);
    for (i = 0; i < n; i++) ls->wt_rc[n - 1 - i] = ls->wt[i];
    return 0;
}

static void free_lmer_set(lmer_set *ls)
{
    free(ls->fwd); free(ls->rc); free(ls->wt); free(ls->wt_rc);
}

/* P_m(a,b): forward l-mers of a against forward+rc l-mers of b.  Unsigned
 * accumulation = two's-complement wrap, like the reference's int on overflow. */
static void profile_sets(const lmer_set *a, const lmer_set *b, int d, int32_t *prof)
{
    uint32_t acc[GKMO_MAX_L + 1];
    int p, q, m;
    for (m = 0; m <= d; m++) acc[m] = 0;
    for (p = 0; p < a->n; p++) {
        uint32_t u = a->fwd[p], w = a->wt[p];
        for (q = 0; q < b->n; q++) {
            m = lmer_mismatches(u, b->fwd[q]);
            if (m <= d) acc[m] += w * b->wt[q];
            m = lmer_mismatches(u, b->rc[q]);
            if (m <= d) acc[m] += w * b->wt_rc[q];
        }
    }
    for (m = 0; m <= d; m++) prof[m] = (int32_t)acc[m];
}

void gkmo_profile(const gkmo_opt *o, const uint8_t *a, int la, const uint8_t *b, int lb, int32_t *prof)
{
    lmer_set A, B;
    int m;
    for (m = 0; m <= o->d; m++) prof[m] = 0;
    if (build_lmer_set(o, a, la, &A)) return;
    if (build_lmer_set(o, b, lb, &B)) { free_lmer_set(&A); return; }
    profile_sets(&A, &B, o->d, prof);
    free_lmer_set(&A);
    free_lmer_set(&B);
}

/* Σ_m c_m P_m in ascending m from 0.0 (libgkm.c:576-582, 753-756) */
static double weighted_sum(const double *c, const int32_t *prof, int d)
{
    double sum = 0;
    int m;
    for (m = 0; m <= d; m++) sum += (c[m] * prof[m]);
    return sum;
}

typedef struct {
    const gkmo_opt *o;
    const lmer_set *sets;
    const double *c;
    const double *sqnorm;
    int n, tid, nthreads;
    int32_t *P;
    double *K;
    double **rows; /* alternative output: row pointers (pywrapper) */
} gram_task;

static void *gram_worker(void *arg)
{
    gram_task *t = (gram_task *)arg;
    int d = t->o->d, a, j, m;
    int rbf = (t->o->kernel_type == 3 || t->o->kernel_type == 5);
    int32_t prof[GKMO_MAX_L + 1];
    for (a = t->tid; a < t->n; a += t->nthreads) { /* gkmkern_pylib.c:81-83 row interleave */
        for (j = 0; j < a; j++) {
            double v;
            profile_sets(&t->sets[a], &t->sets[j], d, prof);
            if (t->P) for (m = 0; m <= d; m++) t->P[((size_t)a * t->n + j) * (d + 1) + m] = prof[m];
            v = weighted_sum(t->c, prof, d);
            v /= (t->sqnorm[a] * t->sqnorm[j]);          /* libgkm.c:1169-1172 */
            if (rbf) v = exp(t->o->gamma * (v - 1));     /* libgkm.c:1175-1179 */
            if (t->K) t->K[(size_t)a * t->n + j] = v;
            if (t->rows) t->rows[a][j] = v;
        }
    }
    return NULL;
}

static int gram_impl(const gkmo_opt *o, const gkmo_problem *p, double *sqnorm_out, int32_t *P,
                     double *K, double **rows, int nthreads)
{
    double c[GKMO_MAX_L + 1];
    lmer_set *sets;
    double *sq;
    int i, m, d = o->d, rc = 0;
    pthread_t *th;
    gram_task *tasks;
    int32_t prof[GKMO_MAX_L + 1];

    if (gkmo_check_params(o->kernel_type, o->L, o->k, o->d)) return 1;
    if (nthreads < 1) nthreads = 1;
    gkmo_mismatch_weights(o->kernel_type, o->L, o->k, c);
    sets = (lmer_set *)calloc((size_t)p->n, sizeof(lmer_set));
    sq = (double *)malloc(sizeof(double) * (size_t)p->n);
    for (i = 0; i < p->n; i++) {
        if (build_lmer_set(o, p->seq[i], p->len[i], &sets[i])) { rc = 3; break; }
    }
    if (rc) { /* a sequence shorter than L: undefined in the reference, an error here */
        for (m = 0; m < i; m++) free_lmer_set(&sets[m]);
        free(sets); free(sq);
        return rc;
    }

    for (i = 0; i < p->n; i++) { /* self norm, libgkm.c:723-759 */
        profile_sets(&sets[i], &sets[i], d, prof);
        if (P) for (m = 0; m <= d; m++) P[((size_t)i * p->n + i) * (d + 1) + m] = prof[m];
        sq[i] = sqrt(weighted_sum(c, prof, d));
        if (sqnorm_out) sqnorm_out[i] = sq[i];
    }

    th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    tasks = (gram_task *)malloc(sizeof(gram_task) * (size_t)nthreads);
    for (i = 0; i < nthreads; i++) {
        gram_task t = {o, sets, c, sq, p->n, i, nthreads, P, K, rows};
        tasks[i] = t;
        if (i > 0) pthread_create(&th[i], NULL, gram_worker, &tasks[i]);
    }
    gram_worker(&tasks[0]);
    for (i = 1; i < nthreads; i++) pthread_join(th[i], NULL);

    for (i = 0; i < p->n; i++) { /* gkmkern_pylib.c:218-221 */
        if (K) K[(size_t)i * p->n + i] = 1.0;
        if (rows) rows[i][i] = 1.0;
    }
    for (i = 0; i < p->n; i++) free_lmer_set(&sets[i]);
    free(sets); free(sq); free(th); free(tasks);
    return 0;
}

int gkmo_gram(const gkmo_opt *o, const gkmo_problem *p, double *sqnorm, int32_t *P, double *K,
              int nthreads)
{
    return gram_impl(o, p, sqnorm, P, K, NULL, nthreads);
}

int gkmo_main_pywrapper(gkmo_opt *opts, double **kmat, int *kmat_size)
{
    gkmo_problem p;
    int rc;
    if (gkmo_check_params(opts->kernel_type, opts->L, opts->k, opts->d)) return 1;
    if (gkmo_read_problem(opts->posfile, opts->negfile, &p)) return 1;
    rc = gram_impl(opts, &p, NULL, NULL, NULL, kmat, opts->nthreads);
    if (!rc) { kmat_size[0] = p.n_pos; kmat_size[1] = p.n - p.n_pos; }
    gkmo_free_problem(&p);
    return rc;
}
