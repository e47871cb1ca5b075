/*
 * gkm_host.c -- host-side (CPU, plain C) pieces of the gkm kernel-matrix path:
 * logger, parameter check, mismatch weights c_m, positional weights, FASTA reader.
 *
 * These are tiny tables and a ~3 MB text parse; they stay on the host because
 * floor(M*exp(..)+1) and the c_m recurrences must come from the host libm to be
 * bit-identical with the reference.  Everything heavy is in
 * gkm_context.hip ... gkm_copyout.hip (gkm_internal.h).
 */
#include "gkm_host.h"

#include <ctype.h>
#include <fcntl.h>
#include <math.h>
#include <stdarg.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

/* ------------------------------------------------------------------ logger */
/* One line per message on fd 1: "LEVEL YYYY-MM-DD HH:MM:SS: text" -- the format the
 * reference configures (libgkm.h:27, gkmkern_pylib.c:100-106) so pipeline logs keep
 * their shape.  verbosity 0..4 -> ERROR/WARN/INFO/DEBUG/TRACE (gkmkern_pylib.c:118-134). */
static int g_log_threshold = GKM_LOG_WARN;

void gkm_log_set_level(int level) { g_log_threshold = level; }
int gkm_log_enabled(int level) { return level >= g_log_threshold; }

int gkm_log_level_from_verbosity(int verbosity)
{
    switch (verbosity) {
    case 0: return GKM_LOG_ERROR;
    case 1: return GKM_LOG_WARN;
    case 2: return GKM_LOG_INFO;
    case 3: return GKM_LOG_DEBUG;
    case 4: return GKM_LOG_TRACE;
    default: return -1;
    }
}

void gkm_log(int level, const char *fmt, ...)
{
    static const char *names[] = {"TRACE", "DEBUG", "INFO", "WARN", "ERROR"};
    char line[1200];
    char stamp[40];
    time_t now;
    struct tm tmv;
    va_list ap;
    int n, m;
    if (level < g_log_threshold || level < 0 || level > GKM_LOG_ERROR) return;
    now = time(NULL);
    localtime_r(&now, &tmv);
    strftime(stamp, sizeof stamp, "%Y-%m-%d %H:%M:%S", &tmv);
    n = snprintf(line, sizeof line, "%s %s: ", names[level], stamp);
    va_start(ap, fmt);
    m = vsnprintf(line + n, sizeof line - (size_t)n - 2, fmt, ap);
    va_end(ap);
    if (m < 0) m = 0;
    n += (m < (int)sizeof line - n - 2) ? m : (int)sizeof line - n - 3;
    line[n++] = '\n';
    if (write(1, line, (size_t)n) < 0) { /* nothing sensible to do */ }
}

/* --------------------------------------------------------- parameter check */
const char *gkm_check_parameter_values(int kernel_type, int L, int k, int d)
{
    /* same tests in the same order as gkmkern_pylib.c:38-64 */
    if (kernel_type < GKM || kernel_type > EST_TRUNC_PW_RBF) return "unknown kernel type";
    if (L < 2) return "L < 2";
    if (L > 12) return "L > 12";
    if (k > L) return "k > L";
    if (d > (L - k)) return "d > L - k";
    return NULL;
}

/* -------------------------------------------------------- mismatch weights */
/* C(n,r) in fp64 by Pascal's rule; negative n reflected as the reference does
 * (libgkm.c:73-105).  All values involved are integers far below 2^53. */
static double choose(int n, int r)
{
    double tri[GKM_MAX_L * 2 + 4];
    if (r < 0) return 0.0;
    if (n < 0) return (r & 1) ? -choose(r - n - 1, r) : choose(r - n - 1, r);
    if (n < r) return 0.0;
    if (r > GKM_MAX_L * 2 + 2) return 0.0;
    memset(tri, 0, sizeof tri);
    tri[0] = 1.0;
    for (int row = 1; row <= n; row++)
        for (int col = (row < r ? row : r); col >= 1; col--) tri[col] += tri[col - 1];
    return tri[r];
}

/* Estimated-l-mer weights, kernel types 1..5 (libgkm.c:107-202).  Three stages:
 * (1) wm[i] from a two-buffer recurrence over iL=1..L, (2) filter h[m] and its
 * truncation at the first value below 1e-50, (3) the triple sum giving c_{L-m}.
 * Operation order is kept as in the reference so c_m is bit-identical. */
static void estimated_lmer_weights(int L, int K, int truncate, double *c)
{
    enum { S = GKM_MAX_L + 1 };
    double bufA[S][S], bufB[S][S];
    double (*fresh)[S] = bufA, (*old)[S] = bufB;
    double wm[S], h[S], hcut[S];

    for (int i = 0; i <= K; i++)
        for (int j = 0; j <= K; j++) fresh[i][j] = old[i][j] = 1.0;

    for (int iL = 1; iL <= L; iL++) {
        for (int iK = 1; iK <= K; iK++) {
            fresh[iK][0] = old[iK][0] + 3 * old[iK - 1][0];
            /* columns >= 1 chain through the buffer being filled (libgkm.c:138) */
            for (int jM = 1; jM <= iK; jM++) fresh[iK][jM] = (fresh[iK - 1][jM - 1] * (iK - iL)) / iK;
        }
        double (*t)[S] = old; old = fresh; fresh = t;
    }

    const double norm = choose(L, K) * pow(4, 1.0 * L);
    for (int i = 0; i <= K; i++) wm[i] = old[K][i] / norm;

    for (int m = 0; m <= L; m++) {
        const int top = m < K ? m : K;
        h[m] = 0;
        for (int i = 0; i <= top; i++) h[m] += wm[i] * choose(L - m, K - i) * choose(m, i);
    }
    int alive = 1;
    for (int i = 0; i <= L; i++) {
        if (h[i] < 1e-50) alive = 0;
        hcut[i] = alive ? h[i] : 0.0;
    }
    const double *f = truncate ? hcut : h;

    for (int m = 0; m <= L; m++) {
        double acc = 0;
        for (int m1 = 0; m1 <= L; m1++)
            for (int m2 = 0; m2 <= L; m2++)
                for (int t = 0; t <= L; t++) {
                    const int r = m1 + m2 - 2 * t - L + m;
                    if (t > m || (m1 - t) > (L - m) || r > (m1 - t) || r < 0) continue;
                    const double ways = choose(m, t) * choose(L - m, m1 - t) * choose(m1 - t, r) *
                                        pow(3, 1.0 * t) * pow(2, 1.0 * r);
                    acc += ways * f[m1] * f[m2];
                }
        c[L - m] = acc;
    }
}

int gkm_mismatch_weights(int kernel_type, int L, int k, double *out)
{
    if (L < 1 || L > GKM_MAX_L || k < 0 || k > L || kernel_type < 0 || kernel_type > 5) return 1;
    for (int m = 0; m <= L; m++) out[m] = 0.0;
    if (kernel_type == GKM) {
        for (int m = 0; m <= L; m++) /* libgkm.c:204-217 */
            if (L - m >= k) out[m] = choose(L - m, k);
    } else {
        estimated_lmer_weights(L, k, kernel_type != EST_FULL, out); /* libgkm.c:997-1019 */
    }
    return 0;
}

/* ------------------------------------------------------ positional weights */
void gkm_position_weights(int kernel_type, int n, uint8_t M, double H, uint8_t *wt)
{
    if (kernel_type != EST_TRUNC_PW && kernel_type != EST_TRUNC_PW_RBF) {
        memset(wt, 1, (size_t)(n > 0 ? n : 0)); /* libgkm.c:926-932 */
        return;
    }
    const int center = n / 2; /* libgkm.c:912 */
    for (int i = 0; i < n; i++) {
        /* libgkm.c:921: (u_int8_t) floor(M*exp(-ln2*|center-i|/H) + 1), then min(.,M).
         * A value of 256 (M = 255 at the centre) becomes 0, as it does in the
         * reference build on x86-64. */
        const double v = floor(M * exp((-1) * log(2) * abs(center - i) / H) + 1);
        uint8_t w = (uint8_t)(int)v;
        if (w > M) w = M;
        wt[i] = w;
    }
}

/* ------------------------------------------------------------------- FASTA */
struct gkm_problem {
    int n, n_pos, cap;
    int64_t *off;    /* [n+1] offsets into codes */
    uint8_t *codes;  /* concatenated base codes 0..3 */
    int64_t used, codes_cap;
    long invalid, truncated;
};

static int problem_reserve(gkm_problem *p, int64_t extra)
{
    if (p->n + 2 > p->cap) {
        int ncap = p->cap ? p->cap * 2 : 1024;
        int64_t *no = (int64_t *)realloc(p->off, sizeof(int64_t) * (size_t)(ncap + 1));
        if (!no) return 1;
        p->off = no;
        p->cap = ncap;
    }
    if (p->used + extra > p->codes_cap) {
        int64_t ncap = p->codes_cap ? p->codes_cap * 2 : (1 << 20);
        while (ncap < p->used + extra) ncap *= 2;
        uint8_t *nc = (uint8_t *)realloc(p->codes, (size_t)ncap);
        if (!nc) return 1;
        p->codes = nc;
        p->codes_cap = ncap;
    }
    return 0;
}

/* base codes by table: 0..3 for ACGT in either case; 4 marks "anything else", which counts as
 * 'A' like in the reference (libgkm.c:864-875) and is tallied for the warning */
static uint8_t g_code[256];
static void init_code_table(void)
{
    if (g_code[(unsigned char)'T'] == 3) return;
    for (int i = 0; i < 256; i++) g_code[i] = 4;
    g_code[(unsigned char)'A'] = g_code[(unsigned char)'a'] = 0;
    g_code[(unsigned char)'C'] = g_code[(unsigned char)'c'] = 1;
    g_code[(unsigned char)'G'] = g_code[(unsigned char)'g'] = 2;
    g_code[(unsigned char)'T'] = g_code[(unsigned char)'t'] = 3;
}

/* Single linear pass over the mapped file.  Record rules of libgkm.c:1251-1314:
 * a line starting with '>' opens a record; other lines are appended to the open
 * record up to GKM_MAX_SEQ bases; a logical line ends at the first CR or LF; text
 * before the first header is ignored; blank lines are harmless. */
static int parse_fasta(const char *path, gkm_problem *p)
{
    int fd = open(path, O_RDONLY);
    struct stat st;
    if (fd < 0) return 1;
    if (fstat(fd, &st) != 0) { close(fd); return 1; }
    const size_t size = (size_t)st.st_size;
    const char *buf = NULL;
    if (size > 0) {
        buf = (const char *)mmap(NULL, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (buf == MAP_FAILED) { close(fd); return 1; }
    }
    close(fd);

    init_code_table();
    int open_record = 0, cur = 0;
    size_t pos = 0;
    while (pos < size) {
        const char *nl = (const char *)memchr(buf + pos, '\n', size - pos);
        const size_t eol = nl ? (size_t)(nl - buf) : size;
        const char *cr = (const char *)memchr(buf + pos, '\r', eol - pos);
        const size_t end = cr ? (size_t)(cr - buf) : eol;
        if (end > pos && buf[pos] == '>') {
            if (open_record) p->off[++p->n] = p->used;
            if (problem_reserve(p, GKM_MAX_SEQ)) { munmap((void *)buf, size); return 1; }
            open_record = 1;
            cur = 0;
            p->off[p->n] = p->used;
        } else if (open_record && cur < GKM_MAX_SEQ) {
            size_t take = end - pos;
            if ((size_t)cur + take > GKM_MAX_SEQ) { /* libgkm.c:1294-1299 */
                take = (size_t)(GKM_MAX_SEQ - cur);
                p->truncated++;
            }
            uint8_t *dst = p->codes + p->used;
            long bad = 0;
            for (size_t i = 0; i < take; i++) {
                const uint8_t c = g_code[(unsigned char)buf[pos + i]];
                bad += c >> 2;
                dst[i] = c & 3;
            }
            p->invalid += bad;
            p->used += (int64_t)take;
            cur += (int)take;
        }
        pos = eol + 1;
    }
    if (open_record) p->off[++p->n] = p->used;
    if (size > 0) munmap((void *)buf, size);
    return 0;
}

typedef struct {
    const char *path;
    gkm_problem *p;
    int rc;
} parse_job;

static void *parse_thread(void *arg)
{
    parse_job *j = (parse_job *)arg;
    j->rc = parse_fasta(j->path, j->p);
    return NULL;
}

/* The two files are parsed at the same time (the negatives on a helper thread, into a problem of their own that is
 * then appended): the parse is ~1 ms per 1.5 MB file and sits in front of everything else in the drop-in call. */
gkm_problem *gkm_problem_read(const char *posfile, const char *negfile)
{
    gkm_problem *p = (gkm_problem *)calloc(1, sizeof *p), *q = (gkm_problem *)calloc(1, sizeof *q);
    if (!p || !q || problem_reserve(p, GKM_MAX_SEQ) || problem_reserve(q, GKM_MAX_SEQ)) goto fail;
    p->off[0] = 0;
    q->off[0] = 0;
    init_code_table(); /* (before the helper thread exists: both parsers only read the table) */
    parse_job neg = {negfile, q, 1};
    pthread_t th;
    const int threaded = pthread_create(&th, NULL, parse_thread, &neg) == 0;
    const int rc_pos = parse_fasta(posfile, p);
    if (threaded) pthread_join(th, NULL);
    else parse_thread(&neg);
    if (rc_pos || neg.rc) goto fail;
    p->n_pos = p->n;
    /* append the negatives: offsets shifted by the positives' bases */
    if (p->n + q->n + 2 > p->cap) {
        int64_t *no = (int64_t *)realloc(p->off, sizeof(int64_t) * (size_t)(p->n + q->n + 2));
        if (!no) goto fail;
        p->off = no;
        p->cap = p->n + q->n + 1;
    }
    if (p->used + q->used > p->codes_cap) {
        uint8_t *nc = (uint8_t *)realloc(p->codes, (size_t)(p->used + q->used + 1));
        if (!nc) goto fail;
        p->codes = nc;
        p->codes_cap = p->used + q->used + 1;
    }
    if (q->used > 0) memcpy(p->codes + p->used, q->codes, (size_t)q->used);
    for (int i = 1; i <= q->n; i++) p->off[p->n + i] = p->used + q->off[i];
    p->n += q->n;
    p->used += q->used;
    p->invalid += q->invalid;
    p->truncated += q->truncated;
    gkm_problem_free(q);
    return p;
fail:
    gkm_problem_free(p);
    gkm_problem_free(q);
    return NULL;
}

void gkm_problem_free(gkm_problem *p)
{
    if (!p) return;
    free(p->off);
    free(p->codes);
    free(p);
}

int gkm_problem_size(const gkm_problem *p) { return p->n; }
int gkm_problem_npos(const gkm_problem *p) { return p->n_pos; }
int gkm_problem_seqlen(const gkm_problem *p, int i) { return (int)(p->off[i + 1] - p->off[i]); }
const uint8_t *gkm_problem_codes(const gkm_problem *p, int i) { return p->codes + p->off[i]; }
long gkm_problem_invalid_chars(const gkm_problem *p) { return p->invalid; }
long gkm_problem_truncated(const gkm_problem *p) { return p->truncated; }
const int64_t *gkm_problem_offsets(const gkm_problem *p) { return p->off; }
const uint8_t *gkm_problem_all_codes(const gkm_problem *p) { return p->codes; }
