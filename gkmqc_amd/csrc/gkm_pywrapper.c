/*
 * gkm_pywrapper.c -- the drop-in C-ABI entry point (include/gkmkern_pylib.h).
 *
 * Thin C host: options -> host tables (c_m, positional weights) -> FASTA -> device
 * layer (include/gkm_hip.h) -> the caller's row pointers.  Replaces the reference's
 * src/gkmkern_pylib.c:92-246; the k-mer tree and the pthread row loop are gone, the
 * rows are computed on the GPU.  There is NO CPU compute path: without a usable HIP
 * device the call logs an ERROR and returns non-zero.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/gkm_hip.h"
#include "gkm_host.h"

#define GKM_MAX_DEVICES 16

/* one GPU's share when the boundary call spreads the matrix over several devices */
typedef struct {
    int device, part, nparts, L, d;
    const double *c;
    int rbf;
    double gamma;
    int n;
    const uint8_t *codes;
    const int64_t *offsets;
    const uint8_t *wd;
    int wd_len;
    double **kmat;
    int nthreads;
    int rc;
    char err[256];
    gkmhip_ctx *ctx;
    double *dG;
} device_job;

/* Two phases, with ALL devices through the first before any starts the second, so that what can reasonably fail --
 * a device that cannot be used, the upload, the n x n allocation -- fails before a single cell of the caller's
 * matrix has been written (the reference returns from its own checks before writing anything,
 * src/gkmkern_pylib.c:157-161).  GKM_FAULT_INJECT=alloc:<part> makes that part's allocation fail (tests). */
static void *device_setup(void *arg)
{
    device_job *j = (device_job *)arg;
    const char *inject = getenv("GKM_FAULT_INJECT");
    j->rc = 1;
    j->ctx = gkmhip_create(j->device, j->L, j->d, j->c, j->rbf, j->gamma);
    if (j->ctx && !gkmhip_set_sequences(j->ctx, j->n, j->codes, j->offsets, j->wd, j->wd_len, NULL)) {
        if (inject && !strncmp(inject, "alloc:", 6) && atoi(inject + 6) == j->part) {
            snprintf(j->err, sizeof j->err, "injected allocation failure (GKM_FAULT_INJECT)");
            return NULL;
        }
        j->dG = (double *)gkmhip_malloc(j->device, (size_t)j->n * (size_t)j->n * sizeof(double));
        if (j->dG) j->rc = 0;
    }
    if (j->rc) snprintf(j->err, sizeof j->err, "%s", gkmhip_last_error());
    return NULL;
}

static void *device_compute(void *arg)
{
    device_job *j = (device_job *)arg;
    j->rc = gkmhip_gram_part_to_host_rows(j->ctx, j->dG, j->n, j->kmat, j->nthreads, j->part, j->nparts) ? 1 : 0;
    if (j->rc) snprintf(j->err, sizeof j->err, "%s", gkmhip_last_error());
    return NULL;
}

/* `phase` for every job: job 0 on the calling thread, the others on threads of their own (or here, one after the
 * other, if a thread cannot be started) */
static void run_on_all_devices(device_job *jobs, int ndev, void *(*phase)(void *))
{
    pthread_t th[GKM_MAX_DEVICES];
    int started[GKM_MAX_DEVICES];
    for (int i = 1; i < ndev; i++) started[i] = pthread_create(&th[i], NULL, phase, &jobs[i]) == 0;
    phase(&jobs[0]);
    for (int i = 1; i < ndev; i++) {
        if (started[i]) pthread_join(th[i], NULL);
        else phase(&jobs[i]);
    }
}

/* Kept from call to call (single-device calls): the context -- with its per-launch scratch, ~0.6 GB at n = 10 000 --
 * and the n x n device matrix.  bin/gkmqc.py makes ~20 calls of one shape per run (gkmqc.py:341-343), and creating
 * and freeing them cost every call ~25 hipMalloc / hipFree (each waits for the device).  What the reference's
 * "callee frees everything before it returns" (src/gkmkern_pylib.c:226-243) promises still holds for HOST memory;
 * device memory of the last shape stays allocated until gkm_release_device_cache() or the end of the process.
 * GKM_KEEP_DEVICE=0 frees it on return as before.  The call is not re-entrant (include/gkmkern_pylib.h), so no lock. */
static struct {
    gkmhip_ctx *ctx;
    int device, L, d, rbf;
    double gamma, c[GKM_MAX_L + 1];
    double *dG;
    size_t dG_elems;
    long hits; /* calls that found both the context and a large enough matrix */
} g_keep;

long gkm_device_cache_hits(void) { return g_keep.hits; }

void gkm_release_device_cache(void)
{
    const int caller_device = gkmhip_current_device();
    const long hits = g_keep.hits;
    if (g_keep.dG) gkmhip_free(g_keep.dG);
    if (g_keep.ctx) gkmhip_destroy(g_keep.ctx);
    memset(&g_keep, 0, sizeof g_keep);
    g_keep.hits = hits;
    if (caller_device >= 0) gkmhip_set_current_device(caller_device);
}

static gkmhip_ctx *kept_context(int device, int L, int d, const double *c, int rbf, double gamma)
{
    if (g_keep.ctx && g_keep.device == device && g_keep.L == L && g_keep.d == d && g_keep.rbf == rbf &&
        g_keep.gamma == gamma && !memcmp(g_keep.c, c, sizeof(double) * (size_t)(d + 1)))
        return g_keep.ctx;
    gkm_release_device_cache();
    g_keep.ctx = gkmhip_create(device, L, d, c, rbf, gamma);
    if (!g_keep.ctx) return NULL;
    g_keep.device = device; g_keep.L = L; g_keep.d = d; g_keep.rbf = rbf; g_keep.gamma = gamma;
    memcpy(g_keep.c, c, sizeof(double) * (size_t)(d + 1));
    return g_keep.ctx;
}

static double *kept_matrix(int device, size_t elems)
{
    if (g_keep.dG && g_keep.dG_elems >= elems) return g_keep.dG;
    if (g_keep.dG) gkmhip_free(g_keep.dG);
    g_keep.dG_elems = 0;
    g_keep.dG = (double *)gkmhip_malloc(device, elems * sizeof(double));
    if (g_keep.dG) g_keep.dG_elems = elems;
    return g_keep.dG;
}

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int gkm_main_pywrapper(gkmOpt *opts, double **kmat, int *kmat_size)
{
    int rc = 1;
    gkm_problem *prob = NULL;
    gkmhip_ctx *ctx = NULL;
    double *dG = NULL;
    uint8_t *wd = NULL;
    int caller_device = -1;
    const double t_start = now_ms();

    if (!opts || !kmat || !kmat_size) return 1;

    const int level = gkm_log_level_from_verbosity(opts->verbosity);
    if (level < 0) { /* the reference prints this and calls exit(0), gkmkern_pylib.c:135-137 */
        fprintf(stderr, "Unknown verbosity: %d\n", opts->verbosity);
        return 1;
    }
    gkm_log_set_level(level);

    const int kt = opts->kernel_type, L = opts->L, k = opts->k, d = opts->d;
    const int weighted = (kt == EST_TRUNC_PW || kt == EST_TRUNC_PW_RBF);
    const int rbf = (kt == EST_TRUNC_RBF || kt == EST_TRUNC_PW_RBF);

    /* same INFO lines as gkmkern_pylib.c:140-155 */
    gkm_log(GKM_LOG_INFO, "Arguments:");
    gkm_log(GKM_LOG_INFO, "  posfile = %s", opts->posfile ? opts->posfile : "(null)");
    gkm_log(GKM_LOG_INFO, "  negfile = %s", opts->negfile ? opts->negfile : "(null)");
    gkm_log(GKM_LOG_INFO, "Parameters:");
    gkm_log(GKM_LOG_INFO, "  kernel-type = %d", kt);
    gkm_log(GKM_LOG_INFO, "  L = %d", L);
    gkm_log(GKM_LOG_INFO, "  k = %d", k);
    gkm_log(GKM_LOG_INFO, "  d = %d", d);
    if (rbf) gkm_log(GKM_LOG_INFO, "  gamma = %g", opts->gamma);
    if (weighted) {
        gkm_log(GKM_LOG_INFO, "  M = %d", opts->M);
        gkm_log(GKM_LOG_INFO, "  H = %g", opts->H);
    }

    const char *bad = gkm_check_parameter_values(kt, L, k, d);
    if (bad) {
        gkm_log(GKM_LOG_ERROR, "%s", bad);
        return 1;
    }
    if (!opts->posfile || !opts->negfile) {
        gkm_log(GKM_LOG_ERROR, "can't open file");
        return 1;
    }

    double c[GKM_MAX_L + 1] = {0};
    if (gkm_mismatch_weights(kt, L, k, c)) { /* e.g. k < 0, which the reference's check lets through */
        gkm_log(GKM_LOG_ERROR, "k should be in the range 0..L");
        return 1;
    }
    gkm_log(GKM_LOG_DEBUG, "gkm-kernel weights:");
    for (int m = 0; m <= d; m++) gkm_log(GKM_LOG_DEBUG, "  c[%d] = %.6f", m, c[m]);

    prob = gkm_problem_read(opts->posfile, opts->negfile);
    if (!prob) {
        gkm_log(GKM_LOG_ERROR, "can't open file");
        return 1;
    }
    const int n = gkm_problem_size(prob), n_pos = gkm_problem_npos(prob);
    gkm_log(GKM_LOG_INFO, "read %d sequences from %s", n_pos, opts->posfile);
    gkm_log(GKM_LOG_INFO, "read %d sequences from %s", n - n_pos, opts->negfile);
    if (n_pos == 0 || n == n_pos) {
        gkm_log(GKM_LOG_ERROR, "no sequences in %s", n_pos == 0 ? opts->posfile : opts->negfile);
        goto done;
    }
    if (gkm_problem_invalid_chars(prob) > 0)
        gkm_log(GKM_LOG_WARN, "%ld characters are not valid nucleotides and were read as 'A'. Only ACGT are allowed",
                gkm_problem_invalid_chars(prob));
    if (gkm_problem_truncated(prob) > 0)
        gkm_log(GKM_LOG_WARN, "maximum sequence length allowed is %d. Only the first %d nucleotides of %ld longer sequence(s) are used",
                GKM_MAX_SEQ, GKM_MAX_SEQ, gkm_problem_truncated(prob));

    /* positional weights: the reference's w(n, p) depends on n and p only through the
     * distance |n/2 - p| to the centre l-mer (libgkm.c:912-925), so one table indexed by
     * that distance serves every sequence length */
    int maxn = 0;
    for (int i = 0; i < n; i++) {
        const int len = gkm_problem_seqlen(prob, i);
        if (len < L) { /* undefined behaviour in the reference (negative l-mer count) */
            gkm_log(GKM_LOG_ERROR, "sequence %d has %d nucleotides, fewer than L = %d", i, len, L);
            goto done;
        }
        if (len - L + 1 > maxn) maxn = len - L + 1;
    }
    int wd_len = 0;
    if (weighted) {
        const int dmax = maxn / 2 + 1;
        uint8_t *full = (uint8_t *)malloc((size_t)(2 * dmax + 1));
        wd = (uint8_t *)malloc((size_t)dmax + 1);
        if (!full || !wd) { free(full); goto done; }
        gkm_position_weights(kt, 2 * dmax + 1, opts->M, opts->H, full); /* centre index = dmax */
        memcpy(wd, full + dmax, (size_t)dmax + 1);
        free(full);
        wd_len = dmax + 1;
    }
    const double t_parsed = now_ms();

    /* devices: GKM_DEVICES = "all", a count ("4") or a list ("0,2,3"); GKM_DEVICE = one ordinal.
     * Several GPUs: one context + host thread per device, each filling disjoint row blocks of the
     * caller's matrix (no collective needed: the result goes to host memory anyway). */
    int devs[GKM_MAX_DEVICES], ndev = 0;
    {
        const char *list = getenv("GKM_DEVICES");
        const char *one = getenv("GKM_DEVICE");
        const int avail = gkmhip_device_count();
        int bad_list = 0;
        if (list && *list) {
            if (!strcmp(list, "all")) {
                for (int i = 0; i < avail && ndev < GKM_MAX_DEVICES; i++) devs[ndev++] = i;
            } else if (strchr(list, ',')) {
                for (const char *q = list; ndev < GKM_MAX_DEVICES;) {
                    char *end = NULL;
                    const long v = strtol(q, &end, 10);
                    if (end == q || (*end && *end != ',') || v < 0 || v >= avail) { bad_list = 1; break; }
                    devs[ndev++] = (int)v;
                    if (!*end) break;
                    q = end + 1;
                }
            } else {
                char *end = NULL;
                const long v = strtol(list, &end, 10);
                if (end == list || *end || v < 1 || v > avail) bad_list = 1;
                for (int i = 0; !bad_list && i < (int)v && ndev < GKM_MAX_DEVICES; i++) devs[ndev++] = i;
            }
        } else if (one && *one) {
            char *end = NULL;
            const long v = strtol(one, &end, 10);
            if (end == one || *end || v < 0 || v >= (avail > 0 ? avail : 1)) bad_list = 1;
            else devs[ndev++] = (int)v;
        }
        if (bad_list) {
            gkm_log(GKM_LOG_ERROR, "GKM_DEVICES / GKM_DEVICE must name HIP devices 0..%d (\"all\", a count, or a comma-separated list)",
                    avail - 1);
            goto done;
        }
        if (ndev == 0) devs[ndev++] = 0;
    }
    /* the call must not change the caller's current HIP device (it may be a torch process) */
    caller_device = gkmhip_current_device();
    double t_created = t_parsed, t_uploaded = t_parsed, t_alloc = t_parsed;
    const char *keep_env = getenv("GKM_KEEP_DEVICE");
    const int keep = !(keep_env && !strcmp(keep_env, "0"));
    if (ndev == 1) {
        const int device = devs[0];
        const int warm = keep && g_keep.ctx && g_keep.dG && g_keep.dG_elems >= (size_t)n * (size_t)n;
        ctx = keep ? kept_context(device, L, d, c, rbf, opts->gamma) : gkmhip_create(device, L, d, c, rbf, opts->gamma);
        if (warm && ctx && g_keep.dG) g_keep.hits++;
        t_created = now_ms();
        if (!ctx) {
            gkm_log(GKM_LOG_ERROR, "cannot use HIP device %d: %s", device, gkmhip_last_error());
            goto done;
        }
        if (gkmhip_set_sequences(ctx, n, gkm_problem_all_codes(prob), gkm_problem_offsets(prob), wd, wd_len, NULL)) {
            gkm_log(GKM_LOG_ERROR, "device upload failed: %s", gkmhip_last_error());
            goto done;
        }
        t_uploaded = now_ms();
        dG = keep ? kept_matrix(device, (size_t)n * (size_t)n)
                  : (double *)gkmhip_malloc(device, (size_t)n * (size_t)n * sizeof(double));
        t_alloc = now_ms();
        if (!dG) {
            gkm_log(GKM_LOG_ERROR, "device allocation of the %d x %d matrix failed: %s", n, n, gkmhip_last_error());
            goto done;
        }
        /* rows a: K(a, 0..a-1) and the unit diagonal -- exactly the cells the reference writes;
         * computed, normalised and shipped block by block (compute overlaps the PCIe transfer) */
        if (gkmhip_gram_to_host_rows(ctx, dG, n, kmat, opts->nthreads > 0 ? opts->nthreads : 1)) {
            gkm_log(GKM_LOG_ERROR, "gram kernel failed: %s", gkmhip_last_error());
            goto done;
        }
    } else {
        device_job jobs[GKM_MAX_DEVICES];
        int failed = 0;
        for (int i = 0; i < ndev; i++) {
            device_job j = {devs[i], i, ndev, L, d, c, rbf, opts->gamma, n, gkm_problem_all_codes(prob),
                            gkm_problem_offsets(prob), wd, wd_len, kmat, opts->nthreads > 0 ? opts->nthreads : 1, 0, {0},
                            NULL, NULL};
            jobs[i] = j;
        }
        run_on_all_devices(jobs, ndev, device_setup);
        for (int i = 0; i < ndev; i++)
            if (jobs[i].rc) {
                gkm_log(GKM_LOG_ERROR, "HIP device %d: %s", jobs[i].device, jobs[i].err);
                failed = 1;
            }
        /* nothing has been written to the caller's rows yet: a device that failed here leaves them untouched */
        if (!failed) {
            run_on_all_devices(jobs, ndev, device_compute);
            for (int i = 0; i < ndev; i++)
                if (jobs[i].rc) {
                    gkm_log(GKM_LOG_ERROR, "HIP device %d: %s", jobs[i].device, jobs[i].err);
                    failed = 1;
                }
        }
        for (int i = 0; i < ndev; i++) {
            if (jobs[i].dG) gkmhip_free(jobs[i].dG);
            if (jobs[i].ctx) gkmhip_destroy(jobs[i].ctx);
        }
        if (failed) goto done;
        gkm_log(GKM_LOG_DEBUG, "row blocks computed on %d devices", ndev);
    }
    const double t_kernel = now_ms();
    kmat_size[0] = n_pos;
    kmat_size[1] = n - n_pos;
    rc = 0;
    gkm_log(GKM_LOG_DEBUG, "timing: read+tables %.1f ms, context %.1f ms, upload %.1f ms, device malloc %.1f ms, "
            "gram + copy-out pipeline %.1f ms (kernel %s)", t_parsed - t_start, t_created - t_parsed,
            t_uploaded - t_created, t_alloc - t_uploaded, t_kernel - t_alloc, ctx ? gkmhip_last_kernel_name(ctx) : "multi");

done:;
    const double t_done = now_ms();
    if (ctx && ctx == g_keep.ctx) { /* kept for the next call of this shape (see g_keep) */
        if (rc) gkm_release_device_cache(); /* ... but not after a failure: the next call starts from scratch */
    } else {
        if (dG) gkmhip_free(dG);
        if (ctx) gkmhip_destroy(ctx);
    }
    if (caller_device >= 0) gkmhip_set_current_device(caller_device);
    free(wd);
    gkm_problem_free(prob);
    gkm_log(GKM_LOG_DEBUG, "timing: teardown %.1f ms, whole call %.1f ms", now_ms() - t_done, now_ms() - t_start);
    return rc;
}
