/*
 * gkm_normalize.hip -- raw Gram values -> kernel values: K(a,j) = G(a,j) / (sqrt(G(a,a)) sqrt(G(j,j))), RBF types
 * exp(gamma (K - 1)), unit diagonal (src/libgkm.c:753-758,1156-1185; src/gkmkern_pylib.c:218-221); the same from the
 * all-gathered row slabs of the multi-GPU path (gkm_multi.hip).  HBM-bound, 16 bytes per pair.
 */
#include "gkm_internal.h"

/* ------------------------------------------------------------ normalise */
__global__ void k_sqnorm(const double *__restrict__ G, int64_t ld, int r0, int r1, double *__restrict__ sq)
{
    const int i = r0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < r1) sq[i] = sqrt(G[(int64_t)i * ld + i]); /* libgkm.c:753-758 */
}

/* K(a,j) = G(a,j) / (sq_a * sq_j): product first, one division (libgkm.c:1169-1172);
 * RBF types: exp(gamma (K-1)) (:1175-1179); K(a,a) = 1.0 (gkmkern_pylib.c:218-221) */
__global__ void k_normalize(double *__restrict__ G, int64_t ld, int r0, const double *__restrict__ sq,
                            int rbf, double gamma, int symmetric)
{
    const int a = r0 + blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > a) return;
    double v;
    if (j == a) {
        v = 1.0;
    } else {
        v = G[(int64_t)a * ld + j] / (sq[a] * sq[j]);
        if (rbf) v = exp(gamma * (v - 1));
        if (symmetric) G[(int64_t)j * ld + a] = v;
    }
    G[(int64_t)a * ld + j] = v;
}

/* K(rows[i], j) = G / (sq[rows[i]] * sq[j]) for every column j of a block of full rows */
__global__ void k_normalize_full(double *__restrict__ G, int64_t ld, const int *__restrict__ rows, int local_rows, int n,
                                 const double *__restrict__ sq, int rbf, double gamma)
{
    const int a = rows[blockIdx.y];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double *cell = G + (int64_t)(local_rows ? (int)blockIdx.y : a) * ld + j;
    double v = 1.0;
    if (j != a) {
        v = *cell / (sq[a] * sq[j]);
        if (rbf) v = exp(gamma * (v - 1));
    }
    *cell = v;
}

extern "C" int gkmhip_normalize_rows_full(gkmhip_ctx *ctx, const int *rows, int nrows, int local_rows, double *G,
                                          int64_t ld, const double *sqnorm, void *stream_)
{
    if (!ctx || !rows || nrows <= 0 || !G || !sqnorm) return set_err_msg("gkmhip_normalize_rows_full: bad arguments", 2);
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->scratch[ctx->sel].rows.ensure((size_t)nrows)) return 4;
    HIPCHK(hipMemcpyAsync(ctx->scratch[ctx->sel].rows.p, rows, (size_t)nrows * sizeof(int), hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream)); /* `rows` is the caller's: see gkmhip_set_sequences */
    hipLaunchKernelGGL(k_normalize_full, dim3((unsigned)((ctx->n + 255) / 256), (unsigned)nrows), dim3(256), 0, stream, G, ld,
                       ctx->scratch[ctx->sel].rows.p, local_rows, ctx->n, sqnorm, ctx->rbf, ctx->gamma);
    HIPCHK(hipGetLastError());
    return 0;
}

/* rows r0..r1-1 of a matrix whose rows < r1 hold raw values: needs sqrt(G(j,j)) for j < r1 only,
 * so row blocks can be normalised (and shipped) in ascending order while later ones compute */
int normalize_rows(gkmhip_ctx *ctx, double *G, int64_t ld, int r0, int r1, double *sq, int symmetric, hipStream_t stream,
                   bool have_norms)
{
    if (!have_norms) { /* take the norms of these rows from their own diagonal */
        hipLaunchKernelGGL(k_sqnorm, dim3((unsigned)((r1 - r0 + 255) / 256)), dim3(256), 0, stream, G, ld, r0, r1, sq);
        HIPCHK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_normalize, dim3((unsigned)((r1 + 255) / 256), (unsigned)(r1 - r0)), dim3(256), 0, stream, G,
                       ld, r0, sq, ctx->rbf, ctx->gamma, symmetric);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int gkmhip_normalize(gkmhip_ctx *ctx, double *G, int64_t ld, double *sqnorm, int symmetric,
                                void *stream_)
{
    if (!ctx || !G || ctx->n <= 0) return set_err_msg("gkmhip_normalize: bad arguments", 2);
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(hipSetDevice(ctx->device));
    double *sq = sqnorm;
    if (!sq) {
        if (ctx->sq.ensure((size_t)ctx->n)) return 4;
        sq = ctx->sq.p;
    }
    return normalize_rows(ctx, G, ld, 0, ctx->n, sq, symmetric, stream);
}

/* The same from row slabs (multi-GPU assembly, gkm_multi.hip): matrix row a is row slot[a] of `src`
 * (leading dimension lds); un-permutation and normalisation in one pass */
__global__ void k_assemble_sqnorm(const double *__restrict__ src, int64_t lds, const int64_t *__restrict__ slot, int n,
                                  double *__restrict__ sq)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sq[i] = sqrt(src[slot[i] * lds + i]); /* libgkm.c:753-758 */
}

__global__ void k_assemble_normalize(const double *__restrict__ src, int64_t lds, const int64_t *__restrict__ slot,
                                     double *__restrict__ K, int64_t ld, const double *__restrict__ sq, int rbf,
                                     double gamma, int symmetric)
{
    const int a = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > a) return;
    double v = 1.0;
    if (j != a) {
        v = src[slot[a] * lds + j] / (sq[a] * sq[j]); /* libgkm.c:1169-1172 */
        if (rbf) v = exp(gamma * (v - 1));
        if (symmetric) K[(int64_t)j * ld + a] = v;
    }
    K[(int64_t)a * ld + j] = v;
}

extern "C" int gkmhip_assemble_normalize(gkmhip_ctx *ctx, const double *slabs, int64_t lds, const int64_t *slot_of_row,
                                         double *K, int64_t ld, double *sqnorm, int symmetric, void *stream_)
{
    /* lds >= n: slot_of_row[a] is the ROW of a row-major [*, lds] array that holds matrix row a;
     * lds == 1: slot_of_row[a] is the element offset at which matrix row a starts (packed slabs, gkm_shard.h) */
    if (!ctx || !slabs || !slot_of_row || !K || !sqnorm || ctx->n <= 0 || ld < ctx->n || (lds < ctx->n && lds != 1))
        return set_err_msg("gkmhip_assemble_normalize: bad arguments", 2);
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(hipSetDevice(ctx->device));
    const int n = ctx->n;
    hipLaunchKernelGGL(k_assemble_sqnorm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, slabs, lds, slot_of_row, n, sqnorm);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_assemble_normalize, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, stream, slabs, lds,
                       slot_of_row, K, ld, sqnorm, ctx->rbf, ctx->gamma, symmetric);
    HIPCHK(hipGetLastError());
    return 0;
}
