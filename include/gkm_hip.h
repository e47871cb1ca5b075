/*
 * gkm_hip.h -- device-layer C ABI of the gkm kernel-matrix path (MI355X / gfx950).
 *
 * Plain pointers and sizes only; no torch / C++ types.  `gkm_main_pywrapper`
 * (include/gkmkern_pylib.h) is a thin C host on top of these calls; bench.py and
 * the GPU tests bind them with ctypes and pass device pointers and a HIP stream
 * obtained from PyTorch-ROCm.
 *
 * What each call replaces in the reference (all in src/libgkm.c unless noted):
 *   gkmhip_create / gkmhip_destroy   gkmkernel_init :978-1033, gkmkernel_destroy :1058-1068
 *   gkmhip_set_sequences             gkmkernel_new_object :841-938 (encoding, rc strand,
 *                                    positional weights) + gkmkernel_build_tree :1035-1056
 *                                    (the k-mer tree is replaced by bit-plane tables)
 *   gkmhip_gram_rows                 the row loop pthread_gkmkernel_kernelfunc_batch_all
 *                                    (src/gkmkern_pylib.c:70-90) -> gkmkernel_kernelfunc_batch_all
 *                                    :1156-1185 -> kmertree_dfs :315-387, for a set of rows;
 *                                    it also yields the self terms of
 *                                    gkmkernel_kernelfunc_sqnorm_single :723-759
 *   gkmhip_normalize                 the division / RBF of :1168-1179 and the unit diagonal
 *                                    of src/gkmkern_pylib.c:218-221
 *
 * Output convention: `G` is a row-major fp64 matrix with leading dimension `ld`
 * (elements).  gkmhip_gram_rows writes RAW values G(a,j) = sum_m c_m P_m(a,j) for
 * j <= a (diagonal included, nothing above it).  gkmhip_normalize turns a matrix
 * of raw values (all rows present) into K in place.
 *
 * All functions return 0 on success and a non-zero code on failure;
 * gkmhip_last_error() describes the most recent failure of the calling thread.
 */
#ifndef GKM_HIP_H
#define GKM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gkmhip_ctx gkmhip_ctx;

enum { GKMHIP_KERNEL_AUTO = 0, GKMHIP_KERNEL_DIRECT = 1, GKMHIP_KERNEL_BITSLICE = 2 };

const char *gkmhip_last_error(void);
int gkmhip_device_count(void);
/* the calling thread's current HIP device (-1 if there is none) / make `device` current: the
 * boundary call restores the caller's device before it returns */
int gkmhip_current_device(void);
int gkmhip_set_current_device(int device);

/* c: the d+1 mismatch weights c_0..c_d (host-computed, include/gkmkern_pylib.h).
 * rbf != 0 selects K <- exp(gamma (K-1)) in gkmhip_normalize (kernel types 3, 5). */
gkmhip_ctx *gkmhip_create(int device, int L, int d, const double *c, int rbf, double gamma);
void gkmhip_destroy(gkmhip_ctx *ctx);

int gkmhip_n_sequences(const gkmhip_ctx *ctx); /* sequences uploaded */
int gkmhip_device_of(const gkmhip_ctx *ctx);
void gkmhip_set_error_message(const char *msg); /* what gkmhip_last_error() returns next (calling thread) */

/* choose the kernel family (default AUTO: the bit-sliced kernel where it is instantiated for (L, d) AND the faster one --
 * at most ~7.5 % of the window pairs within d mismatches for iid sequences --, the general kernel elsewhere;
 * BITSLICE fails where the kernel is not instantiated) */
/* Per-launch scratch (row tables of one gkmhip_gram_rows* call) exists twice.  A caller that issues
 * consecutive launches on two different streams, so that one launch fills the CUs the previous one is
 * draining, selects slot 0 / 1 alternately; launches that share a slot must share a stream. */
int gkmhip_set_scratch_slot(gkmhip_ctx *ctx, int slot);

int gkmhip_set_kernel(gkmhip_ctx *ctx, int which);

/* Upload n sequences.  codes: base codes 0..3, concatenated; offsets[n+1] (elements).
 * wdist: positional weights as a function of the distance to the centre l-mer: the
 * weight of l-mer p of a sequence with n l-mers is wdist[|n/2 - p|] (this is all the
 * reference's exponential-decay weights depend on, src/libgkm.c:912-925); it must cover
 * distances 0..max(n)/2, at most 1024 entries.  wdist == NULL: all weights are 1.
 * The upload runs on `stream` and has completed when the call returns (the arrays may be freed);
 * device tables are built on first use. */
int gkmhip_set_sequences(gkmhip_ctx *ctx, int n, const uint8_t *codes, const int64_t *offsets,
                         const uint8_t *wdist, int wdist_len, void *stream);

/* Raw Gram values for the rows listed in `rows` (host array of nrows ascending sequence
 * indices): row rows[i] is written to G + i*ld (local_rows != 0) or to
 * G + rows[i]*ld (local_rows == 0).  `G` is a DEVICE pointer.  If P != NULL (device,
 * int32) the integer mismatch profiles are stored as P[(i_or_row*ldp + j)*(d+1) + m]
 * for j <= a.  Work is enqueued on `stream` (a hipStream_t, may be NULL). */
int gkmhip_gram_rows(gkmhip_ctx *ctx, const int *rows, int nrows, int local_rows, double *G,
                     int64_t ld, int32_t *P, int64_t ldp, void *stream);

/* The same rows into a PACKED slab: row rows[i] is written to G + row_off[i] (host array of nrows element offsets),
 * columns 0..rows[i] only -- rows[i] + 1 doubles, so rows may sit back to back.  This is the send buffer of the
 * multi-GPU all-gather (gkmhip_gram_allgather; layout in gkmqc_amd/csrc/gkm_shard.h): only j <= a is ever read, and
 * shipping n doubles per row moved twice the bytes (round 3). */
int gkmhip_gram_rows_packed(gkmhip_ctx *ctx, const int *rows, int nrows, double *G, const int64_t *row_off,
                            void *stream);

/* Rectangular variant for prediction-style use (the batch-vs-support-vector call of the
 * reference, gkmkernel_kernelfunc_batch, src/libgkm.c:1115-1153): raw G(rows[i], j) for EVERY
 * uploaded sequence j (not only j <= a).  G needs ld >= n. */
int gkmhip_gram_rows_full(gkmhip_ctx *ctx, const int *rows, int nrows, int local_rows, double *G, int64_t ld,
                          void *stream);

/* sqnorm[i] = sqrt(G(i,i)) for all uploaded sequences (device array of n doubles), computed
 * from the diagonal band only (~1 % of the work of the whole matrix).  Replaces
 * gkmkernel_kernelfunc_sqnorm_single, src/libgkm.c:723-759. */
int gkmhip_self_norms(gkmhip_ctx *ctx, double *sqnorm, void *stream);

/* K(rows[i], j) = G / (sqnorm[rows[i]] sqnorm[j]) (+ RBF), 1.0 where j == rows[i], for rows written by
 * gkmhip_gram_rows_full with the same rows / local_rows. */
int gkmhip_normalize_rows_full(gkmhip_ctx *ctx, const int *rows, int nrows, int local_rows, double *G, int64_t ld,
                               const double *sqnorm, void *stream);

/* In place on a device matrix holding raw values for ALL n rows (lower triangle +
 * diagonal): K(a,j) = G(a,j) / (sqrt(G(a,a)) sqrt(G(j,j))), optional RBF, K(a,a)=1.
 * If sqnorm != NULL (device, n doubles) it receives sqrt(G(a,a)).
 * symmetric != 0 additionally mirrors the lower triangle into the upper one. */
int gkmhip_normalize(gkmhip_ctx *ctx, double *G, int64_t ld, double *sqnorm, int symmetric,
                     void *stream);

/* plain device memory helpers so that a C host needs nothing but this header */
void *gkmhip_malloc(int device, size_t bytes);
void gkmhip_free(void *p);
int gkmhip_memcpy_d2h(void *dst, const void *src, size_t bytes);
int gkmhip_memcpy_h2d(void *dst, const void *src, size_t bytes);
int gkmhip_sync(void *stream);

/* Copy the lower triangle (+ diagonal) of a device K (n rows, ld) into caller-owned
 * host row pointers: rows[a][0..a].  Uses pinned staging and `nthreads` host threads. */
int gkmhip_copy_lower_to_rows(gkmhip_ctx *ctx, const double *K, int64_t ld, int n,
                              double **rows, int nthreads);

/* Whole Gram matrix straight into caller-owned host rows (rows[a][0..a] = K(a, 0..a-1), 1.0):
 * gram + normalise + device-to-host as a pipeline over row blocks of about equal work, so that
 * the PCIe transfer and the host-side scatter of one block overlap the kernel of the next.
 * G: device scratch of n x ld doubles.  This is what gkm_main_pywrapper uses. */
int gkmhip_gram_to_host_rows(gkmhip_ctx *ctx, double *G, int64_t ld, double **rows, int nthreads);

/* The same for one of `nparts` contexts (one per GPU, one host thread each) filling disjoint row
 * blocks of ONE host matrix: context `part` takes every nparts-th block; self norms come from a
 * diagonal-band pass, so no device needs another device's rows and no collective is involved.
 * This is how gkm_main_pywrapper uses several GPUs of a node (GKM_DEVICES). */
int gkmhip_gram_part_to_host_rows(gkmhip_ctx *ctx, double *G, int64_t ld, double **rows, int nthreads, int part,
                                  int nparts);

/* ---- several GPUs, one host process (gkm_multi.hip) ----
 * Every context (one per device, same parameters, same sequences uploaded) computes the rows of its
 * folded row blocks; the row slabs -- packed, a + 1 doubles for row a: n^2 / (2 nctx) doubles per rank and
 * matrix -- are all-gathered over xGMI (RCCL ncclAllGather on communicators
 * made by ncclCommInitAll; peer copies when several contexts share one device or RCCL cannot be
 * loaded; GKM_ALLGATHER=rccl|p2p forces one), then every device un-permutes and normalises its copy.
 * K[g]: device pointer ON ctxs[g]'s DEVICE to an n x ld matrix that receives K (lower triangle +
 * unit diagonal, the upper triangle too if symmetric != 0) -- the same matrix, bit for bit, as
 * gkmhip_gram_rows + gkmhip_normalize produce on one device.  chunks: slabs per rank whose transfer
 * overlaps the next slab's kernel (0 = chosen from (n, nctx): 1 for one context, else 2, or 3 / 4 where that pads the slabs 3 % less -- gkm_shard.h auto_chunks).  One host thread per device for the duration of
 * the call; blocks until every device holds the matrix.  This is what feeds the GPU-resident
 * cross-validation (include/gkm_svm.h) from an N-GPU matrix; the reference's consumer is
 * scripts/gkmsvm.py:104-122. */
int gkmhip_gram_allgather(gkmhip_ctx **ctxs, int nctx, double **K, int64_t ld, int symmetric, int chunks);
/* ONE rank of a `ranks`-way gkmhip_gram_allgather ALONE on its device (measurement: what a rank's step costs without
 * the transfer, on a box with one GPU): rank `rank`'s chunks exactly as gkmhip_gram_allgather runs them (same layout,
 * streams, scratch slots, packed slabs), its slab copied into its gathered buffer on the device, then the whole matrix
 * assembled and normalised into K from that buffer -- whose other ranks' slabs must be there from an earlier
 * gkmhip_gram_allgather over `ranks` contexts with the same `chunks` (all on this device: the one-GPU rehearsal).
 * out6 = {wall ms on the host clock, kernels ms (sum over the chunks' launches incl. tables / row planes / untile),
 * copy-in ms, un-permute + normalise ms, l-mer comparisons, chunks}. */
int gkmhip_gram_rank_alone(gkmhip_ctx *ctx, int rank, int ranks, int chunks, double *K, int64_t ld, int symmetric,
                           double *out6);
/* "rccl", "p2p" or "none": how the most recent gkmhip_gram_allgather moved the slabs */
const char *gkmhip_last_transport(void);
/* RCCL communicators AND each rank's buffers (its slab, the gathered slabs, gather index, self norms, streams,
 * events: ~1.7 GB per device at n = 10 000 on two devices) are kept for the life of the process, keyed by
 * (device, n, ranks, chunks): bin/gkmqc.py asks for ~20 matrices of one size per run, and hipMalloc / hipFree
 * synchronise the device.  This destroys the communicators and frees the buffers (optional). */
void gkmhip_release_comms(void);
/* hipMalloc calls gkmhip_gram_allgather has made so far in this process (a second call of the same shape makes
 * none) */
long gkmhip_allgather_alloc_count(void);
/* bytes every rank RECEIVED from its peers in the most recent gkmhip_gram_allgather (per matrix) */
long long gkmhip_allgather_bytes_per_rank(void);
/* What the most recent successful gkmhip_gram_allgather measured with HIP events on its own streams:
 * out[0] = ranks, out[1] = chunks, out[2] = transport (0 none, 1 peer copies, 2 RCCL), then for every rank
 * {kernel ms summed over its chunks, transfer ms summed over its chunks, un-permute + normalise ms, l-mer
 * comparisons of its rows}.  Returns the number of doubles written, 0 if `cap` is too small or nothing ran. */
int gkmhip_allgather_stats(double *out, int cap);
/* WHEN the chunks of rank `rank` ran in the most recent gkmhip_gram_allgather / gkmhip_gram_rank_alone: for every chunk
 * c, out[4c .. 4c+3] = start and end of its launch group (tables, row planes, Gram kernel, untile) and start and end of
 * its transfer, in ms from the start of the rank's first launch group (HIP event timestamps).  The transfer of chunk c
 * can only hide behind the kernel of chunk c + 1 if chunk c ENDS well before chunk c + 1 does.  Returns the number of
 * doubles written, 0 if `cap` is too small or nothing ran. */
int gkmhip_allgather_chunk_times(int rank, double *out, int cap);

/* Un-permutation + normalisation in one pass: matrix row a is row slot_of_row[a] (device array, n
 * int64) of `slabs` (device, leading dimension lds >= n, raw values); with lds == 1 slot_of_row[a] is the
 * element offset at which row a starts (packed slabs: gkmhip_gram_rows_packed).  K receives what
 * gkmhip_normalize would produce, sqnorm (device, n doubles) the self norms. */
int gkmhip_assemble_normalize(gkmhip_ctx *ctx, const double *slabs, int64_t lds, const int64_t *slot_of_row,
                              double *K, int64_t ld, double *sqnorm, int symmetric, void *stream);

/* A new non-blocking HIP stream on the current device that is PROVEN to run beside the `nbusy` streams of `busy`
 * (hipStream_t each): HIP maps streams onto a few hardware queues in creation order and two streams that share one
 * execute in order -- a copy or collective stream that lands on the compute stream's queue overlaps nothing.  Each
 * candidate is tried with a 2-ms spin kernel on the busy stream and a 4-byte copy on the candidate; after six
 * candidates the last one is returned anyway (*beside = 0).  NULL on error.  Destroy it with hipStreamDestroy. */
void *gkmhip_create_stream_beside(void *const *busy, int nbusy, int *beside);
/* The same with a stream priority (hipStreamCreateWithPriority; hipDeviceGetStreamPriorityRange gives the range, the
 * numerically LOWEST value is the highest priority). */
void *gkmhip_create_stream_beside_prio(void *const *busy, int nbusy, int *beside, int priority);
/* A compute stream that leaves `reserve_cus` (8, 16, 24 or 32: the same number in each of the MI355X's 8 XCDs) compute
 * units to OTHER streams (hipExtStreamCreateWithCUMask).  Why: a workgroup of several waves -- a collective's kernel, a
 * blit copy -- cannot start while the Gram kernel holds 7 of 8 wave slots and 504 of 512 VGPRs of every SIMD (each wave
 * that retires is replaced by the kernel's next one-wave workgroup at once); enqueued mid-kernel on a high-priority stream
 * it ends when the kernel does (tools/collective_beside_probe.py).  NULL if the device is not laid out as 8 x 32 CUs. */
void *gkmhip_create_stream_reserving(void *const *busy, int nbusy, int *beside, int reserve_cus);
/* Keeps `stream` busy for `microseconds` (at most 10 000) with one wave that does nothing. */
int gkmhip_pause_stream(void *stream, int microseconds);
/* Measurement only: `blocks` workgroups of `threads` threads that hold their wave slots for `microseconds` and do nothing
 * (a collective's workgroups waiting for their peers). */
int gkmhip_probe_spin(int blocks, int threads, int microseconds, void *stream);
/* Measurement only: copies `bytes` (a multiple of 16) on the device with `blocks` workgroups of `threads` threads on
 * `stream` -- the launch shape of a collective's kernel (tools/collective_beside_probe.py). */
int gkmhip_probe_copy(void *dst, const void *src, size_t bytes, int blocks, int threads, void *stream);

/* The pinned staging buffers of the copy-out calls (2 x 64 MB) are kept for the life of the
 * process; this releases them (optional). */
void gkmhip_release_host_cache(void);

/* elapsed milliseconds of the device work of the most recent gkmhip_gram_rows call
 * (HIP events recorded on its stream around the dominant kernel); <0 if unavailable */
double gkmhip_last_kernel_ms(gkmhip_ctx *ctx);
/* Timing a LOOP of launches from outside: between gkmhip_kernel_timeline(ctx, 1) and gkmhip_kernel_timeline(ctx, 0) every
 * launch keeps its own pair of events (no host wait in between); gkmhip_kernel_timeline_ms waits for them and returns
 * the sum of the Gram kernels' durations since the last switch-on (*launches = how many), <0 on failure. */
int gkmhip_kernel_timeline(gkmhip_ctx *ctx, int on);
double gkmhip_kernel_timeline_ms(gkmhip_ctx *ctx, int *launches);
/* ... and WHEN they ran: out[2i], out[2i+1] = start and end of the i-th Gram kernel since the switch-on, in ms from the
 * start of the first (launches on different streams may overlap).  Returns the number of doubles written. */
int gkmhip_kernel_timeline_spans(gkmhip_ctx *ctx, double *out, int cap);
/* number of l-mer comparisons that call evaluated (algorithmic: 2 n_a n_j per pair) */
double gkmhip_last_comparisons(gkmhip_ctx *ctx);
const char *gkmhip_last_kernel_name(gkmhip_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
