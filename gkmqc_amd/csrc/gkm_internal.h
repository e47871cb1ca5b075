/*
 * gkm_internal.h -- what the translation units of the MI355X (gfx950) device layer share.  Not installed; the public
 * C ABI is include/gkm_hip.h.
 *
 *   gkm_context.hip        errors, context, upload, per-sequence device tables (k_build_sb, k_pack_strands, k_pack_lmers),
 *                          pinned host pools, memory helpers
 *   gkm_gram_bitslice.hip  HOT: k_gram_bitslice (bit-sliced diagonal mismatch profiles) and its instantiation table
 *   gkm_gram.hip           launch geometry (row packing, work-item order, per-launch tables), k_build_rowplanes, k_untile,
 *                          the general kernel k_gram_direct, gkmhip_gram_rows*
 *   gkm_normalize.hip      k_sqnorm, k_normalize, k_assemble_normalize: square roots of the diagonal, division, RBF
 *   gkm_copyout.hip        device matrix -> the caller's host rows: row blocks, staging pieces, the stream prober
 */
#ifndef GKM_INTERNAL_H
#define GKM_INTERNAL_H

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/gkm_hip.h"
#include "gkm_bitslice.h"
#include "gkm_pack.h"

#define GKM_MAXD1 13 /* d <= 12 */
#define GKM_MAXLEN 2047 /* longest sequence (the reference truncates there: libgkm.h:29-34, libgkm.c:1286-1291) */
#define GKM_SCRATCH_SLOTS 2 /* per-launch scratch sets of a context (gkmhip_set_scratch_slot) */

/* ------------------------------------------------------------------ errors (gkm_context.hip) */
int gkm_set_err(const char *what, hipError_t e, const char *file, int line);
int gkm_set_err_msg(const std::string &m, int code);
#define set_err gkm_set_err
#define set_err_msg gkm_set_err_msg
#define HIPCHK(expr)                                                        \
    do {                                                                    \
        hipError_t e_ = (expr);                                             \
        if (e_ != hipSuccess) return set_err(#expr, e_, __FILE__, __LINE__); \
    } while (0)

/* --------------------------------------------------------- pinned host pools (gkm_context.hip) */
constexpr int STAGE_SLOTS = 16; /* one pair of staging buffers per concurrently copying device thread */
int acquire_staging(size_t want, double **out, int slot);
struct PinBuf {
    char *p = nullptr;
    size_t cap = 0;
    std::atomic<int> in_use{0};
};
PinBuf *pin_acquire(size_t bytes);
void pin_release(void *ud);
void gkm_release_pipe_streams(); /* (the copy-out pipeline's cached streams, gkm_copyout.hip) */

/* ----------------------------------------------------------------- context */
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    /* headroom: per-launch scratch whose size drifts from launch to launch (the row blocks of the boundary
     * call) is allocated half as big again, because growing means hipFree, and hipFree waits for the device */
    int ensure(size_t count, bool headroom = false)
    {
        if (count <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = std::max<size_t>(headroom ? count + count / 2 : count, 1);
        HIPCHK(hipMalloc((void **)&p, want * sizeof(T)));
        cap = want;
        return 0;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct gkmhip_ctx {
    int device = 0;
    int L = 0, d = 0, rbf = 0, kernel_pref = GKMHIP_KERNEL_AUTO;
    double c[GKM_MAXD1] = {0};
    double gamma = 1.0;
    int n = 0, weighted = 0, maxlen = 0, minlen = 0;
    std::vector<int> h_len;
    std::vector<int64_t> h_lmoff;
    std::vector<double> h_cum_n; /* prefix sums of n_j = len_j - L + 1 */
    DevBuf<uint8_t> codes, wd; /* wd: distance-indexed positional weights */
    int wd_len = 0;
    DevBuf<uint32_t> wdc; /* the same weights CENTRED (byte wdc_centre + s = wd[|s|]): the row side of the several-pieces group variants */
    int wdc_words = 0, wdc_centre = 0;
    DevBuf<int64_t> off, lmoff;
    DevBuf<int> len;
    DevBuf<uint32_t> lmf, sb; /* lmf: forward l-mer table, then the reverse-strand table (general kernel only) */
    DevBuf<uint32_t> colpk;   /* 2-bit packed strands [seq][pkw][strand], the two strands interleaved word by word
                               * (k_pack_strands; gkm_bitslice.h pk_word): the hit path's column side */
    int pkw = 0;
    bool have_colpk = false;
    DevBuf<uint32_t> postab;  /* [seq][ptw] positional weights with zero guard bands (k_build_postab): the hit path's column
                               * weights in the one-piece variants */
    int ptw = 0;
    bool have_postab = false;
    uint32_t lm_stride = 0;
    int sb_xw = 0, sb_W = 0;
    bool have_lmers = false, have_sb = false;
    /* per-call scratch, two sets (gkmhip_set_scratch_slot): a caller that alternates launches between
     * two streams alternates the slot, so a launch never rewrites what the previous one still reads */
    struct Scratch {
        DevBuf<int> rows;
        DevBuf<int64_t> rowoff;    /* packed row offsets (general kernel only; the bit-sliced one has them in `tables`) */
        DevBuf<char> tables;       /* all per-launch tables of the bit-sliced kernel, one upload */
        DevBuf<uint32_t> rowplanes, rowpk;
        DevBuf<double> S;          /* tile-transposed raw values (k_gram_bitslice -> k_untile) */
        void release()
        {
            rows.release(); rowoff.release(); tables.release(); rowplanes.release(); rowpk.release(); S.release();
        }
    } scratch[GKM_SCRATCH_SLOTS];
    int sel = 0;
    DevBuf<double> sq;
    hipEvent_t ev0 = nullptr, ev1 = nullptr; /* around the Gram kernel of the most recent launch */
    bool ev_valid = false;
    /* gkmhip_kernel_timeline: while on, every launch records its own pair of events (no host wait in between), so
     * that a caller can time a loop of launches from outside and read the kernels' share of it afterwards */
    std::vector<std::pair<hipEvent_t, hipEvent_t>> tl_pairs;
    size_t tl_used = 0;
    bool tl_on = false;
    hipEvent_t last_e0 = nullptr, last_e1 = nullptr;
    double sampled_hit_share = -1.0; /* share of sampled l-mer pairs of THESE sequences within d mismatches (set_sequences) */
    double last_comparisons = 0;
    const char *last_kernel = "none";
};

/* the pair of events the next Gram kernel is bracketed by (gkm_context.hip) */
int gkm_launch_events(gkmhip_ctx *ctx, hipEvent_t *e0, hipEvent_t *e1);

constexpr int WD_LDS = 1024; /* distance weight table entries: >= max |n/2 - p| + 1 for n <= 2047 */

/* per-sequence device tables (gkm_context.hip): built by gkmhip_set_sequences, complete before any launch */
/* (wait: the table is complete on return, so that a launch on ANOTHER stream may read it -- what a launch that finds the
 * table missing needs; gkmhip_set_sequences builds all of them and waits once) */
int ensure_lmers(gkmhip_ctx *ctx, hipStream_t stream, bool wait = true);
int ensure_colpk(gkmhip_ctx *ctx, hipStream_t stream, bool wait = true);
int ensure_postab(gkmhip_ctx *ctx, hipStream_t stream, bool wait = true);
int ensure_sb(gkmhip_ctx *ctx, int W, hipStream_t stream, bool wait = true);
bool bitslice_serves(const gkmhip_ctx *ctx); /* which kernel this context's launches take (gkm_gram.hip) */

/* ------------------------------------------------------------ what a Gram launch writes */
struct GramOut {
    double *G;      /* raw values G(a,j); may be NULL when only `diag` is wanted */
    int64_t ld;
    int32_t *P;
    int64_t ldp;
    int local_rows;
    int write_all;  /* 0: only j <= a (lower triangle + diagonal); 1: every column visited */
    double *diag;   /* if set: diag[a] = G(a,a) */
    /* if set (packed row slabs, gkm_shard.h): local row r starts at G + row_off[r] instead of G + r * ld.
     * gram_launch() receives the HOST array and replaces it by its device copy. */
    const int64_t *row_off;
};

__device__ __forceinline__ double *gram_cell(const GramOut &out, int64_t r, int j)
{
    return out.G + (out.row_off ? out.row_off[r] : r * out.ld) + j;
}

/* which columns a tile of rows visits */
enum { COLS_TRIANGLE = 0, COLS_FULL = 1, COLS_DIAGONAL = 2 };

/* rows r0..r1-1 of a matrix whose rows < r1 hold raw values (gkm_normalize.hip) */
int normalize_rows(gkmhip_ctx *ctx, double *G, int64_t ld, int r0, int r1, double *sq, int symmetric, hipStream_t stream,
                   bool have_norms = false);

#endif
