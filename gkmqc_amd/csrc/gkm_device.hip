/*
 * gkm_device.hip -- MI355X (gfx950) device layer of the gkm kernel-matrix path.
 * Implements include/gkm_hip.h.  Written for CDNA4 only: 64-wide wavefronts, one
 * wavefront per workgroup in the hot kernels, column tables streamed through the
 * scalar unit (SGPRs), per-wave hit queues in LDS.
 *
 * Kernels
 *   k_build_sb        column-strand bit-plane tables, strided layout (gkm_bitslice.h)
 *   k_pack_strands    both strands of every sequence, 16 bases per word (the hit path's column side)
 *   k_build_rowplanes row-segment bit planes + the lanes' packed positions for one set of rows
 *   k_gram_bitslice   HOT: bit-sliced diagonal mismatch profile -> raw Gram values, tile-transposed
 *   k_untile          tile-transposed values -> matrix rows (both sides in 512-byte runs)
 *   k_pack_lmers, k_gram_direct   general fallback: per-l-mer tables, l-mer by l-mer XOR/popcount
 *   k_sqnorm, k_normalize, k_assemble_normalize   square roots of the diagonal, division, RBF, unit diagonal
 *   k_spin            2-ms busy kernel of the stream probe (gkmhip_create_stream_beside: do two streams share a queue?)
 */
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
#define GKM_SCRATCH_SLOTS 2 /* per-launch scratch sets of a context (gkmhip_set_scratch_slot) */

/* ------------------------------------------------------------------ errors */
static thread_local std::string g_err;

static int set_err(const char *what, hipError_t e, const char *file, int line)
{
    char buf[512];
    snprintf(buf, sizeof buf, "%s: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_err = buf;
    return 100 + (int)e;
}
static int set_err_msg(const std::string &m, int code)
{
    g_err = m;
    return code;
}
#define HIPCHK(expr)                                                        \
    do {                                                                    \
        hipError_t e_ = (expr);                                             \
        if (e_ != hipSuccess) return set_err(#expr, e_, __FILE__, __LINE__); \
    } while (0)

extern "C" const char *gkmhip_last_error(void) { return g_err.c_str(); }
/* (used by gkm_multi.hip so that one call reports every layer's failures) */
extern "C" void gkmhip_set_error_message(const char *msg) { g_err = msg ? msg : ""; }

extern "C" int gkmhip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int gkmhip_current_device(void)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    return dev;
}

extern "C" int gkmhip_set_current_device(int device)
{
    HIPCHK(hipSetDevice(device));
    return 0;
}

void gkm_release_pipe_streams(); /* (the copy-out pipeline's cached streams, defined with the pipeline below) */

/* Pinned staging for device-to-host copies, kept for the life of the process: the pipeline
 * calls the boundary once per peak subset (20x per run, bin/gkmqc.py:341-343) and pinning
 * 2 x 64 MB costs ~30 ms per call otherwise.  gkmhip_release_host_cache() frees it. */
constexpr int STAGE_SLOTS = 16; /* one pair of buffers per concurrently copying device thread */
static std::mutex g_stage_mutex;
static double *g_stage[STAGE_SLOTS][2];
static size_t g_stage_bytes[STAGE_SLOTS];

static int acquire_staging(size_t want, double **out, int slot)
{
    if (slot < 0 || slot >= STAGE_SLOTS) return set_err_msg("too many device threads", 2);
    std::lock_guard<std::mutex> lock(g_stage_mutex);
    if (g_stage_bytes[slot] < want) {
        for (int i = 0; i < 2; i++) {
            if (g_stage[slot][i]) (void)hipHostFree(g_stage[slot][i]);
            g_stage[slot][i] = nullptr;
        }
        g_stage_bytes[slot] = 0;
        for (int i = 0; i < 2; i++) HIPCHK(hipHostMalloc((void **)&g_stage[slot][i], want, hipHostMallocPortable));
        g_stage_bytes[slot] = want;
    }
    out[0] = g_stage[slot][0];
    out[1] = g_stage[slot][1];
    return 0;
}

/* Host side of the per-launch table uploads: a process-wide pool of pinned buffers, so that the upload is a true
 * asynchronous copy from memory that outlives the call.  A buffer is handed back by a host function enqueued on
 * the stream right behind the copy (hipLaunchHostFunc: it runs when the copy engine has finished reading), so
 * no thread ever waits for, or queries, an event of another thread's stream.  A pool, not one buffer per
 * context: the boundary call enqueues 13 launches up front, and waiting for the previous upload would make the
 * host follow the device launch by launch (measured: 81 ms of enqueueing instead of 2.5, the copy-out pipeline
 * starting only when the compute was over). */
struct PinBuf {
    char *p = nullptr;
    size_t cap = 0;
    std::atomic<int> in_use{0};
};
static std::mutex g_pin_mutex;
static std::vector<PinBuf *> g_pin;

static void pin_release(void *ud) { ((PinBuf *)ud)->in_use.store(0, std::memory_order_release); }

static PinBuf *pin_acquire(size_t bytes)
{
    PinBuf *b = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_pin_mutex);
        for (PinBuf *c : g_pin)
            if (c->in_use.load(std::memory_order_acquire) == 0 && (!b || (c->cap >= bytes && b->cap < bytes))) b = c;
        if (!b) {
            b = new PinBuf();
            g_pin.push_back(b);
        }
        b->in_use.store(1, std::memory_order_relaxed);
    }
    if (b->cap < bytes) { /* (this thread owns b now) */
        if (b->p) (void)hipHostFree(b->p);
        b->p = nullptr;
        b->cap = 0;
        const size_t want = std::max<size_t>(bytes + bytes / 2, (size_t)1 << 20);
        if (hipHostMalloc((void **)&b->p, want, hipHostMallocPortable) != hipSuccess) {
            b->in_use.store(0);
            return nullptr;
        }
        b->cap = want;
    }
    return b;
}

extern "C" void gkmhip_release_host_cache(void)
{
    {
        std::lock_guard<std::mutex> lock(g_pin_mutex);
        for (PinBuf *b : g_pin)
            if (b->in_use.load() == 0 && b->p) {
                (void)hipHostFree(b->p);
                b->p = nullptr;
                b->cap = 0;
            }
    }
    gkm_release_pipe_streams();
    std::lock_guard<std::mutex> lock(g_stage_mutex);
    for (int s = 0; s < STAGE_SLOTS; s++) {
        for (int i = 0; i < 2; i++) {
            if (g_stage[s][i]) (void)hipHostFree(g_stage[s][i]);
            g_stage[s][i] = nullptr;
        }
        g_stage_bytes[s] = 0;
    }
}

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
    int n = 0, weighted = 0, maxlen = 0;
    std::vector<int> h_len;
    std::vector<int64_t> h_lmoff;
    std::vector<double> h_cum_n; /* prefix sums of n_j = len_j - L + 1 */
    DevBuf<uint8_t> codes, wd; /* wd: distance-indexed positional weights */
    int wd_len = 0;
    DevBuf<int64_t> off, lmoff;
    DevBuf<int> len;
    DevBuf<uint32_t> lmf, sb; /* lmf: forward l-mer table, then the reverse-strand table (general kernel only) */
    DevBuf<uint32_t> colpk;   /* 2-bit packed strands [seq][pkw][strand], the two strands interleaved word by word
                               * (k_pack_strands; gkm_bitslice.h pk_word): the hit path's column side */
    int pkw = 0;
    bool have_colpk = false;
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
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_valid = false;
    double last_comparisons = 0;
    const char *last_kernel = "none";
};

extern "C" gkmhip_ctx *gkmhip_create(int device, int L, int d, const double *c, int rbf, double gamma)
{
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_err = "no HIP device available (hipGetDeviceCount)";
        return nullptr;
    }
    if (device < 0 || device >= ndev) { g_err = "device ordinal out of range"; return nullptr; }
    if (L < 2 || L > 12 || d < 0 || d > 12 || d > L) { g_err = "unsupported (L, d)"; return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { g_err = "hipSetDevice failed"; return nullptr; }
    gkmhip_ctx *ctx = new gkmhip_ctx();
    ctx->device = device;
    ctx->L = L;
    ctx->d = d;
    ctx->rbf = rbf;
    ctx->gamma = gamma;
    for (int m = 0; m <= d; m++) ctx->c[m] = c[m];
    if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        g_err = "hipEventCreate failed";
        delete ctx;
        return nullptr;
    }
    const char *env = getenv("GKM_KERNEL");
    if (env) {
        if (!strcmp(env, "direct")) ctx->kernel_pref = GKMHIP_KERNEL_DIRECT;
        else if (!strcmp(env, "bitslice")) ctx->kernel_pref = GKMHIP_KERNEL_BITSLICE;
    }
    return ctx;
}

extern "C" void gkmhip_destroy(gkmhip_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    /* (hipFree waits for the work that may still use the buffers; no separate device-wide wait) */
    ctx->codes.release(); ctx->wd.release(); ctx->off.release(); ctx->lmoff.release();
    ctx->len.release(); ctx->lmf.release(); ctx->sb.release(); ctx->colpk.release();
    for (auto &scr : ctx->scratch) scr.release();
    ctx->sq.release();
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    delete ctx;
}

extern "C" int gkmhip_n_sequences(const gkmhip_ctx *ctx) { return ctx ? ctx->n : 0; }
extern "C" int gkmhip_device_of(const gkmhip_ctx *ctx) { return ctx ? ctx->device : -1; }

extern "C" int gkmhip_set_scratch_slot(gkmhip_ctx *ctx, int slot)
{
    if (!ctx || slot < 0 || slot >= GKM_SCRATCH_SLOTS) return set_err_msg("bad scratch slot", 2);
    ctx->sel = slot;
    return 0;
}

extern "C" int gkmhip_set_kernel(gkmhip_ctx *ctx, int which)
{
    if (!ctx || which < 0 || which > 2) return set_err_msg("bad kernel selector", 2);
    ctx->kernel_pref = which;
    return 0;
}

/* ----------------------------------------------------------- prep kernels */
/* one workgroup per sequence: l-mer table entries (gkm_bitslice.h lmer_entry) of the forward
 * strand and of the reverse-complement strand (rc l-mer p = l-mer p of rc(seq),
 * libgkm.c:877-888), each with its positional weight in the top byte: wt[p] = wd[|n/2 - p|],
 * wt_rc[p] = wt[n-1-p] (libgkm.c:912-925); all weights 1 for the unweighted kernel types */
__global__ void k_pack_lmers(const uint8_t *__restrict__ codes, const int64_t *__restrict__ off,
                             const int64_t *__restrict__ lmoff, int L, const uint8_t *__restrict__ wd,
                             int weighted, uint32_t *__restrict__ lmf, uint32_t *__restrict__ lmr)
{
    const int s = blockIdx.x;
    const uint8_t *seq = codes + off[s];
    const int len = (int)(off[s + 1] - off[s]);
    const int n = len - L + 1;
    const int64_t o = lmoff[s];
    for (int p = threadIdx.x; p < n; p += blockDim.x) {
        const uint32_t wf = weighted ? gkmbs::dist_weight(wd, n / 2, p) : 1u;
        const uint32_t wr = weighted ? gkmbs::dist_weight(wd, n / 2, n - 1 - p) : 1u;
        lmf[o + p] = gkmbs::lmer_entry(seq, len, L, 0, p, wf);
        lmr[o + p] = gkmbs::lmer_entry(seq, len, L, 1, p, wr);
    }
}

/* grid (sequence*2+strand); threads over the words of the strand's 2-bit packed copy (gkm_bitslice.h pk_word).
 * The two strands of a sequence are interleaved word by word, colpk[(seq * pkw + x) * 2 + strand]: the hot kernel
 * copies the 2 * pkw words of a column to LDS as they are, and a hit reads words x and x + 1 of its strand at byte
 * offset (x * 8) | (strand * 4) -- the strand costs the address one OR instead of a multiply-add */
__global__ void k_pack_strands(const uint8_t *__restrict__ codes, const int64_t *__restrict__ off, int pkw,
                               uint32_t *__restrict__ colpk)
{
    const int e = blockIdx.x, s = e >> 1, strand = e & 1;
    const uint8_t *seq = codes + off[s];
    const int T = (int)(off[s + 1] - off[s]);
    for (int x = threadIdx.x; x < pkw; x += blockDim.x)
        colpk[((size_t)s * pkw + x) * 2 + strand] = gkmbs::pk_word(seq, T, strand, x);
}

/* grid (sequence*2+strand, plane); threads over words of the strand's SB table */
__global__ void k_build_sb(const uint8_t *__restrict__ codes, const int64_t *__restrict__ off, int W,
                           int L, int xw, uint32_t *__restrict__ sb)
{
    const int e = blockIdx.x, plane = blockIdx.y;
    const int s = e >> 1, strand = e & 1;
    const uint8_t *seq = codes + off[s];
    const int T = (int)(off[s + 1] - off[s]);
    uint32_t *dst = sb + ((size_t)e * 2 + plane) * xw; /* planes: 0 = hi bit, 1 = lo bit of the base code */
    for (int x = threadIdx.x; x < xw; x += blockDim.x)
        dst[x] = (x < T + W) ? gkmbs::sb_word(seq, T, strand, x, W, L, plane) : 0u;
}

/* Packed lanes (gkm_pack.h): grid (tile, plane); 64 threads = the tile's lanes; output layout
 * [tile][plane][w][lane].  desc holds MAX_PIECES x {row, b0, nb, p0, cnt} per lane (nb = 0: unused).
 * plane 3: the lane's positions 2-bit packed for the hit path, rowpk[(tile*64 + lane) * rpw + x]
 * (16 positions per word, position i = bit row i / W, word i % W of the bit planes; rpw = 32 words = 128 bytes
 * per lane, of which 32 W / 16 + 1 are used). */
__global__ void k_build_rowplanes(const uint8_t *__restrict__ codes, const int64_t *__restrict__ off,
                                  const int *__restrict__ desc, int W, uint32_t *__restrict__ planes,
                                  uint32_t *__restrict__ rowpk, int rpw)
{
    /* grid (tile, plane, part): planes 0..2 one word w = part per block (part < W), plane 3 four packed words per block
     * (round 4: one block per (tile, plane) looped over all of them -- 0.7 ms per 10 000 rows, 1 % of a step) */
    const int tile = blockIdx.x, plane = blockIdx.y, part = blockIdx.z, lane = threadIdx.x;
    const int *d = desc + (size_t)(tile * 64 + lane) * gkmpack::MAX_PIECES * 5;
    if (plane == 3) {
        for (int x = part * 4; x < rpw && x < part * 4 + 4; x++) {
            uint32_t v = 0u;
            for (int k = 0; k < gkmpack::MAX_PIECES && x * 16 < 32 * W; k++) { /* (words past the lane's positions: 0) */
                const int row = d[k * 5 + 0], b0 = d[k * 5 + 1], nb = d[k * 5 + 2], p0 = d[k * 5 + 3];
                if (nb <= 0) continue;
                const uint8_t *seq = codes + off[row];
                const int len = (int)(off[row + 1] - off[row]);
                for (int q = 0; q < 16; q++) {
                    const int i = x * 16 + q, b = i / W;
                    if (b < b0 || b >= b0 + nb) continue;
                    const int pos = p0 + i - b0 * W;
                    if (pos < len) v |= (uint32_t)seq[pos] << (2 * q);
                }
            }
            rowpk[(size_t)(tile * 64 + lane) * rpw + x] = v;
        }
        return;
    }
    for (int w = part; w < W; w += (int)gridDim.z) {
        uint32_t v = 0u;
        for (int k = 0; k < gkmpack::MAX_PIECES; k++) {
            const int row = d[k * 5 + 0], b0 = d[k * 5 + 1], nb = d[k * 5 + 2], p0 = d[k * 5 + 3], cnt = d[k * 5 + 4];
            if (nb <= 0) continue;
            const uint8_t *seq = codes + off[row];
            const int len = (int)(off[row + 1] - off[row]);
            for (int b = b0; b < b0 + nb; b++) v |= gkmbs::piece_bit(seq, len, b0, nb, p0, cnt, b, w, W, plane) << b;
        }
        planes[(((size_t)tile * 3 + plane) * W + w) * 64 + lane] = v;
    }
}

/* ------------------------------------------------------------ hot kernels */
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

struct BsArgs {
    const uint32_t *rowplanes;  /* [tile][plane 3][W][64] */
    const uint32_t *lane_mask;  /* [tile*64] bit rows at which a piece starts */
    const uint32_t *lane_piece; /* [tile*64][MAX_PIECES][2]: row slot, l-mer table base of the piece */
    const int *tile_row, *tile_out, *tile_nrows, *tile_cbeg, *tile_cend; /* columns [cbeg, cend) per tile */
    const uint32_t *rowpk;      /* [tile*64 + lane][rpw] the lanes' positions, 2-bit packed (k_build_rowplanes) */
    const uint32_t *colpk;      /* [seq][pkw][strand] 2-bit packed strands, the two strands interleaved        */
    const uint32_t *wd32;       /* distance-indexed positional weights (bytes), wd_words dwords               */
    int rpw, pkw, wd_words;
    const uint32_t *sb;
    int xw;
    const int *len;
    double c[GKM_MAXD1];
    GramOut out;
    /* Work items = (tile, column) pairs, one wavefront each, as a 1-D grid of exactly the pairs inside the
     * visited region (the 2-D (column, tile) grid launched as many empty blocks as real ones): item =
     * tile_soff[tile] + (j - cbeg[tile]), columns fastest.  Neighbouring blocks -- the waves resident on
     * a CU at the same time -- therefore work on the SAME row tile (its packed rows stay in the CU's L1)
     * and on DIFFERENT columns.  The opposite order (all tiles of a column next to each other on one XCD,
     * so that the column tables come from that XCD's L2) was measured: 91 instead of 85 ms on config 2
     * and 1000 instead of 509 ms on the peak-like set -- the waves of a CU then hit the same dense column
     * regions at the same moment and all wait on the hit path together. */
    int ntiles;
    /* raw Gram values leave the kernel tile-transposed: S[(tile_soff[tile] + j - cbeg) * NSLOT + row slot],
     * 64 consecutive doubles per store instruction; k_untile turns them into rows of G */
    double *S;
    const int64_t *tile_soff;
    /* The ORDER of the work items (round 4): entries (column chunk, tile), chunks outermost -- all tiles take the columns
     * [c C, (c + 1) C) before any takes the next chunk, so that a chunk's column tables (SB planes: 5-10 KB per column)
     * are streamed from HBM once per chunk and then come from the XCD's L2 for every further tile, instead of once per
     * tile.  Entry e covers columns [ent_j0[e], ent_j1[e]) of tile ent_tile[e] and starts at work item ent_off[e]; every
     * entry's item count is rounded up to a multiple of 8 (padding items return at once), so that column j of a chunk
     * has the same index mod 8 -- the same XCD under round-robin placement -- for every tile.  Inside an entry the
     * order is what it always was: one tile, consecutive columns.  nent = 0: the plain tile-major order. */
    int nent;
    const int64_t *ent_off;
    const int *ent_tile, *ent_j0, *ent_j1;
};

constexpr int WD_LDS = 1024; /* distance weight table entries: >= max |n/2 - p| + 1 for n <= 2047 */

typedef const uint32_t __attribute__((address_space(4))) * sgpr_words;

/* wave64 inclusive prefix sum on the DPP network (no LDS round trips): four row_shr steps
 * scan each row of 16 lanes, row_bcast:15 / row_bcast:31 carry the row totals across */
__device__ __forceinline__ int wave_inclusive_scan(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true); /* row_shr:1 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true); /* row_shr:2 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true); /* row_shr:4 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true); /* row_shr:8 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false); /* row_bcast:15 -> rows 1,3 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false); /* row_bcast:31 -> rows 2,3 */
    return x;
}

/* position of the lowest set bit, 0xFFFFFFFF for 0 (v_ffbl_b32's own convention; __builtin_ctz(0) is
 * undefined and the generic cttz costs a second instruction) */
__device__ __forceinline__ uint32_t ffbl_or_ones(uint32_t x)
{
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

/* 2 x as an addition: on gfx950 v_lshlrev_b32 issues at HALF the rate of v_add_u32 (tools/valu_ops.hip), and hipcc
 * turns x + x back into a shift */
__device__ __forceinline__ uint32_t twice(uint32_t x)
{
    uint32_t r;
    asm("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(x));
    return r;
}

/* popcount(x) + acc in one instruction (hipcc sums separate popcounts with extra adds) */
__device__ __forceinline__ uint32_t popc_add(uint32_t x, uint32_t acc)
{
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

#ifndef GKM_BS_GRP
#define GKM_BS_GRP 5 /* config 2 / gkmQC's default L=10 k=6 d=3: 2 -> 89.8 / 118.1 ms, 5 -> 81.0 / 119.3, 10 -> 87.5 / 138.3 */
#endif
constexpr int BS_GRP = GKM_BS_GRP;     /* hit words per list record: the lanes are compacted once per BS_GRP words */
#ifndef GKM_BS_TRIP
#define GKM_BS_TRIP 64 /* config 2: 64 -> 87.2 ms (ring of 128: index wrap is one AND), 128 -> 89.1, 192 -> 97.3 */
#endif
constexpr int BS_TRIP = GKM_BS_TRIP; /* records resolved per trip: one per lane */
static_assert(BS_TRIP == 64, "a trip resolves one record per lane");
/* records the wave-wide hit list holds: >= BS_TRIP + 64, a multiple of 64 (merged LDS stores) */
constexpr int BS_CAP = BS_TRIP + 64;
#ifndef GKM_BS_DU
#define GKM_BS_DU 4 /* shifts per refill of the column words.  Round 2's final kernel, same-run A/B, config 2 / gkmQC's
                       defaults / config 5: 1 -> 78.2 / 437.2 / 177.9 ms, 2 -> 77.0-77.5 / 436.8-437.6 / 176.7, 3 -> 76.7 / 437.0 /
                       176.2, 4 -> 76.4-76.8 / 436.7-438.6 / 175.6-175.8, 5, 6, 8 -> 79.1-79.4 / 452-454 / 176.5 */
#endif
#ifndef GKM_BS_WAVES
#define GKM_BS_WAVES 6 /* waves per SIMD asked of the compiler for the one-piece-per-lane kernel (<= 80 VGPRs):
                          config 2 with the grouped hit ring: 5 -> 96.1 ms, 6 -> 91.8, 7 (spills) -> 97.5 */
#endif
#ifndef GKM_BS_PACKED_WAVES
#define GKM_BS_PACKED_WAVES 6 /* several-pieces-per-lane kernels: 75-78 VGPRs */
#endif
constexpr int BS_DU = GKM_BS_DU; /* shifts per SB register refill */
#ifndef GKM_TRIP_PRIO
#define GKM_TRIP_PRIO 3 /* wave priority (s_setprio, 0..3) inside a trip; 0 = as rounds 1-3 */
#endif

/*
 * One wavefront = 64 row segments (one per lane) x ONE column sequence.
 * For both strands of the column the wave sweeps all T cyclic shifts; per shift each lane
 * evaluates 32*W l-mer window comparisons with ~13 VALU instructions per 32 (gkm_bitslice.h).
 * Hit words are parked, compacted over the lanes, in a wave-wide LDS list (a stack) of records and turned
 * into weighted profile counts in full-wave batches, so the hot loop has no data-dependent
 * control flow besides the push.
 */
template <int W, int L, int D, int PK>
__global__ __launch_bounds__(64, D > 4 ? 1 : (PK == 1 || PK == 2) ? GKM_BS_PACKED_WAVES : GKM_BS_WAVES) void k_gram_bitslice(const BsArgs A)
{
    /* PK = 0: one piece per lane, up to 64 rows per tile (every fixed-length data set);
     *      1: several pieces per lane (gkm_pack.h), up to 64 rows per tile; 2: up to 128 rows per tile;
     *      3: as 0, but a trip fetches the source lane's piece entry from that lane's registers (ds_bpermute_b32)
     *         instead of a 512-byte table in LDS -- taken when those 512 bytes cost an LDS allocation granule,
     *         i.e. waves per CU (rows and columns of 600 bp: 24 -> 32 one-wave workgroups per CU by LDS, 449 -> 436 ms
     *         on gkmQC's defaults); where they do not, the table is 0.5 % faster (config 2: 76.7 vs 77.1 ms) */
    constexpr bool PACKED = PK == 1 || PK == 2;
    constexpr bool BPERM = PK == 3;
    /* (The ablation builds of rounds 1-3 -- parts of this kernel skipped to time the rest, results wrong -- lived
     * here as a fifth template parameter; they are gone from the source since round 4.  tools/variants.sh rebuilds
     * them from revision a4bed73, profiles/r2_ablation_timings.txt and r2_pmc_ablation_builds*.txt hold what they
     * measured.) */
    using namespace gkmbs;
    /* LDS per wave: 3 KB hit list + 0-1.3 KB piece table + 1-2.5 KB accumulators + (dynamic) the column's two
     * 2-bit packed strands and the distance-indexed weight table, 0.4 KB at 300 bp, 0.7 KB at 600 bp.  What the
     * hit resolution reads per hit: two words of the column strand and two weight bytes from LDS, two words
     * of the row lane's packed positions from global memory (8 KB per tile -- 128 bytes per lane, of which 84 are
     * used -- L1 resident: the waves of a CU work on the same tile). */
    extern __shared__ uint32_t s_dyn[]; /* [wd_words] weight bytes, then [2 * pkw] column strands (forward, reverse complement) */
    /* The hit list.  A record is the BS_GRP hit words of one lane for BS_GRP consecutive words of a
     * shift plus their origin; word k of record i sits at s_list[k * BS_CAP + i], the origin at
     * k = BS_GRP (arrays a multiple of 64 dwords apart: the stores of a push merge into
     * ds_write2st64_b32).  Compacting once per group instead of once per word takes 3 VALU
     * instructions per word out of the hot loop (config 2: 111.0 -> 96.2 ms). */
    __shared__ uint32_t s_list[(BS_GRP + 1) * BS_CAP];
    /* (array BS_GRP of s_list: first word of the group, delta, strand, row lane) */
    /* PACKED: lanes may hold several pieces (gkm_pack.h).  When no lane of the call holds more than one
     * piece (e.g. every fixed-length data set) the leaner variant runs: one (slot, centre) pair per lane.
     * The several-pieces variant exists for 64 and for 128 row slots per tile: the profiles of 128 slots
     * (2.5 KB at d = 4) cost a wave per SIMD, so the host packs at most 64 rows into a tile unless
     * that would leave lanes empty (many rows shorter than half a lane).  LDS per wave, d = 4, 600 bp:
     * 3 KB ring + 0.25 KB piece starts + 1 KB piece table + 1.25 KB profiles + 0.6 KB strands and weights
     * = 6.1 KB -> 6 waves per SIMD (8.4 KB -> 4.75 with 128 slots and two-word piece entries). */
    constexpr int NP = PACKED ? gkmpack::MAX_PIECES : 1;   /* pieces per lane */
    constexpr int NSLOT = PK == 2 ? gkmpack::MAX_ROWS : 64; /* row slots per tile */
    __shared__ uint32_t lmask[PACKED ? 64 : 1];  /* piece-start bit rows of every lane         */
    /* per piece: row slot * 4 and the biased centre offset c0 + 2048 -- two words in the one-piece variant
     * (one ds_read_b64), one word (slot * 4 | c0b << 16) in the several-pieces variants */
    __shared__ uint32_t lpiece[PACKED ? 64 * NP : BPERM ? 1 : 128];
    __shared__ uint32_t accl[(D + 1) * NSLOT];   /* mismatch profiles [m][row slot]            */
    static_assert(W % BS_GRP == 0, "a shift is a whole number of record groups");
    /* The list is a STACK (round 3; a ring before): a trip is due as soon as it holds BS_TRIP records and it is checked
     * after every group (at most 64 new records); a trip takes the BS_TRIP records on TOP and puts at most as many back:
     * the list never holds more than BS_TRIP + 63 records.  The order in which hits are resolved is immaterial
     * (integer adds), and a stack needs no head and no wrap: one AND less per push, per trip and per re-push, and the
     * trip's read address is lane * 4 + a scalar. */
    static_assert(BS_CAP >= BS_TRIP + 64 && BS_CAP % 64 == 0, "hit list too small");
    static_assert(gkmpack::MAX_ROWS % 64 == 0, "row slots are finished 64 at a time");

    const int lane = threadIdx.x;
    /* (raising the priority of a NEW wave too, until its row planes are loaded, was measured: 395.4 against 388.9 ms on
     * gkmQC's shape, nothing on config 2 -- profiles/r4_kernel_ab_trip_priority.txt) */
    /* block -> (tile, column): see BsArgs.  All of this is wave-uniform (scalar loads, SALU). */
    int tile, j0;
    if (A.nent > 0) { /* (chunk, tile) entries: largest e with ent_off[e] <= blockIdx.x */
        int lo = 0, hi = A.nent;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (A.ent_off[mid] <= (int64_t)blockIdx.x) lo = mid;
            else hi = mid;
        }
        tile = A.ent_tile[lo];
        j0 = A.ent_j0[lo] + (int)((int64_t)blockIdx.x - A.ent_off[lo]);
        if (j0 >= A.ent_j1[lo]) return; /* padding item */
    } else {
        int lo = 0, hi = A.ntiles; /* largest tile with tile_soff[tile] <= blockIdx.x */
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (A.tile_soff[mid] <= (int64_t)blockIdx.x) lo = mid;
            else hi = mid;
        }
        tile = lo;
        j0 = A.tile_cbeg[tile] + (int)((int64_t)blockIdx.x - A.tile_soff[tile]);
    }
    const int j1 = j0 + 1;
    const int nrows = A.tile_nrows[tile];
    constexpr int NE = NSLOT / 64; /* row slots a lane finishes in the epilogue */
    /* A tile with at most NSLOT / 2 rows (600-bp rows: 32 per tile) keeps TWO copies of every profile, NSLOT / 2 slots
     * apart, and the epilogue adds them: the hits of odd source lanes go to the second copy (the host puts the offset
     * into those lanes' piece entries, gram_launch), so the ds_add_u32 of a trip spread over twice the addresses --
     * 88 % of all hits have m = d and 64 lanes were adding into 32 words. */
    const bool two_copies = 2 * nrows <= NSLOT;

    uint32_t Ahi[W], Alo[W], AV[W];
#pragma unroll
    for (int w = 0; w < W; w++) {
        Ahi[w] = A.rowplanes[(((size_t)tile * 3 + 0) * W + w) * 64 + lane];
        Alo[w] = A.rowplanes[(((size_t)tile * 3 + 1) * W + w) * 64 + lane];
        AV[w] = A.rowplanes[(((size_t)tile * 3 + 2) * W + w) * 64 + lane];
    }

    /* one wavefront per workgroup: the LDS traffic of a wave is ordered, no barriers needed */
    if (PACKED) lmask[lane] = A.lane_mask[tile * 64 + lane];
    constexpr int LPW = PACKED ? NP : 2; /* lpiece words per lane */
    if (!BPERM) {
#pragma unroll
        for (int k = 0; k < LPW; k++) lpiece[lane * LPW + k] = A.lane_piece[(size_t)(tile * 64 + lane) * LPW + k];
    }
    /* BPERM: the lane keeps its own (row slot * 4, biased centre offset) in two registers and a trip fetches
     * the source lane's pair over the permute network (ds_bpermute_b32: no LDS storage, no bank conflicts).
     * The 512 bytes this takes out of LDS bring a wave under 5 120 bytes = 4 allocation granules of 1 280
     * (tools/lds_occupancy.hip): 32 instead of 24 one-wave workgroups fit a CU at 600 bp. */
    uint32_t my_both = 0u; /* row slot * 4 (< 256) | biased centre offset (< 8192) << 16 */
    if (BPERM)
        my_both = A.lane_piece[(size_t)(tile * 64 + lane) * 2] | (A.lane_piece[(size_t)(tile * 64 + lane) * 2 + 1] << 16);
    const uint32_t lane_tag = (uint32_t)lane << META_LANE_SHIFT, lane4 = (uint32_t)lane << 2;
    const int pkw = A.pkw;
    /* dynamic LDS: the column's two packed strands first, interleaved word by word (their address is then a constant
     * of the kernel and folds into the offset field of the reads), the weight table behind them (its offset rides in
     * the third operand of the v_sad_u32 that forms the index) */
    uint32_t *const s_col = s_dyn;
    /* byte offset of the weight table, kept in a VGPR: the column-side index |q - centre| + wbase would otherwise name
     * two SGPRs in one v_sad_u32 (one is the limit) and cost a v_mov per hit */
    uint32_t wbase;
    asm volatile("v_mov_b32 %0, %1" : "=v"(wbase) : "s"((uint32_t)pkw * 8u));
    for (int x = lane; x < A.wd_words; x += 64) s_dyn[2 * pkw + x] = A.wd32[x];
    /* this tile's packed lanes: 32-bit byte offsets from a wave-uniform base (global_load with an SGPR
     * base instead of a 64-bit address computed per lane); 128 bytes per lane, so that the lane field of a
     * record's origin word IS the lane's byte offset */
    const char *const rowpk_tile = (const char *)(A.rowpk + (size_t)tile * 64 * A.rpw);

    for (int j = j0; j < j1; j++) {
        const int T = A.len[j];
        const int nB = T - L + 1;
        const uint32_t rcpT = mod_magic((uint32_t)T);
        for (int x = lane; x < 2 * pkw; x += 64) s_col[x] = A.colpk[(size_t)j * 2 * pkw + x];
        /* weight of a column l-mer q: forward strand wd[|nB/2 - q|]; reverse strand wt_rc[q] = wt[nB-1-q]
         * (libgkm.c:924) = wd[|nB/2 - (nB-1-q)|] = wd[|q + [nB even] - nB/2|]; the [nB even] of the reverse strand
         * travels in bit 5 of the record's origin word (pack_meta) */
        const uint32_t ccen = (uint32_t)(nB / 2);
        const int ceven = (nB & 1) ? 0 : 1;
#pragma unroll
        for (int m = 0; m <= D; m++)
            for (int rs = lane; rs < (two_copies ? NSLOT : nrows); rs += 64) accl[m * NSLOT + rs] = 0u;
        int s_n = 0; /* records in the hit list (wave-uniform) */

        /* One hit -> accl[m][row slot] += wa * wb.  (meta + sel, bit) name the row lane r, the lane position
         * i0 = bit*W + w of the window, the shift and the strand; lane and bit row name the piece (gkm_pack.h),
         * the piece names the row slot and c0, which makes |c0 - i0| the row l-mer's distance to its sequence's
         * centre l-mer (libgkm.c:912-925 depends on nothing else).  Written for the ISSUE COST -- the kernel is
         * bound by VALU issue, and on gfx950 only the plain two-operand integer operations and v_bitop3_b32 issue
         * at the full rate; v_bfe, v_mad_u32_u24, v_min, v_sad, v_alignbit, v_ffbl, v_bcnt, compares, SDWA and
         * anything with an SGPR operand take twice as long (tools/valu_ops.hip).  Hence the layout of the origin
         * word (gkm_bitslice.h pack_meta: fields that are masked in place or shifted out of the top), 128 bytes per
         * lane of packed positions, the column's strands interleaved word by word, (a & const) | b as one
         * v_bitop3_b32, the weight table's LDS offset as the third operand of the v_sad_u32 that forms the index.
         * Same arithmetic as resolve_hit_packed (gkm_bitslice.h), which the CPU tests run against the oracle. */
        auto resolve = [&](uint32_t ms, uint32_t bit, uint32_t pslot4, uint32_t pc0b) {
            const uint32_t lane128 = ms & (63u << META_LANE_SHIFT); /* source lane * 128 */
            const int k = PACKED ? piece_of_bitrow(*(const uint32_t *)((const char *)lmask + (PACKED ? (lane128 >> 5) : 0u)), (int)bit) : 0;
            uint32_t slot4, c0b; /* row slot * 4; (l-mers of the row) / 2 - p0 + b0*W + 2048 */
            if (PACKED) {
                static_assert(!PACKED || NP == 4, "lpiece is addressed as lane * 16 + piece * 4");
                const uint32_t lp = *(const uint32_t *)((const char *)lpiece + ((lane128 >> 3) + ((uint32_t)k << 2)));
                slot4 = lp & 0xFFFFu;
                c0b = lp >> 16;
            } else if (BPERM) {
                slot4 = pslot4;
                c0b = pc0b;
            } else {
                const uint32_t *lp2 = (const uint32_t *)((const char *)lpiece + (lane128 >> 4));
                slot4 = lp2[0];
                c0b = lp2[1];
            }
            const uint32_t i0 = __umul24(bit, (uint32_t)W) + (ms & 15u);
            const uint32_t x = i0 + (ms >> 21);
            uint32_t q;
            if ((uint32_t)T >= (uint32_t)(32 * W)) q = min(x - (uint32_t)T, x); /* x < 2T (wave-uniform test) */
            else q = mod_small(x, (uint32_t)T, rcpT);
            /* a window that wraps around the end of the strand is not an l-mer (gkm_bitslice.h window_hits) */
            if ((int)q < nB) {
                /* (a & -4) | b and (a & -8) | b as ONE v_bitop3_b32 each (truth table 0xEA), inline constants */
                const uint32_t *rw = (const uint32_t *)(rowpk_tile + lop3<0xEA>(i0 >> 2, ~3u, lane128));
                const uint32_t *cw = (const uint32_t *)((const char *)s_col + lop3<0xEA>(q >> 1, ~7u, (ms >> 2) & 4u));
                const uint8_t *wdb = (const uint8_t *)s_dyn;
                const uint32_t wa = wdb[__usad(c0b, i0 | 2048u, wbase)];
                const uint32_t wb = wdb[__usad(q + ((ms >> 5) & 1u), ccen, wbase)];
                /* (v_alignbit_b32 uses the low 5 bits of its count: 2 i0 mod 32 = 2 (i0 mod 16)) */
                const uint32_t ea = __builtin_amdgcn_alignbit(rw[1], rw[0], twice(i0));
                const uint32_t eb = __builtin_amdgcn_alignbit(cw[2], cw[0], twice(q));
                const uint32_t m = (uint32_t)pk_mismatch(ea, eb, L);
                if (m <= (uint32_t)D) /* LDS atomic: ds_add_u32 */
                    atomicAdd((uint32_t *)((char *)accl + (m * (uint32_t)(NSLOT * 4) + slot4)), wa * wb);
            }
        };

        /* one trip over the `c` records on top of the list (PARTIAL: c < BS_TRIP, the last trip of a column) */
        auto trip = [&](auto partial_tag, int c) {
            constexpr bool PARTIAL = decltype(partial_tag)::value;
            /* A wave inside a trip issues AHEAD of the waves that are in the counting loop (s_setprio; back to 0 at the
             * end of the trip).  A trip is a chain of short instruction runs between LDS and memory round trips (record
             * -> piece entry -> row words -> column words and weights -> accumulate); at equal priority each run waits
             * its turn behind six waves of straight-line counting code, and the chain -- with the LDS list and the other
             * lanes' hits waiting on it -- stretches.  Round 4, same-run A/B (profiles/r4_kernel_ab_trip_priority.txt):
             * config 2 75.2 -> 72.8 ms, gkmQC's own shape 433.3 -> 396.0 ms, config 5 167.4 -> 152.7 ms; priority 1 and
             * 3 do the same.  The total VALU work is unchanged: this is issue ORDER, not instruction count. */
            __builtin_amdgcn_s_setprio(GKM_TRIP_PRIO);
            /* the c records on top: lane * 4 + a scalar (kept apart from the lane term: hipcc would fuse the shift into a
             * half-rate v_lshl_add_u32 and split the reads around a negative offset) */
            const uint32_t top4 = (uint32_t)__builtin_amdgcn_readfirstlane((s_n - c) << 2);
            uint32_t at_off;
            asm("v_add_u32 %0, %1, %2" : "=v"(at_off) : "s"(top4), "v"(lane4));
            const char *const at = (const char *)s_list + at_off;
            uint32_t h[BS_GRP];
            /* (every ring slot is readable: the lanes past the end of a short, final trip are
             * cleared afterwards instead of being masked out of the loads) */
#pragma unroll
            for (int g = 0; g < BS_GRP; g++) h[g] = *(const uint32_t *)(at + g * BS_CAP * 4);
            const uint32_t meta = *(const uint32_t *)(at + BS_GRP * BS_CAP * 4);
            if (PARTIAL) {
#pragma unroll
                for (int g = 0; g < BS_GRP; g++) h[g] = (lane < c) ? h[g] : 0u;
            }
            uint32_t first = ffbl_or_ones(h[0]), total = 0u;
#pragma unroll
            for (int g = 1; g < BS_GRP; g++) first = min(first, ffbl_or_ones(h[g]) | (uint32_t)(g << 5));
#pragma unroll
            for (int g = 0; g < BS_GRP; g++) total = popc_add(h[g], total);
            const uint32_t sel = first >> 5, bit = first & 31u;
            const uint32_t ms = meta + sel; /* the word index w0 + sel <= W - 1 stays inside its 4 bits */
            uint32_t pslot4 = 0u, pc0b = 0u;
            if (BPERM) { /* every lane takes part (ds_bpermute_b32 reads 0 from lanes that EXEC masks out) */
                const int from = (int)((ms >> (META_LANE_SHIFT - 2)) & 0xFCu); /* source lane * 4 */
                /* ONE permute of (slot * 4 | c0b << 16) and two full-rate VALU operations to take it apart, not two
                 * permutes: the LDS pipe is busy two thirds of the time on gkmQC's shape (SQ_LDS_IDX_ACTIVE per CU against
                 * the kernel's cycles, profiles/r4_pmc_peaks.json): 392.6 -> 388.6 ms (profiles/r4_kernel_ab_trip_priority.txt) */
                const uint32_t both = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)my_both);
                pslot4 = both & 0xFFFFu;
                pc0b = both >> 16;
            }
            /* every record of a full trip holds a hit (only records with one are pushed or pushed again): no test */
            if (!PARTIAL || total) resolve(ms, bit, pslot4, pc0b);
            s_n -= c;
            const unsigned long long more = __ballot(total > 1u);
            if (more) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(more >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((uint32_t)more, 0u));
                if (total > 1u) {
                    char *const to = (char *)s_list + ((rank + (uint32_t)s_n) << 2);
#pragma unroll
                    for (int g = 0; g < BS_GRP; g++) *(uint32_t *)(to + g * BS_CAP * 4) = h[g];
                    *(uint32_t *)(to + BS_GRP * BS_CAP * 4) = meta;
                    atomicXor((uint32_t *)(to + sel * (uint32_t)(BS_CAP * 4)), 1u << bit); /* ds_xor_b32: that hit is done */
                }
                s_n += (int)__popcll(more);
            }
            __builtin_amdgcn_s_setprio(0);
        };
        /* Resolve the hit list in FULL trips of 64 records with every lane busy: each record gives up
         * its first hit (lowest bit of its first non-empty word), a record with more hits is appended
         * again.  Fewer than one trip's worth of records waits in the list; the last call of a column
         * (final) empties it.
         * No select chains: the position of the first hit is min over the words of ffbl(word) | 32 g
         * (v_ffbl_b32 gives all ones for an empty word, so empty words lose the min), the number of
         * hits left is a popcount sum, and a record that goes back to the list is copied unchanged and
         * then loses that hit by ONE LDS xor on the copy (the LDS operations of a wave execute in order). */
        auto trips = [&](bool final) {
            while (s_n >= BS_TRIP) trip(std::false_type(), BS_TRIP);
            if (final)
                while (s_n > 0) {
                    if (s_n >= BS_TRIP) trip(std::false_type(), BS_TRIP);
                    else trip(std::true_type(), s_n);
                }
        };

        for (int strand = 0; strand < 2; strand++) {
            /* read-only, wave-uniform: address space 4 makes hipcc fetch these words with
             * scalar loads (s_load_dwordx*) into SGPRs instead of per-lane vector loads */
            const sgpr_words sbh = (sgpr_words)(A.sb + ((size_t)(j * 2 + strand) * 2) * A.xw);
            const sgpr_words sbl = sbh + A.xw;
            for (int d0 = 0; d0 < T; d0 += BS_DU) {
                /* (copying the words to VGPRs once instead of using them as SGPR operands was measured
                 * slower: 119-129 ms against 111.6 ms on config 2; requesting the next block's words one
                 * block ahead changes nothing: 92.3 against 92.4 ms) */
                /* the strand's window-validity plane (third SB plane) is not streamed: wrapped
                 * windows are rejected when a hit is resolved (gkm_bitslice.h window_hits) */
                uint32_t bh[BS_DU + W - 1], bl[BS_DU + W - 1];
#pragma unroll
                for (int i = 0; i < BS_DU + W - 1; i++) {
                    bh[i] = sbh[d0 + i];
                    bl[i] = sbl[d0 + i];
                }
#pragma unroll
                for (int u = 0; u < BS_DU; u++) {
                    if (d0 + u < T) {
                        uint32_t hit[W];
                        window_hits<W, L, D>(Ahi, Alo, AV, bh + u, bl + u, (const uint32_t *)nullptr, hit);
                        const uint32_t vbase = lane_tag | pack_meta(d0 + u, 0, strand, ceven);
#pragma unroll
                        for (int w0 = 0; w0 < W; w0 += BS_GRP) {
                            /* wave-level compaction at the source, once per group of BS_GRP words: the
                             * lanes with a hit in the group append (words, origin) to the list at tail
                             * + their rank among the hit lanes (ballot + mbcnt); EXEC-masked stores, no
                             * divergent control flow */
                            uint32_t any = hit[w0];
#pragma unroll
                            for (int g = 1; g + 1 < BS_GRP; g += 2) any = lop3<TT_OR3>(any, hit[w0 + g], hit[w0 + g + 1]);
                            if (BS_GRP % 2 == 0) any |= hit[w0 + BS_GRP - 1];
                            const unsigned long long mask = __ballot(any != 0u);
                            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                            __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                            if (any != 0u) {
                                char *const at = (char *)s_list + (((uint32_t)rank + (uint32_t)s_n) << 2);
#pragma unroll
                                for (int g = 0; g < BS_GRP; g++) *(uint32_t *)(at + g * BS_CAP * 4) = hit[w0 + g];
                                *(uint32_t *)(at + BS_GRP * BS_CAP * 4) = vbase | (uint32_t)w0;
                            }
                            s_n += (int)__popcll(mask);
                            if (s_n >= BS_TRIP) trips(false);
                        }
                    }
                }
            }
        }
        trips(true);

        /* epilogue: one lane per row slot of the tile */
#pragma unroll
        for (int k = 0; k < NE; k++) {
            /* (row and output row of the slot are read here, not kept in registers through the sweep) */
            const int rs = k * 64 + lane;
            const int row = rs < nrows ? A.tile_row[tile * gkmpack::MAX_ROWS + rs] : -1;
            if (row < 0 || (j > row && !A.out.write_all)) continue;
            /* the profile: both copies where there are two (uint32 addition: the int32 wrap-around of the
             * reference's accumulator, libgkm.c:338, is kept) */
            uint32_t prof[D + 1];
#pragma unroll
            for (int m = 0; m <= D; m++) prof[m] = accl[m * NSLOT + rs] + (two_copies ? accl[m * NSLOT + rs + NSLOT / 2] : 0u);
            /* sum_m c_m P_m in ascending m from 0.0 (libgkm.c:576-582) */
            double g = 0.0;
#pragma unroll
            for (int m = 0; m <= D; m++) g += A.c[m] * (double)(int32_t)prof[m];
            const int64_t r = A.out.local_rows ? A.tile_out[tile * gkmpack::MAX_ROWS + rs] : row;
            if (A.out.diag && j == row) A.out.diag[row] = g;
            if (A.S) A.S[(A.tile_soff[tile] + (j - A.tile_cbeg[tile])) * NSLOT + rs] = g;
            if (A.out.P) {
#pragma unroll
                for (int m = 0; m <= D; m++)
                    A.out.P[(r * A.out.ldp + j) * (D + 1) + m] = (int32_t)prof[m];
            }
        }
    }
}

/* S (tile-transposed, see BsArgs) -> rows of G.  Block = 64 columns x 64 row slots of one tile, moved
 * through LDS so that both the reads (64 slots of one column) and the writes (64 columns of one row)
 * are 512-byte runs.  grid (column blocks, tiles * NSLOT / 64). */
template <int NSLOT>
__global__ __launch_bounds__(256) void k_untile(const double *__restrict__ S, const int64_t *__restrict__ tile_soff,
                                                const int *__restrict__ tile_cbeg, const int *__restrict__ tile_cend,
                                                const int *__restrict__ tile_nrows, const int *__restrict__ tile_row,
                                                const int *__restrict__ tile_out, GramOut out)
{
    __shared__ double buf[64][65];
    const int tile = blockIdx.y / (NSLOT / 64), half = blockIdx.y % (NSLOT / 64);
    const int cbeg = tile_cbeg[tile], cend = tile_cend[tile], nrows = tile_nrows[tile];
    const int jb = cbeg + (int)blockIdx.x * 64;
    if (jb >= cend || half * 64 >= nrows) return;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const double *src = S + (tile_soff[tile] + (jb - cbeg)) * NSLOT + half * 64;
    for (int c = ty; c < 64; c += 4)
        if (jb + c < cend && half * 64 + tx < nrows) buf[c][tx] = src[(int64_t)c * NSLOT + tx];
    __syncthreads();
    for (int rl = ty; rl < 64; rl += 4) {
        const int rs = half * 64 + rl;
        if (rs >= nrows) break;
        const int row = tile_row[tile * gkmpack::MAX_ROWS + rs];
        const int j = jb + tx;
        if (j >= cend || (j > row && !out.write_all)) continue;
        const int64_t r = out.local_rows ? tile_out[tile * gkmpack::MAX_ROWS + rs] : row;
        *gram_cell(out, r, j) = buf[tx][rl];
    }
}

struct DirectArgs {
    const int *rows;
    int nrows;
    const int *len;
    const int64_t *lmoff;
    const uint32_t *lmf, *lmr; /* l-mer | weight << 24 */
    double c[GKM_MAXD1];
    GramOut out;
    int cj, L, d, mode, n;
};

/*
 * General fallback (any L <= 12, d <= 12): lane = row sequence, R row l-mers held in
 * registers, the column strand's packed l-mers streamed as wave-uniform scalars;
 * XOR / fold / popcount per comparison, rare exec-masked accumulate.
 */
__global__ __launch_bounds__(64) void k_gram_direct(const DirectArgs A)
{
    constexpr int R = 8;
    __shared__ uint32_t acc[GKM_MAXD1][64];
    const int lane = threadIdx.x;
    const int tile = blockIdx.y;
    const int ridx = tile * 64 + lane;
    const int a = ridx < A.nrows ? A.rows[ridx] : -1;
    const int amin = A.rows[tile * 64], amax = A.rows[min(tile * 64 + 63, A.nrows - 1)];
    const int cbeg = A.mode == COLS_DIAGONAL ? amin : 0, cend = A.mode == COLS_FULL ? A.n : amax + 1;
    const int j0 = cbeg + blockIdx.x * A.cj;
    const int j1 = min(j0 + A.cj, cend);
    if (j0 >= j1) return;
    const int d = A.d;
    const int na = a >= 0 ? A.len[a] - A.L + 1 : 0;
    const int64_t offa = a >= 0 ? A.lmoff[a] : 0;
    int namax = na;
    for (int s = 32; s >= 1; s >>= 1) namax = max(namax, __shfl_xor(namax, s));

    for (int j = j0; j < j1; j++) {
        const int nj = A.len[j] - A.L + 1;
        const int64_t offj = A.lmoff[j];
        for (int m = 0; m <= d; m++) acc[m][lane] = 0u;
        for (int p0 = 0; p0 < namax; p0 += R) {
            uint32_t u[R], wu[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const bool ok = (p0 + r) < na;
                u[r] = ok ? A.lmf[offa + p0 + r] : 0u;
                wu[r] = u[r] >> 24; /* padding rows have weight 0 and add nothing */
            }
            /* The column's l-mers come as SCALARS, eight of each strand per request (s_load_dwordx8 through the constant
             * address space).  As written in round 1 -- one vector load of a wave-uniform address per column l-mer, waited
             * for before its sixteen comparisons -- the kernel spent its time on that round trip: 226 ms whatever (L, d)
             * for 2 000 x 300 bp, of which the LDS read-add-write per hit was 68 (now one ds_add_u32, no return) and the
             * starved grid 100 (gram_launch: columns per workgroup by the size of the problem).  The
             * entries past the column's last l-mer (the next sequence's, or the 8 words of padding behind the table)
             * are compared like the others and carry the weight 0. */
            constexpr int QB = 8;
            const sgpr_words lf = (sgpr_words)(A.lmf + offj), lr = (sgpr_words)(A.lmr + offj);
            for (int q0 = 0; q0 < nj; q0 += QB) {
                uint32_t xf[QB], xr[QB];
#pragma unroll
                for (int t = 0; t < QB; t++) {
                    xf[t] = lf[q0 + t];
                    xr[t] = lr[q0 + t];
                }
#pragma unroll
                for (int t = 0; t < QB; t++) {
                    const bool live = q0 + t < nj;
                    const uint32_t wf = live ? xf[t] >> 24 : 0u, wr = live ? xr[t] >> 24 : 0u;
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        /* no test for m <= d: acc has a row for every possible m (GKM_MAXD1 = 13 >= L + 1), the rows
                         * above d are never read.  At the mismatch budgets this kernel serves a fifth to all of the
                         * pairs are hits anyway, and the compare + EXEC save / restore cost more than the LDS add. */
                        atomicAdd(&acc[gkmbs::lmer_mismatch(u[r], xf[t])][lane], wu[r] * wf);
                        atomicAdd(&acc[gkmbs::lmer_mismatch(u[r], xr[t])][lane], wu[r] * wr);
                    }
                }
            }
        }
        if (a >= 0 && (j <= a || A.out.write_all)) {
            double g = 0.0;
            for (int m = 0; m <= d; m++) g += A.c[m] * (double)(int32_t)acc[m][lane];
            const int64_t r = A.out.local_rows ? ridx : a;
            if (A.out.diag && j == a) A.out.diag[a] = g;
            if (A.out.G) *gram_cell(A.out, r, j) = g;
            if (A.out.P)
                for (int m = 0; m <= d; m++) A.out.P[(r * A.out.ldp + j) * (d + 1) + m] = (int32_t)acc[m][lane];
        }
    }
}

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

/* ---------------------------------------------------------- host: upload */
static int ensure_lmers(gkmhip_ctx *ctx, hipStream_t stream);
static int ensure_colpk(gkmhip_ctx *ctx, hipStream_t stream);
static int ensure_sb(gkmhip_ctx *ctx, int W, hipStream_t stream);
static bool bitslice_serves(const gkmhip_ctx *ctx); /* which kernel this context's launches take (gram_launch) */

extern "C" int gkmhip_set_sequences(gkmhip_ctx *ctx, int n, const uint8_t *codes, const int64_t *offsets,
                                    const uint8_t *wdist, int wdist_len, void *stream_)
{
    if (!ctx || n <= 0 || !codes || !offsets) return set_err_msg("gkmhip_set_sequences: bad arguments", 2);
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(hipSetDevice(ctx->device));
    const int L = ctx->L;
    const int weighted = (wdist != nullptr && wdist_len > 0) ? 1 : 0;
    /* a context may be reused for another set of sequences (gkmsvm.init_many keeps one per device): every
     * per-sequence table of the previous set is stale from here on, BEFORE anything below sizes itself by them */
    ctx->have_lmers = false;
    ctx->have_sb = false;
    ctx->have_colpk = false;
    ctx->n = 0;
    ctx->weighted = weighted;
    ctx->h_len.resize((size_t)n);
    ctx->h_lmoff.resize((size_t)n + 1);
    ctx->h_cum_n.resize((size_t)n + 1);
    ctx->h_lmoff[0] = 0;
    ctx->h_cum_n[0] = 0.0;
    ctx->maxlen = 0;
    for (int i = 0; i < n; i++) {
        const int64_t len = offsets[i + 1] - offsets[i];
        if (len < L) return set_err_msg("sequence " + std::to_string(i) + " is shorter than L", 3);
        if (len > 2047) return set_err_msg("sequence longer than 2047", 3);
        ctx->h_len[(size_t)i] = (int)len;
        ctx->h_lmoff[(size_t)i + 1] = ctx->h_lmoff[(size_t)i] + (len - L + 1);
        ctx->h_cum_n[(size_t)i + 1] = ctx->h_cum_n[(size_t)i] + (double)(len - L + 1);
        ctx->maxlen = std::max(ctx->maxlen, (int)len);
    }
    ctx->n = n;
    if (weighted && (wdist_len <= (ctx->maxlen - L + 1) / 2 || wdist_len > WD_LDS))
        return set_err_msg("distance weight table must cover 0..max(n)/2 and hold at most 1024 entries", 3);
    const size_t total = (size_t)offsets[n];
    if (ctx->codes.ensure(total) || ctx->off.ensure((size_t)n + 1) || ctx->lmoff.ensure((size_t)n + 1) ||
        ctx->len.ensure((size_t)n) || ctx->wd.ensure(WD_LDS))
        return 4;
    HIPCHK(hipMemcpyAsync(ctx->codes.p, codes, total, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(ctx->off.p, offsets, ((size_t)n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(ctx->lmoff.p, ctx->h_lmoff.data(), ((size_t)n + 1) * sizeof(int64_t),
                          hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(ctx->len.p, ctx->h_len.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, stream));
    /* unweighted kernel types: every positional weight is 1 (libgkm.c:926-932) -- a table of ones keeps the
     * hit path free of a weighted / unweighted branch */
    HIPCHK(hipMemsetAsync(ctx->wd.p, weighted ? 0 : 1, WD_LDS, stream));
    if (weighted) HIPCHK(hipMemcpyAsync(ctx->wd.p, wdist, (size_t)wdist_len, hipMemcpyHostToDevice, stream));
    ctx->wd_len = weighted ? wdist_len : (ctx->maxlen - L + 1) / 2 + 1;
    /* The per-sequence device tables are built HERE, not at the first launch: callers alternate launches between
     * two streams (gkm_multi.hip, bench.py), and a table built by the first launch on one stream was read by the
     * second launch on the other stream before it was complete (found when the host stopped waiting for its
     * uploads: the config-4 stand-in through two contexts differed in a few hundred rows). */
    if (bitslice_serves(ctx) && (ensure_sb(ctx, 10, stream) || ensure_colpk(ctx, stream))) return 4;
    if (!bitslice_serves(ctx) && ensure_lmers(ctx, stream)) return 4;
    /* the sources are the caller's (pageable) arrays: an asynchronous copy of more than a few KB may still be
     * reading them after this call has returned, so the upload is finished here (3 MB, once per matrix) */
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
}

static int ensure_lmers(gkmhip_ctx *ctx, hipStream_t stream)
{
    if (ctx->have_lmers) return 0;
    const size_t total_lm = (size_t)ctx->h_lmoff[(size_t)ctx->n];
    /* one buffer: the reverse-strand table sits lm_stride entries after the forward one, so the hit
     * path selects the strand with an index offset instead of a pointer select */
    if (total_lm >= (size_t)1 << 29) return set_err_msg("l-mer tables exceed 2^29 entries per strand", 4);
    if (ctx->lmf.ensure(2 * total_lm + 8)) return 4; /* (+ 8: k_gram_direct reads the column's l-mers eight at a time) */
    ctx->lm_stride = (uint32_t)total_lm;
    hipLaunchKernelGGL(k_pack_lmers, dim3((unsigned)ctx->n), dim3(128), 0, stream, ctx->codes.p, ctx->off.p,
                       ctx->lmoff.p, ctx->L, ctx->wd.p, ctx->weighted, ctx->lmf.p, ctx->lmf.p + total_lm);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(stream)); /* complete before any OTHER stream may read the table */
    ctx->have_lmers = true;
    return 0;
}

static int ensure_colpk(gkmhip_ctx *ctx, hipStream_t stream)
{
    if (ctx->have_colpk) return 0;
    /* one word more than the bases need: the hit path reads words q/16 and q/16 + 1 */
    const int pkw = (ctx->maxlen + 15) / 16 + 1;
    if (ctx->colpk.ensure((size_t)ctx->n * 2 * (size_t)pkw)) return 4;
    hipLaunchKernelGGL(k_pack_strands, dim3((unsigned)ctx->n * 2), dim3(64), 0, stream, ctx->codes.p, ctx->off.p, pkw,
                       ctx->colpk.p);
    HIPCHK(hipGetLastError());
    ctx->pkw = pkw;
    HIPCHK(hipStreamSynchronize(stream)); /* complete before any OTHER stream may read the table */
    ctx->have_colpk = true;
    return 0;
}

static int ensure_sb(gkmhip_ctx *ctx, int W, hipStream_t stream)
{
    if (ctx->have_sb && ctx->sb_W == W) return 0;
    const int xw = ((ctx->maxlen + W + 2 * BS_DU + 15) / 16) * 16;
    if (ctx->sb.ensure((size_t)ctx->n * 2 * 2 * (size_t)xw)) return 4;
    hipLaunchKernelGGL(k_build_sb, dim3((unsigned)ctx->n * 2, 2), dim3(256), 0, stream, ctx->codes.p, ctx->off.p,
                       W, ctx->L, xw, ctx->sb.p);
    HIPCHK(hipGetLastError());
    ctx->sb_xw = xw;
    ctx->sb_W = W;
    HIPCHK(hipStreamSynchronize(stream)); /* complete before any OTHER stream may read the table */
    ctx->have_sb = true;
    return 0;
}

/* ------------------------------------------------- host: launch dispatch */
typedef void (*bs_kernel_t)(const BsArgs);

template <int W, int PACKED>
static bs_kernel_t pick_bitslice(int L, int d)
{
#define GKM_BS(LL, DD) \
    if (L == LL && d == DD) return k_gram_bitslice<W, LL, DD, PACKED>;
    /* every (L, d) with 3 <= L <= 12, d <= min(4, L - 1) (what bin/gkmqc.py:185 can ask for), plus the d > 4 pairs
     * where this kernel beats k_gram_direct -- see auto_takes_bitslice() below for where that is. */
#define GKM_BS_L(LL) GKM_BS(LL, 0) GKM_BS(LL, 1) GKM_BS(LL, 2) GKM_BS(LL, 3) GKM_BS(LL, 4)
    GKM_BS(3, 0) GKM_BS(3, 1) GKM_BS(3, 2)
    GKM_BS(4, 0) GKM_BS(4, 1) GKM_BS(4, 2) GKM_BS(4, 3)
    GKM_BS_L(5) GKM_BS_L(6) GKM_BS_L(7) GKM_BS_L(8) GKM_BS_L(9) GKM_BS_L(10) GKM_BS_L(11) GKM_BS_L(12)
    GKM_BS(11, 5) GKM_BS(12, 5) GKM_BS(12, 6)
#undef GKM_BS_L
#undef GKM_BS
    return nullptr;
}

/* Which kernel `auto` takes.  The general kernel's time does not depend on (L, d) or on the data; the bit-sliced
 * kernel's grows with the share of window pairs within d mismatches, every one of which takes a lane of a trip.
 * Measured in round 4 (tools/high_d_ab.py, profiles/r4_high_d_bitslice_vs_direct.txt; 8 000 x 300 bp iid, whole
 * triangle): general kernel 688 ms throughout (7.9e12 comparisons/s; 810 before it dropped the test for m <= d, ~1 160
 * as rounds 1-3 had it); bit-sliced (12,5) 152 ms at 1.4 % hits, (11,5) 329 at 3.4 %, (9,4) 454 at 4.9 %, (12,6) 485 at
 * 5.4 %, (7,3) 618 at 7.1 %, (10,5) 717 at 7.8 %, (11,6) 1 000 at 11.5 %, (8,4) 1 021 at 11.4 % -- break-even at ~7.5 % of
 * the windows, close to where rounds 1-3 had put it by counting instructions (8 %).  (A first measurement on 2 000
 * sequences said 30 %: at that size the general kernel's grid did not fill the GPU -- its column chunks now shrink
 * with the problem.)  The rule is the iid share of (L, d); it also sends the dense pairs of short words -- (8,4) 11 %,
 * (7,4) 24 %, (6,3) 17 %, (5,2) 10 %, ... -- to the general kernel, up to 3.1x faster there.  Peak-like data costs the
 * bit-sliced kernel ~4 % more at the threshold; (7,3) keeps its lead there. */
static double iid_hit_share(int L, int d)
{
    double sum = 0.0, term = 1.0; /* C(L, m) 3^m */
    for (int m = 0; m <= d && m <= L; m++) {
        sum += term;
        term = term * 3.0 * (double)(L - m) / (double)(m + 1);
    }
    return sum / pow(4.0, (double)L);
}
#ifndef GKM_BITSLICE_MAX_HIT_SHARE
#define GKM_BITSLICE_MAX_HIT_SHARE 0.075
#endif
static bool auto_takes_bitslice(int L, int d) { return iid_hit_share(L, d) <= GKM_BITSLICE_MAX_HIT_SHARE; }

static bool bitslice_serves(const gkmhip_ctx *ctx)
{
    if (ctx->kernel_pref == GKMHIP_KERNEL_DIRECT || !pick_bitslice<10, 2>(ctx->L, ctx->d)) return false;
    return ctx->kernel_pref == GKMHIP_KERNEL_BITSLICE || auto_takes_bitslice(ctx->L, ctx->d);
}


/* One launch of the Gram kernel for a set of rows.  mode says which columns every tile of rows
 * visits: COLS_TRIANGLE j <= largest row of the tile (the path of gkm_main_pywrapper),
 * COLS_FULL every sequence, COLS_DIAGONAL only the band of the tile's own rows (self norms). */
static int gram_launch(gkmhip_ctx *ctx, const int *rows, int nrows, int mode, GramOut out, hipStream_t stream)
{
    if (!ctx || !rows || nrows <= 0) return set_err_msg("gram: bad arguments", 2);
    if (ctx->n <= 0) return set_err_msg("gram: no sequences uploaded", 2);
    HIPCHK(hipSetDevice(ctx->device));
    (void)hipGetLastError(); /* the launch checks below must see this call's errors only */
    const int L = ctx->L, d = ctx->d, n = ctx->n;
    double comparisons = 0;
    for (int i = 0; i < nrows; i++) {
        if (rows[i] < 0 || rows[i] >= n || (i > 0 && rows[i] <= rows[i - 1]))
            return set_err_msg("rows must be strictly ascending sequence indices", 2);
        const double na = (double)(ctx->h_len[(size_t)rows[i]] - L + 1);
        comparisons += 2.0 * na * (mode == COLS_FULL ? ctx->h_cum_n[(size_t)n] : mode == COLS_DIAGONAL ? na : ctx->h_cum_n[(size_t)rows[i] + 1]);
    }
    out.write_all = mode == COLS_FULL ? 1 : 0;

    /* W = 10 words per lane; W = 20 was measured too (config 2: 121 vs 118 ms, 150 bp: 56 vs 31 ms):
     * the longer per-shift chain does not pay for the registers it costs */
    bs_kernel_t bs10 = nullptr;
    if (ctx->kernel_pref != GKMHIP_KERNEL_DIRECT) bs10 = pick_bitslice<10, 2>(L, d);
    if (ctx->kernel_pref == GKMHIP_KERNEL_BITSLICE && !bs10)
        return set_err_msg("bit-sliced kernel not instantiated for this (L, d)", 5);
    if (!bitslice_serves(ctx)) bs10 = nullptr; /* auto: the general kernel where it is the faster one */

    if (bs10) {
        /* pack the rows into lanes at bit-row granularity (gkm_pack.h) */
        std::vector<int> nwin((size_t)nrows);
        for (int i = 0; i < nrows; i++) nwin[(size_t)i] = ctx->h_len[(size_t)rows[i]] - L + 1;
        /* At most 64 rows per tile unless that leaves lanes empty (rows shorter than half a lane): the
         * 64-slot kernels keep a wave more per SIMD (k_gram_bitslice) */
        gkmpack::Packing pk = gkmpack::pack_rows(rows, nwin.data(), nrows, 10, L, 64);
        int slots = 64;
        {
            gkmpack::Packing wide = gkmpack::pack_rows(rows, nwin.data(), nrows, 10, L, gkmpack::MAX_ROWS);
            if (getenv("GKM_FORCE_PACKED") ? !strcmp(getenv("GKM_FORCE_PACKED"), "128")
                                           : (double)pk.ntiles > 1.04 * (double)wide.ntiles) {
                pk = std::move(wide);
                slots = gkmpack::MAX_ROWS;
            }
        }
        const int W = pk.W, ntiles = pk.ntiles;
        /* no lane with a second piece -> the leaner kernel variant */
        bool packed = getenv("GKM_FORCE_PACKED") != nullptr || slots != 64;
        for (size_t k = 1; k < pk.pieces.size() && !packed; k++) packed = pk.pieces[k].lane == pk.pieces[k - 1].lane;
        const int NP = packed ? gkmpack::MAX_PIECES : 1, LPW = packed ? NP : 2;
        bs_kernel_t bs = !packed ? pick_bitslice<10, 0>(L, d) : slots == 64 ? pick_bitslice<10, 1>(L, d) : bs10;
        /* (normally built by gkmhip_set_sequences; before ctx->pkw sizes the dynamic LDS below) */
        if (ensure_sb(ctx, W, stream) || ensure_colpk(ctx, stream)) return 4;
        /* GKM_LDS_PAD=<bytes> (experiments): extra dynamic LDS per wave, i.e. fewer waves per CU -- how much does the
         * kernel depend on its occupancy? */
        const size_t lds_pad = getenv("GKM_LDS_PAD") ? (size_t)atoi(getenv("GKM_LDS_PAD")) : 0;
        const size_t dyn_lds = (size_t)(2 * ctx->pkw + (ctx->wd_len + 3) / 4) * sizeof(uint32_t) + lds_pad;
        bool bperm = false;
        if (!packed) { /* the variant without the piece table in LDS, where that saves an LDS allocation granule */
            hipFuncAttributes fa, fb;
            const char *force = getenv("GKM_FORCE_BPERM");
            bs_kernel_t bsp = pick_bitslice<10, 3>(L, d);
            auto granules = [](size_t bytes) { return (bytes + 1279) / 1280; };
            if (force ? atoi(force) != 0
                      : (hipFuncGetAttributes(&fa, (const void *)bs) == hipSuccess &&
                         hipFuncGetAttributes(&fb, (const void *)bsp) == hipSuccess &&
                         granules(fa.sharedSizeBytes + dyn_lds) > granules(fb.sharedSizeBytes + dyn_lds)))
                bperm = true;
            if (bperm) bs = bsp;
        }
        const size_t nl = (size_t)ntiles * 64;
        std::vector<int> desc(nl * gkmpack::MAX_PIECES * 5, 0);
        std::vector<uint32_t> lane_mask(nl, 0u), lane_piece(nl * (size_t)LPW, 0u);
        std::vector<int> fill(nl, 0);
        for (const gkmpack::Piece &pc : pk.pieces) {
            const int k = fill[(size_t)pc.lane]++;
            int *dd = &desc[((size_t)pc.lane * gkmpack::MAX_PIECES + k) * 5];
            dd[0] = pc.row; dd[1] = pc.b0; dd[2] = pc.nb; dd[3] = pc.p0; dd[4] = pc.cnt;
            lane_mask[(size_t)pc.lane] |= 1u << pc.b0;
            /* byte offset of the row slot in accl[m][.]; the row l-mer at lane position i0 is |c0 - i0| l-mers
             * away from its sequence's centre l-mer, c0 > -2048 is stored with a bias of 2048 so that the
             * kernel's unsigned |a - b| applies */
            /* second profile copy (k_gram_bitslice two_copies): odd lanes of a tile with at most slots / 2 rows */
            const int tile_of = pc.lane / 64;
            const bool second = 2 * pk.tile_nrows[(size_t)tile_of] <= slots && (pc.lane & 1);
            const uint32_t slot4 = ((uint32_t)pc.slot + (second ? (uint32_t)slots / 2u : 0u)) * 4u;
            const uint32_t c0b = (uint32_t)((ctx->h_len[(size_t)pc.row] - L + 1) / 2 - pc.p0 + pc.b0 * W + 2048);
            if (packed) {
                lane_piece[(size_t)pc.lane * NP + k] = slot4 | (c0b << 16);
            } else {
                lane_piece[(size_t)pc.lane * 2] = slot4;
                lane_piece[(size_t)pc.lane * 2 + 1] = c0b;
            }
        }
        /* columns [cbeg, cend) per tile */
        std::vector<int> cbeg((size_t)ntiles, 0), cend((size_t)ntiles, 0);
        std::vector<int64_t> soff((size_t)ntiles + 1, 0);
        for (int t = 0; t < ntiles; t++) {
            int amin = n;
            for (int rs = 0; rs < pk.tile_nrows[(size_t)t]; rs++) amin = std::min(amin, pk.tile_row[(size_t)t * gkmpack::MAX_ROWS + rs]);
            cbeg[(size_t)t] = mode == COLS_DIAGONAL ? amin : 0;
            cend[(size_t)t] = mode == COLS_FULL ? n : pk.tile_amax[(size_t)t] + 1;
            soff[(size_t)t + 1] = soff[(size_t)t] + (cend[(size_t)t] - cbeg[(size_t)t]);
        }
        if (soff[(size_t)ntiles] <= 0 || soff[(size_t)ntiles] > 0x7fffffffLL) return set_err_msg("gram: bad work item count", 2);

        /* every per-launch table goes to the device in ONE copy (the boundary call issues 13 launches) */
        std::vector<char> blob;
        auto put = [&](const void *src, size_t bytes) {
            const size_t at = (blob.size() + 255) & ~(size_t)255;
            blob.resize(at + bytes);
            memcpy(blob.data() + at, src, bytes);
            return at;
        };
        const size_t o_desc = put(desc.data(), desc.size() * sizeof(int));
        const size_t o_mask = put(lane_mask.data(), nl * sizeof(uint32_t));
        const size_t o_piece = put(lane_piece.data(), lane_piece.size() * sizeof(uint32_t));
        const size_t o_trow = put(pk.tile_row.data(), pk.tile_row.size() * sizeof(int));
        const size_t o_tout = put(pk.tile_out.data(), pk.tile_out.size() * sizeof(int));
        const size_t o_tn = put(pk.tile_nrows.data(), (size_t)ntiles * sizeof(int));
        const size_t o_cbeg = put(cbeg.data(), (size_t)ntiles * sizeof(int));
        const size_t o_cend = put(cend.data(), (size_t)ntiles * sizeof(int));
        const size_t o_soff = put(soff.data(), soff.size() * sizeof(int64_t));
        const size_t o_roff = out.row_off ? put(out.row_off, (size_t)nrows * sizeof(int64_t)) : 0;
        /* work-item order: (column chunk, tile) entries (BsArgs) where that is free.  Measured (tools/col_chunk_sweep2.sh,
         * profiles/r4_col_chunk_sweep.txt; kernel ms / GB read from L2 misses per launch): config 2 plain order 72.4 / 5.48,
         * chunks of 4 096 columns 72.5 / 0.19 -- the column tables (5.3 KB per 300-bp column) of a chunk, dealt over the 8
         * XCDs, are 2.7 MB per L2 and stay there; config 5 151.2 / 12.1 against 151.8 / 2.3; gkmQC's shape (10.3 KB per
         * column) 384.1 / 28.7 against 385.4 / 21.9 at 4 096 (5.3 MB per L2: no reuse) and 386.4 / 4.2 at 2 560.  SMALLER
         * chunks cost time: the 28 waves of a CU then belong to 3-5 tiles instead of 1-2 and their hit paths evict each
         * other's packed rows from the 32 KB L1 (1 024 columns: +2 % on config 2, +6 % on the other two).  The traffic
         * binds nothing (80 GB/s against 8 TB/s), the kernel's time is what counts: chunks of 4 096 columns where a chunk's
         * tables fit 3 MB per XCD (config 2, config 3), the plain tile-major order everywhere else.  GKM_COL_CHUNK=<columns>
         * overrides, 0 = plain. */
        std::vector<int64_t> ent_off;
        std::vector<int> ent_tile, ent_j0, ent_j1;
        int64_t n_items = soff[(size_t)ntiles];
        {
            const double mean_len = ctx->h_cum_n[(size_t)n] / n + (L - 1);
            const double col_bytes = (4.0 * (mean_len + W) + 2.0 * (mean_len / 16.0 + 1.0)) * sizeof(uint32_t);
            long chunk = col_bytes * 4096.0 / 8.0 <= 3.0 * 1048576.0 ? 4096 : 0;
            if (const char *cc = getenv("GKM_COL_CHUNK")) chunk = atol(cc) & ~7L;
            if (chunk >= 8 && chunk < n) {
                int64_t at = 0;
                for (long c0 = 0; c0 < n; c0 += chunk)
                    for (int t = 0; t < ntiles; t++) {
                        const int j0 = std::max<long>(cbeg[(size_t)t], c0), j1 = (int)std::min<long>(cend[(size_t)t], c0 + chunk);
                        if (j0 >= j1) continue;
                        ent_off.push_back(at);
                        ent_tile.push_back(t);
                        ent_j0.push_back(j0);
                        ent_j1.push_back(j1);
                        at += (j1 - j0 + 7) & ~7;
                    }
                ent_off.push_back(at);
                if (at > 0x7fffffffLL) { ent_off.clear(); ent_tile.clear(); } /* (too many items with the padding: plain order) */
                else n_items = at;
            }
        }
        const int nent = (int)ent_tile.size();
        const size_t o_eoff = nent ? put(ent_off.data(), ent_off.size() * sizeof(int64_t)) : 0;
        const size_t o_etile = nent ? put(ent_tile.data(), (size_t)nent * sizeof(int)) : 0;
        const size_t o_ej0 = nent ? put(ent_j0.data(), (size_t)nent * sizeof(int)) : 0;
        const size_t o_ej1 = nent ? put(ent_j1.data(), (size_t)nent * sizeof(int)) : 0;
        const int NS = slots;
        auto &scr = ctx->scratch[ctx->sel];
        /* (The tables and row planes on a second stream and untile on a third, so that the Gram kernels of the drop-in
         * call's row blocks follow each other with nothing in between, was built and measured in round 4: k_untile's
         * 33 KB workgroups then wait for room beside the next block's Gram kernel -- its 28 waves per CU leave 17 KB of
         * LDS -- and finish only when it does; the copies start one block late: 90.6 instead of 81.4 ms for the call.) */
        /* words of a lane's packed positions: 32 W / 16 + 1 are used (the hit path reads two); the stride is 128 bytes,
         * so that the lane field of a record's origin word is the lane's byte offset (gkm_bitslice.h pack_meta) */
        const int rpw = 32;
        static_assert(32 * 10 / 16 + 1 <= 32, "a lane's packed positions fit 128 bytes");
        PinBuf *hb = pin_acquire(blob.size());
        if (!hb) return set_err_msg("gram: pinned host buffer for the launch tables", 4);
        if (scr.tables.ensure(blob.size(), true) || scr.rowplanes.ensure(nl * 3 * W, true) ||
            scr.rowpk.ensure(nl * (size_t)rpw, true) ||
            (out.G && scr.S.ensure((size_t)soff[(size_t)ntiles] * (size_t)NS, true))) {
            pin_release(hb);
            return 4;
        }
        /* through a pinned buffer that outlives the call: an asynchronous copy from a local (pageable) vector
         * may still be reading it after this function has returned and freed it */
        memcpy(hb->p, blob.data(), blob.size());
        {
            hipError_t ce = hipMemcpyAsync(scr.tables.p, hb->p, blob.size(), hipMemcpyHostToDevice, stream);
            if (ce == hipSuccess && hipLaunchHostFunc(stream, pin_release, hb) != hipSuccess) {
                ce = hipStreamSynchronize(stream); /* no host function: hand the buffer back once the copy is over */
                pin_release(hb);
            } else if (ce != hipSuccess) {
                pin_release(hb);
            }
            HIPCHK(ce);
        }
        char *tb = scr.tables.p;
        static_assert(10 * 4 >= 32, "plane 3: ten parts of four packed words cover the lane's 32");
        hipLaunchKernelGGL(k_build_rowplanes, dim3((unsigned)ntiles, 4, (unsigned)W), dim3(64), 0, stream, ctx->codes.p,
                           ctx->off.p, (const int *)(tb + o_desc), W, scr.rowplanes.p, scr.rowpk.p, rpw);
        HIPCHK(hipGetLastError());

        BsArgs A;
        A.rowplanes = scr.rowplanes.p; A.lane_mask = (const uint32_t *)(tb + o_mask); A.lane_piece = (const uint32_t *)(tb + o_piece);
        A.tile_row = (const int *)(tb + o_trow); A.tile_out = (const int *)(tb + o_tout); A.tile_nrows = (const int *)(tb + o_tn);
        A.tile_cbeg = (const int *)(tb + o_cbeg); A.tile_cend = (const int *)(tb + o_cend);
        A.rowpk = scr.rowpk.p; A.colpk = ctx->colpk.p; A.wd32 = (const uint32_t *)ctx->wd.p;
        A.rpw = rpw; A.pkw = ctx->pkw; A.wd_words = (ctx->wd_len + 3) / 4;
        A.sb = ctx->sb.p; A.xw = ctx->sb_xw;
        A.len = ctx->len.p;
        for (int m = 0; m < GKM_MAXD1; m++) A.c[m] = ctx->c[m];
        A.out = out;
        if (out.row_off) A.out.row_off = (const int64_t *)(tb + o_roff);
        A.ntiles = ntiles;
        A.S = out.G ? scr.S.p : nullptr;
        A.tile_soff = (const int64_t *)(tb + o_soff);
        A.nent = nent;
        A.ent_off = (const int64_t *)(tb + o_eoff);
        A.ent_tile = (const int *)(tb + o_etile);
        A.ent_j0 = (const int *)(tb + o_ej0);
        A.ent_j1 = (const int *)(tb + o_ej1);
        /* One column sequence per work item: a wave of the full-size problem lives ~0.6 ms, which is what
         * the drain at the end of every launch costs -- nothing for one big launch, but the boundary call
         * issues 13 launches and the multi-GPU path one per chunk. */
        HIPCHK(hipEventRecord(ctx->ev0, stream));
        hipLaunchKernelGGL(bs, dim3((unsigned)n_items), dim3(64), dyn_lds, stream, A);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(ctx->ev1, stream));
        if (out.G) {
            int span = 0;
            for (int t = 0; t < ntiles; t++) span = std::max(span, cend[(size_t)t] - cbeg[(size_t)t]);
            const dim3 ug((unsigned)((span + 63) / 64), (unsigned)(ntiles * (NS / 64)));
            if (slots != 64)
                hipLaunchKernelGGL(k_untile<gkmpack::MAX_ROWS>, ug, dim3(256), 0, stream, scr.S.p, A.tile_soff, A.tile_cbeg,
                                   A.tile_cend, A.tile_nrows, A.tile_row, A.tile_out, A.out);
            else
                hipLaunchKernelGGL(k_untile<64>, ug, dim3(256), 0, stream, scr.S.p, A.tile_soff, A.tile_cbeg, A.tile_cend,
                                   A.tile_nrows, A.tile_row, A.tile_out, A.out);
            HIPCHK(hipGetLastError());
        }
        ctx->last_kernel = bperm ? "k_gram_bitslice<bperm>" : !packed ? "k_gram_bitslice" : slots == 64 ? "k_gram_bitslice<packed>" : "k_gram_bitslice<packed,128>";
    } else {
        if (ensure_lmers(ctx, stream)) return 4;
        if (ctx->scratch[ctx->sel].rows.ensure((size_t)nrows)) return 4;
        HIPCHK(hipMemcpyAsync(ctx->scratch[ctx->sel].rows.p, rows, (size_t)nrows * sizeof(int), hipMemcpyHostToDevice, stream));
        if (out.row_off) {
            if (ctx->scratch[ctx->sel].rowoff.ensure((size_t)nrows)) return 4;
            HIPCHK(hipMemcpyAsync(ctx->scratch[ctx->sel].rowoff.p, out.row_off, (size_t)nrows * sizeof(int64_t),
                                  hipMemcpyHostToDevice, stream));
            out.row_off = ctx->scratch[ctx->sel].rowoff.p;
        }
        HIPCHK(hipStreamSynchronize(stream)); /* `rows` is the caller's: see gkmhip_set_sequences */
        DirectArgs A;
        A.rows = ctx->scratch[ctx->sel].rows.p; A.nrows = nrows;
        A.len = ctx->len.p; A.lmoff = ctx->lmoff.p; A.lmf = ctx->lmf.p; A.lmr = ctx->lmf.p + ctx->lm_stride;
        for (int m = 0; m < GKM_MAXD1; m++) A.c[m] = ctx->c[m];
        A.out = out;
        A.L = L; A.d = d; A.mode = mode; A.n = n;
        const unsigned ntiles = (unsigned)((nrows + 63) / 64);
        int span = 0; /* widest column range of any 64-row tile */
        double items = 0; /* (tile, column) pairs of the launch */
        for (unsigned t = 0; t < ntiles; t++) {
            const int amin = rows[t * 64], amax = rows[std::min<int>((int)t * 64 + 63, nrows - 1)];
            const int cols = mode == COLS_FULL ? n : mode == COLS_DIAGONAL ? amax + 1 - amin : amax + 1;
            span = std::max(span, cols);
            items += cols;
        }
        /* columns per workgroup: 16 where that still gives the GPU ~16 waves per SIMD, fewer for small problems (2 000
         * sequences: 2 000 workgroups of 16 columns left three quarters of the SIMDs idle, 156 ms; now 2 columns) */
        A.cj = (int)std::min(16.0, std::max(1.0, floor(items / 16384.0)));
        const unsigned nchunks = (unsigned)((span + A.cj - 1) / A.cj);
        HIPCHK(hipEventRecord(ctx->ev0, stream));
        hipLaunchKernelGGL(k_gram_direct, dim3(nchunks, ntiles), dim3(64), 0, stream, A);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(ctx->ev1, stream));
        ctx->last_kernel = "k_gram_direct";
    }
    ctx->ev_valid = true;
    ctx->last_comparisons = comparisons;
    return 0;
}

extern "C" int gkmhip_gram_rows(gkmhip_ctx *ctx, const int *rows, int nrows, int local_rows, double *G,
                                int64_t ld, int32_t *P, int64_t ldp, void *stream_)
{
    if (!G || !rows || nrows <= 0) return set_err_msg("gkmhip_gram_rows: bad arguments", 2);
    if (ld <= rows[nrows - 1]) return set_err_msg("leading dimension too small", 2);
    GramOut out;
    out.G = G; out.ld = ld; out.P = P; out.ldp = ldp; out.local_rows = local_rows; out.write_all = 0; out.diag = nullptr; out.row_off = nullptr;
    return gram_launch(ctx, rows, nrows, COLS_TRIANGLE, out, (hipStream_t)stream_);
}

extern "C" int gkmhip_gram_rows_packed(gkmhip_ctx *ctx, const int *rows, int nrows, double *G, const int64_t *row_off,
                                       void *stream_)
{
    if (!G || !rows || nrows <= 0 || !row_off) return set_err_msg("gkmhip_gram_rows_packed: bad arguments", 2);
    for (int i = 0; i < nrows; i++) /* rows may touch (row i ends where row i + 1 starts) but never overlap */
        if (row_off[i] < 0 || (i + 1 < nrows && row_off[i + 1] < row_off[i] + (int64_t)rows[i] + 1))
            return set_err_msg("gkmhip_gram_rows_packed: row offsets must leave rows[i] + 1 doubles per row", 2);
    GramOut out;
    out.G = G; out.ld = 0; out.P = nullptr; out.ldp = 0; out.local_rows = 1; out.write_all = 0; out.diag = nullptr;
    out.row_off = row_off;
    return gram_launch(ctx, rows, nrows, COLS_TRIANGLE, out, (hipStream_t)stream_);
}

extern "C" int gkmhip_gram_rows_full(gkmhip_ctx *ctx, const int *rows, int nrows, int local_rows, double *G,
                                     int64_t ld, void *stream_)
{
    if (!ctx || !G || !rows || nrows <= 0) return set_err_msg("gkmhip_gram_rows_full: bad arguments", 2);
    if (ld < ctx->n) return set_err_msg("leading dimension too small", 2);
    GramOut out;
    out.G = G; out.ld = ld; out.P = nullptr; out.ldp = 0; out.local_rows = local_rows; out.write_all = 1; out.diag = nullptr; out.row_off = nullptr;
    return gram_launch(ctx, rows, nrows, COLS_FULL, out, (hipStream_t)stream_);
}

__global__ void k_sqrt_inplace(double *__restrict__ v, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = sqrt(v[i]); /* libgkm.c:753-758 */
}

extern "C" int gkmhip_self_norms(gkmhip_ctx *ctx, double *sqnorm, void *stream_)
{
    if (!ctx || !sqnorm || ctx->n <= 0) return set_err_msg("gkmhip_self_norms: bad arguments", 2);
    hipStream_t stream = (hipStream_t)stream_;
    std::vector<int> all((size_t)ctx->n);
    for (int i = 0; i < ctx->n; i++) all[(size_t)i] = i;
    GramOut out;
    out.G = nullptr; out.ld = 0; out.P = nullptr; out.ldp = 0; out.local_rows = 0; out.write_all = 0; out.diag = sqnorm; out.row_off = nullptr;
    const int rc = gram_launch(ctx, all.data(), ctx->n, COLS_DIAGONAL, out, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(k_sqrt_inplace, dim3((unsigned)((ctx->n + 255) / 256)), dim3(256), 0, stream, sqnorm, ctx->n);
    HIPCHK(hipGetLastError());
    return 0;
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
static int normalize_rows(gkmhip_ctx *ctx, double *G, int64_t ld, int r0, int r1, double *sq, int symmetric,
                          hipStream_t stream, bool have_norms = false)
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

/* ---- device matrix -> the caller's (pageable) host rows, shared by the drop-in call and gkmhip_copy_lower_to_rows ----
 * A PIECE is rows [r0, r1) with columns [0, r1): it fits one pinned staging buffer.  Piece q+1 travels (hipMemcpy2DAsync
 * on `sd`, issued by `issue`) while the host threads scatter piece q into rows[r][0..r]. */
struct RowPiece { int r0, r1; };

/* pieces of rows [r0, r1) whose staging rectangles hold at most `bytes` */
static void cut_pieces(int r0, int r1, size_t bytes, std::vector<RowPiece> &out)
{
    for (int q0 = r0; q0 < r1;) {
        int q1 = q0 + 1;
        while (q1 < r1 && (size_t)(q1 + 1) * (size_t)(q1 + 1 - q0) * 8 <= bytes) q1++;
        out.push_back({q0, q1});
        q0 = q1;
    }
}

struct StreamSet { /* destroyed on every path out of the function that owns it */
    hipStream_t s[4] = {nullptr, nullptr, nullptr, nullptr};
    int create(int count)
    {
        for (int i = 0; i < count; i++) HIPCHK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
        return 0;
    }
    ~StreamSet()
    {
        for (hipStream_t x : s)
            if (x) (void)hipStreamDestroy(x);
    }
};

/* The copy-out pipeline's streams, kept per device for the life of the process (gkmhip_release_host_cache frees them)
 * and PROVEN to run beside each other.  HIP maps streams onto a handful of hardware queues (4 by default) in creation
 * order, and two streams that land on one queue execute in order: round 4's first version created four streams per
 * call, the copy stream shared the Gram stream's queue, and every device-to-host copy of the call waited for the LAST
 * Gram kernel (first piece in staging at 84 ms of 92, tools/boundary_ab.py --trace).  So: a candidate stream is kept
 * only if a tiny copy on it completes while a 2-ms spin kernel is still running on the compute stream; candidates
 * that fail stay alive until the search is over, so that the next one lands on another queue. */
__global__ void k_spin(long long ticks, unsigned *sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (ticks == 1234567) sink[0] = 1u;
}

/* what the drop-in call's copy-out pipeline measured last time (gram_part_to_host_rows cuts its row blocks by it) */
static struct {
    std::mutex m;
    double scatter_Bps_per_thread = 0, copy_Bps = 0, cmp_per_s = 0;
} g_ship;

struct PipeStreams {
    hipStream_t compute = nullptr, copy = nullptr;
    int probes = 0;
    bool copy_beside = false; /* proven to run beside `compute` */
};
static std::mutex g_pipe_mutex;
static PipeStreams g_pipe[64];

static bool runs_beside(hipStream_t busy, hipStream_t other, unsigned *d_word, unsigned *h_word)
{
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, busy, (long long)200000, d_word); /* 2 ms at 100 MHz */
    bool beside = false;
    if (hipMemcpyAsync(h_word, d_word + 1, sizeof(unsigned), hipMemcpyDeviceToHost, other) == hipSuccess &&
        hipStreamSynchronize(other) == hipSuccess)
        beside = hipStreamQuery(busy) == hipErrorNotReady;
    (void)hipStreamSynchronize(busy);
    return beside;
}

/* A new non-blocking stream on the current device that runs beside every stream of `busy` (see PipeStreams: streams
 * that share a hardware queue execute in order); after six candidates the last one is returned whatever it shares.
 * *beside says which it was. */
extern "C" void *gkmhip_create_stream_beside(void *const *busy, int nbusy, int *beside)
{
    unsigned *d_word = nullptr, *h_word = nullptr;
    if (hipMalloc((void **)&d_word, 2 * sizeof(unsigned)) != hipSuccess) return nullptr;
    if (hipHostMalloc((void **)&h_word, sizeof(unsigned), hipHostMallocPortable) != hipSuccess) {
        (void)hipFree(d_word);
        return nullptr;
    }
    std::vector<hipStream_t> rejected;
    hipStream_t got = nullptr;
    bool ok = false;
    for (int attempt = 0; attempt < 6 && !ok; attempt++) {
        hipStream_t c = nullptr;
        if (hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess) break;
        ok = true;
        for (int i = 0; i < nbusy && ok; i++) ok = runs_beside((hipStream_t)busy[i], c, d_word, h_word);
        if (ok || attempt == 5) got = c;
        else rejected.push_back(c);
    }
    for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
    (void)hipFree(d_word);
    (void)hipHostFree(h_word);
    if (beside) *beside = ok ? 1 : 0;
    return got;
}

static int pipe_streams(int device, PipeStreams **out)
{
    if (device < 0 || device >= 64) return set_err_msg("device ordinal out of range", 2);
    std::lock_guard<std::mutex> lock(g_pipe_mutex);
    PipeStreams &P = g_pipe[device];
    *out = &P;
    if (P.compute) return 0;
    HIPCHK(hipStreamCreateWithFlags(&P.compute, hipStreamNonBlocking));
    void *busy[1] = {P.compute};
    int beside = 0;
    P.copy = (hipStream_t)gkmhip_create_stream_beside(busy, 1, &beside);
    P.copy_beside = beside != 0;
    P.probes = 1;
    if (!P.copy) return set_err_msg("cannot create the copy-out streams", 4);
    if (getenv("GKM_TRACE"))
        fprintf(stderr, "gkmhip: copy-out streams of device %d: the copy stream %s the compute stream\n", device,
                P.copy_beside ? "runs beside" : "SHARES A QUEUE WITH");
    return 0;
}

void gkm_release_pipe_streams()
{
    std::lock_guard<std::mutex> lock(g_pipe_mutex);
    int caller = -1;
    (void)hipGetDevice(&caller);
    for (int d = 0; d < 64; d++) {
        PipeStreams &P = g_pipe[d];
        if (!P.compute) continue;
        (void)hipSetDevice(d);
        for (hipStream_t x : {P.compute, P.copy})
            if (x) (void)hipStreamDestroy(x);
        P = PipeStreams();
    }
    if (caller >= 0) (void)hipSetDevice(caller);
}


template <class Issue>
static hipError_t ship_pieces(const std::vector<RowPiece> &pieces, double *const stage[2], hipStream_t sd, double **rows,
                              int nthreads, Issue issue, double *t_wait, double *t_scatter,
                              std::vector<double> *ready_at = nullptr)
{
    auto now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const size_t NP = pieces.size();
    hipError_t e = NP ? issue((size_t)0) : hipSuccess;
    for (size_t q = 0; e == hipSuccess && q < NP; q++) {
        const double tw = now();
        e = hipStreamSynchronize(sd); /* piece q is in stage[q & 1] */
        if (e != hipSuccess) break;
        if (ready_at) ready_at->push_back(now());
        if (q + 1 < NP) e = issue(q + 1);
        const double ts = now();
        if (t_wait) *t_wait += ts - tw;
        const RowPiece &k = pieces[q];
        const double *src = stage[q & 1];
        auto work = [&](int t) {
            for (int r = k.r0 + t; r < k.r1; r += nthreads)
                memcpy(rows[r], src + (size_t)(r - k.r0) * k.r1, (size_t)(r + 1) * sizeof(double));
        };
        if (nthreads == 1 || k.r1 - k.r0 < 64) {
            for (int t = 0; t < nthreads; t++) work(t);
        } else {
            std::vector<std::thread> th;
            for (int t = 1; t < nthreads; t++) th.emplace_back(work, t);
            work(0);
            for (auto &x : th) x.join();
        }
        if (t_scatter) *t_scatter += now() - ts;
    }
    return e;
}

/* Whole matrix into caller-owned host rows (rows[a][0..a]) as a pipeline over row blocks of
 * about equal work: block k+1 is computed while block k travels device -> pinned staging ->
 * the caller's (pageable) rows.  G is device scratch of n x ld doubles.
 * part / nparts: this context handles every nparts-th block (several GPUs driven by one host
 * process, one context and thread each, all writing disjoint rows of the same host matrix);
 * with nparts > 1 the self norms come from a diagonal-band pass first, so that no device needs
 * another device's rows. */
static int gram_part_to_host_rows(gkmhip_ctx *ctx, double *G, int64_t ld, double **rows, int nthreads, int part,
                                  int nparts)
{
    if (!ctx || !G || !rows || ctx->n <= 0 || ld < ctx->n || nparts < 1 || part < 0 || part >= nparts)
        return set_err_msg("gkmhip_gram_to_host_rows: bad arguments", 2);
    HIPCHK(hipSetDevice(ctx->device));
    const int n = ctx->n;
    const size_t want = (size_t)64 << 20;
    double *stage[2];
    timespec ts_a, ts_b;
    clock_gettime(CLOCK_MONOTONIC, &ts_a);
    if (acquire_staging(want, stage, part)) return 4;
    clock_gettime(CLOCK_MONOTONIC, &ts_b);
    const double staging_ms = (ts_b.tv_sec - ts_a.tv_sec) * 1e3 + (ts_b.tv_nsec - ts_a.tv_nsec) * 1e-6; /* (first call: pinning 128 MB) */
    if (ctx->sq.ensure((size_t)n)) return 4;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 16) nthreads = 16;

    /* Row blocks (one Gram launch each) and copy pieces (one staging rectangle each).
     * Several devices: blocks of at most 1/(4 nparts) of the triangle's area, dealt round-robin, one piece each.
     * One device: every launch costs ~1 ms (ramp and drain: a wave lives ~0.6 ms), the copy-out of a block overlaps
     * the Gram kernels of the blocks behind it, and the LAST block's copy and scatter overlap nothing.  So: as few
     * blocks as the copy-out can keep up with.  With T the time the Gram kernels need for the whole triangle and S the
     * time the copy-out needs for all of it (`nthreads` host threads scatter the rows: the caller's -@, 1 in gkmQC's
     * default), a first block of 1 / (1 + S/T) of the area is shipped just when the rest has been computed; the rest
     * is cut the same way, down to a last block of ~5 %.  T and S come from what the previous call of the process
     * measured (g_ship), estimates before that.  16 threads, n = 10 000: 0.80 / 0.16 / 0.04 of the area -- 3 launches
     * instead of round 3's 6 halvings, 79.5 instead of 81.7 ms for the call (tools/boundary_ab.py; profiles/r4_boundary_ab*);
     * one thread: seven blocks from 0.36 down.  (GKM_EQUAL_BLOCKS=1 keeps the equal blocks of round 1, for A/B runs.) */
    std::vector<RowPiece> blocks, pieces;
    std::vector<int> block_of; /* piece -> block */
    const bool geometric = nparts == 1 && getenv("GKM_EQUAL_BLOCKS") == nullptr;
    /* (whole rows as ONE linear copy per piece instead of a pitched copy of the columns [0, r1) -- twice the bytes --
     * was measured in round 4: 102 instead of 96 ms on the same schedule) */
    const double total_area = (double)n * n / 2.0;
    const double area_cap = total_area / std::max(12, 4 * nparts);
    int index = 0;
    double left = total_area, target = total_area / 2.0;
    double ship_ratio; /* S / T */
    {
        std::lock_guard<std::mutex> lock(g_ship.m);
        const double bytes = total_area * 8.0;
        /* (before anything has been measured: one thread moves ~15-20 GB/s into pageable memory, sixteen ~60) */
        const double per_thread = g_ship.scatter_Bps_per_thread > 0 ? g_ship.scatter_Bps_per_thread : 15.0e9 / sqrt((double)nthreads);
        const double ship_Bps = std::min(per_thread * nthreads, g_ship.copy_Bps > 0 ? g_ship.copy_Bps : 50.0e9);
        const double cmp_rate = g_ship.cmp_per_s > 0 ? g_ship.cmp_per_s : 1.0e14;
        const double cmp = ctx->h_cum_n.empty() ? 0.0 : ctx->h_cum_n[(size_t)n] * ctx->h_cum_n[(size_t)n]; /* ~2 n_a n_j over j <= a */
        ship_ratio = cmp > 0 ? (bytes / ship_Bps) / (cmp / cmp_rate) : 0.5;
    }
    /* (1.3: a piece is copied, THEN scattered; only the copy of the next piece overlaps the scatter) */
    const double first_share = std::min(0.8, std::max(0.3, 1.0 / (1.0 + 1.3 * ship_ratio)));
    /* GKM_BLOCK_FRACTIONS="0.7,0.2" (experiments): the blocks' shares of the triangle's area, the last block takes the rest */
    std::vector<double> fractions;
    if (const char *bf = getenv("GKM_BLOCK_FRACTIONS"))
        for (const char *q = bf; *q;) {
            char *end = nullptr;
            const double v = strtod(q, &end);
            if (end == q) break;
            if (v > 0.0 && v < 1.0) fractions.push_back(v);
            q = *end ? end + 1 : end;
        }
    size_t fi = 0;
    for (int r0 = 0; r0 < n;) {
        int r1 = r0 + 1;
        if (geometric && !fractions.empty()) {
            target = fi < fractions.size() ? fractions[fi++] * total_area : left;
            while (r1 < n && ((double)(r1 + 1) * (r1 + 1) - (double)r0 * r0) / 2.0 <= target) r1++;
            if (fi > fractions.size() || n - r1 < 32) r1 = n;
            if (fi == fractions.size()) fi++; /* the next block is the last one */
            left -= ((double)r1 * r1 - (double)r0 * r0) / 2.0;
        } else if (geometric) {
            target = left <= 0.06 * total_area ? left : first_share * left; /* (the rest in one go) */
            while (r1 < n && ((double)(r1 + 1) * (r1 + 1) - (double)r0 * r0) / 2.0 <= target) r1++;
            if (n - r1 < 32) r1 = n;
            left -= ((double)r1 * r1 - (double)r0 * r0) / 2.0;
        } else {
            while (r1 < n && (size_t)(r1 + 1) * (size_t)(r1 + 1 - r0) * 8 <= want &&
                   ((double)(r1 + 1) * (r1 + 1) - (double)r0 * r0) / 2.0 <= area_cap)
                r1++;
        }
        if (index++ % nparts == part) {
            cut_pieces(r0, r1, want, pieces);
            block_of.resize(pieces.size(), (int)blocks.size());
            blocks.push_back({r0, r1});
        }
        r0 = r1;
    }
    if (blocks.empty()) return 0;
    const size_t B = blocks.size();
    const bool trace = getenv("GKM_TRACE") != nullptr;
    auto now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double t0 = now();
    double t_wait = 0, t_scatter = 0;
    /* Two streams, kept per device and PROVEN to run beside each other (pipe_streams): sc carries the blocks' compute
     * (tables, row planes, Gram kernel, untile, self norms, normalise: ~0.35 ms of small kernels per block boundary),
     * sd the device-to-host copies.  (Alternating the blocks' Gram kernels between two streams was measured in round 2:
     * 107 instead of 89 ms -- two tiles' worth of waves on a CU evict each other's packed rows from L1; moving the small
     * kernels off sc in round 4: see gram_launch.) */
    PipeStreams *ps = nullptr;
    if (pipe_streams(ctx->device, &ps)) return 4;
    const double streams_ms = now() - t0; /* (first call: the streams are created and probed) */
    const hipStream_t sc = ps->compute, sd = ps->copy;
    std::vector<hipEvent_t> done(B, nullptr);
    int rc = 0;
    std::vector<int> idx;
    if (nparts > 1) rc = gkmhip_self_norms(ctx, ctx->sq.p, sc);
    for (size_t b = 0; b < B && !rc; b++) { /* enqueue all the compute up front */
        idx.resize((size_t)(blocks[b].r1 - blocks[b].r0));
        for (size_t i = 0; i < idx.size(); i++) idx[i] = blocks[b].r0 + (int)i;
        rc = gkmhip_gram_rows(ctx, idx.data(), (int)idx.size(), 0, G, ld, nullptr, 0, sc);
        if (!rc) rc = normalize_rows(ctx, G, ld, blocks[b].r0, blocks[b].r1, ctx->sq.p, 0, sc, nparts > 1);
        if (!rc && hipEventCreateWithFlags(&done[b], hipEventDisableTiming) != hipSuccess) rc = 4;
        if (!rc && hipEventRecord(done[b], sc) != hipSuccess) rc = 4;
    }
    const size_t NP = pieces.size();
    auto issue = [&](size_t q) -> hipError_t {
        const RowPiece &k = pieces[q];
        hipError_t e = hipStreamWaitEvent(sd, done[(size_t)block_of[q]], 0);
        if (e != hipSuccess) return e;
        return hipMemcpy2DAsync(stage[q & 1], (size_t)k.r1 * 8, G + (size_t)k.r0 * ld, (size_t)ld * 8,
                                (size_t)k.r1 * 8, (size_t)(k.r1 - k.r0), hipMemcpyDeviceToHost, sd);
    };
    const double t_enq = now();
    std::vector<double> ready_at; /* (GKM_TRACE) when each piece had arrived in its staging buffer */
    const hipError_t e = rc ? hipErrorUnknown
                            : ship_pieces(pieces, stage, sd, rows, nthreads, issue, &t_wait, &t_scatter, trace ? &ready_at : nullptr);
    for (hipStream_t x : {sc, sd}) (void)hipStreamSynchronize(x);
    if (!rc && e == hipSuccess && nparts == 1 && t_scatter > 0) { /* what the next call's block schedule goes by */
        std::lock_guard<std::mutex> lock(g_ship.m);
        g_ship.scatter_Bps_per_thread = total_area * 8.0 / (t_scatter * 1e-3) / nthreads;
        const double whole = now() - t0;
        /* the Gram kernels' share of the call: everything but the last block's copy-out (an estimate is all it takes) */
        g_ship.cmp_per_s = ctx->h_cum_n[(size_t)n] * ctx->h_cum_n[(size_t)n] / (std::max(1.0, whole - 1.5) * 1e-3);
    }
    if (trace)
        fprintf(stderr, "gkmhip_gram_to_host_rows: %zu blocks (first share %.2f of what is left, %d threads), %zu pieces, pinned staging %.1f ms, streams %.1f ms, setup+enqueue %.1f ms (streams included), waiting for blocks %.1f ms, host scatter %.1f ms, total %.1f ms\n",
                B, first_share, nthreads, NP, staging_ms, streams_ms, t_enq - t0, t_wait, t_scatter, now() - t0);
    if (trace) {
        fprintf(stderr, "  pieces (rows, MB, in staging at ms):");
        for (size_t q = 0; q < ready_at.size(); q++)
            fprintf(stderr, " [%d-%d %.0f MB @%.1f]", pieces[q].r0, pieces[q].r1,
                    (double)pieces[q].r1 * (pieces[q].r1 - pieces[q].r0) * 8e-6, ready_at[q] - t0);
        fprintf(stderr, "\n");
    }
    for (auto ev : done)
        if (ev) (void)hipEventDestroy(ev);
    if (rc) return rc;
    if (e != hipSuccess) return set_err("gkmhip_gram_to_host_rows", e, __FILE__, __LINE__);
    return 0;
}

extern "C" int gkmhip_gram_to_host_rows(gkmhip_ctx *ctx, double *G, int64_t ld, double **rows, int nthreads)
{
    return gram_part_to_host_rows(ctx, G, ld, rows, nthreads, 0, 1);
}

extern "C" int gkmhip_gram_part_to_host_rows(gkmhip_ctx *ctx, double *G, int64_t ld, double **rows, int nthreads,
                                             int part, int nparts)
{
    return gram_part_to_host_rows(ctx, G, ld, rows, nthreads, part, nparts);
}

/* ------------------------------------------------------- memory helpers */
extern "C" void *gkmhip_malloc(int device, size_t bytes)
{
    void *p = nullptr;
    if (hipSetDevice(device) != hipSuccess) { g_err = "hipSetDevice failed"; return nullptr; }
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) { set_err("hipMalloc", e, __FILE__, __LINE__); return nullptr; }
    return p;
}
extern "C" void gkmhip_free(void *p) { if (p) (void)hipFree(p); }
extern "C" int gkmhip_memcpy_d2h(void *dst, const void *src, size_t bytes)
{
    HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int gkmhip_memcpy_h2d(void *dst, const void *src, size_t bytes)
{
    HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}
extern "C" int gkmhip_sync(void *stream)
{
    if (stream) HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    else HIPCHK(hipDeviceSynchronize());
    return 0;
}

/* Lower triangle of a device matrix into caller-owned host rows, through the same piece pipeline as the drop-in
 * call (cut_pieces / ship_pieces): the D2H DMA of piece q+1 overlaps the host scatter of piece q. */
extern "C" int gkmhip_copy_lower_to_rows(gkmhip_ctx *ctx, const double *K, int64_t ld, int n, double **rows,
                                         int nthreads)
{
    if (!ctx || !K || !rows || n <= 0) return set_err_msg("gkmhip_copy_lower_to_rows: bad arguments", 2);
    HIPCHK(hipSetDevice(ctx->device));
    const size_t want = (size_t)64 << 20;
    double *stage[2];
    if (acquire_staging(want, stage, 0)) return 4;
    nthreads = std::min(std::max(nthreads, 1), 16);
    StreamSet ss;
    if (ss.create(1)) return 4;
    const hipStream_t sd = ss.s[0];
    std::vector<RowPiece> pieces;
    cut_pieces(0, n, want, pieces);
    auto issue = [&](size_t q) -> hipError_t {
        const RowPiece &k = pieces[q];
        return hipMemcpy2DAsync(stage[q & 1], (size_t)k.r1 * 8, K + (size_t)k.r0 * ld, (size_t)ld * 8,
                                (size_t)k.r1 * 8, (size_t)(k.r1 - k.r0), hipMemcpyDeviceToHost, sd);
    };
    const hipError_t e = ship_pieces(pieces, stage, sd, rows, nthreads, issue, nullptr, nullptr);
    (void)hipStreamSynchronize(sd);
    if (e != hipSuccess) return set_err("copy_lower_to_rows", e, __FILE__, __LINE__);
    return 0;
}

extern "C" double gkmhip_last_kernel_ms(gkmhip_ctx *ctx)
{
    if (!ctx || !ctx->ev_valid) return -1.0;
    if (hipEventSynchronize(ctx->ev1) != hipSuccess) return -1.0;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) != hipSuccess) return -1.0;
    return (double)ms;
}
extern "C" double gkmhip_last_comparisons(gkmhip_ctx *ctx) { return ctx ? ctx->last_comparisons : 0.0; }
extern "C" const char *gkmhip_last_kernel_name(gkmhip_ctx *ctx) { return ctx ? ctx->last_kernel : "none"; }
