/*
 * gkm_context.hip -- MI355X (gfx950) device layer of the gkm kernel-matrix path: context, upload and the per-sequence
 * device tables.  Implements the first part of include/gkm_hip.h (see gkm_internal.h for the other translation units).
 *
 * Kernels
 *   k_build_sb        column-strand bit-plane tables, strided layout (gkm_bitslice.h)
 *   k_pack_strands    both strands of every sequence, 16 bases per word (the hit path's column side)
 *   k_pack_lmers      per-l-mer tables of the general kernel (k_gram_direct)
 */
#include "gkm_gram_bitslice.h" /* (BS_DU pads the SB tables) */

/* ------------------------------------------------------------------ errors */
static thread_local std::string g_err;

int gkm_set_err(const char *what, hipError_t e, const char *file, int line)
{
    char buf[512];
    snprintf(buf, sizeof buf, "%s: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_err = buf;
    return 100 + (int)e;
}
int gkm_set_err_msg(const std::string &m, int code)
{
    g_err = m;
    return code;
}
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

/* Pinned staging for device-to-host copies, kept for the life of the process: the pipeline
 * calls the boundary once per peak subset (20x per run, bin/gkmqc.py:341-343) and pinning
 * 2 x 64 MB costs ~30 ms per call otherwise.  gkmhip_release_host_cache() frees it. */
static std::mutex g_stage_mutex;
static double *g_stage[STAGE_SLOTS][2];
static size_t g_stage_bytes[STAGE_SLOTS];

int acquire_staging(size_t want, double **out, int slot)
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
static std::mutex g_pin_mutex;
static std::vector<PinBuf *> g_pin;

void pin_release(void *ud) { ((PinBuf *)ud)->in_use.store(0, std::memory_order_release); }

PinBuf *pin_acquire(size_t bytes)
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
    ctx->codes.release(); ctx->wd.release(); ctx->wdc.release(); ctx->off.release(); ctx->lmoff.release();
    ctx->len.release(); ctx->lmf.release(); ctx->sb.release(); ctx->colpk.release(); ctx->postab.release();
    for (auto &scr : ctx->scratch) scr.release();
    ctx->sq.release();
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (auto &pr : ctx->tl_pairs) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
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
    /* CYCLIC (round 5): base i of the image is base i mod T of the strand, as in the bit planes of the counting loop, so
     * that a 16-base window read near the strand's end holds the l-mers the counting loop compared there (the
     * same-length variant's trips evaluate five consecutive windows from one such read) */
    for (int x = threadIdx.x; x < pkw; x += blockDim.x)
        colpk[((size_t)s * pkw + x) * 2 + strand] = gkmbs::pk_word_cyclic(seq, T, strand, x);
}

/* one workgroup per sequence: its positional weights as the hit path wants them in LDS (layout: gkm_gram_bitslice.h
 * POSTAB_PAD): byte PAD + L - 1 + p = wt[p] = wd[|n/2 - p|] for the l-mers p < n (libgkm.c:912-925), L - 1 zero bytes either
 * side -- a window that runs over the end of the strand reads a zero --, and outside those the weights of the windows past
 * the end (l-mer j again: wt[j] behind, wt[n-1-j] in front for the reverse strand), zeros to the end of the ptw words */
__global__ void k_build_postab(const int64_t *__restrict__ off, int L, const uint8_t *__restrict__ wd, int ptw,
                               uint32_t *__restrict__ postab)
{
    const int s = blockIdx.x;
    const int T = (int)(off[s + 1] - off[s]);
    const int n = T - L + 1, PAD = (int)POSTAB_PAD;
    auto wt = [&](int p) -> uint32_t {
        if (p < 0 || p >= n) return 0u;
        const int dd = n / 2 - p;
        return wd[dd < 0 ? -dd : dd];
    };
    for (int x = threadIdx.x; x < ptw; x += blockDim.x) {
        uint32_t v = 0u;
        for (int b = 0; b < 4; b++) {
            const int i = x * 4 + b;          /* byte index */
            const int p = i - PAD - (L - 1);  /* l-mer position the byte stands for */
            uint32_t w = 0u;
            if (i < PAD) w = wt(n - 1 - (PAD - 1 - i));          /* in front: index PAD-1-j is l-mer n-1-j */
            else if (p >= 0 && p < n) w = wt(p);
            else if (p >= T && p < T + PAD) w = wt(p - T);       /* behind the zeros: position T + j is l-mer j */
            v |= w << (8 * b);
        }
        postab[(size_t)s * ptw + x] = v;
    }
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

/* ---------------------------------------------------------- host: upload */
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
    ctx->have_postab = false;
    ctx->n = 0;
    ctx->weighted = weighted;
    ctx->h_len.resize((size_t)n);
    ctx->h_lmoff.resize((size_t)n + 1);
    ctx->h_cum_n.resize((size_t)n + 1);
    ctx->h_lmoff[0] = 0;
    ctx->h_cum_n[0] = 0.0;
    ctx->maxlen = 0;
    ctx->minlen = 1 << 30;
    for (int i = 0; i < n; i++) {
        const int64_t len = offsets[i + 1] - offsets[i];
        if (len < L) return set_err_msg("sequence " + std::to_string(i) + " is shorter than L", 3);
        if (len > 2047) return set_err_msg("sequence longer than 2047", 3);
        ctx->h_len[(size_t)i] = (int)len;
        ctx->h_lmoff[(size_t)i + 1] = ctx->h_lmoff[(size_t)i] + (len - L + 1);
        ctx->h_cum_n[(size_t)i + 1] = ctx->h_cum_n[(size_t)i] + (double)(len - L + 1);
        ctx->maxlen = std::max(ctx->maxlen, (int)len);
        ctx->minlen = std::min(ctx->minlen, (int)len);
    }
    ctx->n = n;
    /* What share of the l-mer pairs of THESE sequences lies within d mismatches?  `auto` chooses between the bit-sliced and
     * the general kernel by that share (gkm_gram.hip auto_takes_bitslice); the iid formula is exact for random ACGT and
     * too low for repeat-rich or low-complexity input, so 8 192 l-mer pairs are sampled here on the host -- random
     * sequence pairs, random positions, either strand of the second one, a fixed generator: ~0.2 ms, and the choice of
     * kernel never changes a result. */
    {
        uint64_t st = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
        auto next = [&]() { /* splitmix64 */
            uint64_t z = (st += 0x9E3779B97F4A7C15ull);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            return z ^ (z >> 31);
        };
        const int samples = 8192;
        int within = 0;
        for (int t = 0; t < samples; t++) {
            const int a = (int)(next() % (uint64_t)n), b = (int)(next() % (uint64_t)n);
            const int la = ctx->h_len[(size_t)a], lb = ctx->h_len[(size_t)b];
            const int p = (int)(next() % (uint64_t)(la - L + 1)), q = (int)(next() % (uint64_t)(lb - L + 1));
            const bool rc = (next() & 1u) != 0;
            const uint8_t *sa = codes + offsets[a] + p, *sb = codes + offsets[b];
            int mm = 0;
            for (int i = 0; i < L && mm <= ctx->d; i++) {
                const uint8_t cb = rc ? (uint8_t)(3 - sb[lb - 1 - (q + i)]) : sb[q + i]; /* reverse complement: libgkm.c:877-888 */
                mm += sa[i] != cb;
            }
            within += mm <= ctx->d;
        }
        ctx->sampled_hit_share = (double)within / samples;
    }
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
    std::vector<uint8_t> wdc; /* (source of an asynchronous copy: alive until the wait at the end of this function) */
    {
        /* the centred copy: byte centre + s = wd[|s|] for |s| < wd_len, 0 outside; 4 bytes of slack below and 12 above
         * (k_gram_bitslice reads five consecutive bytes as an aligned pair of words) */
        const int B = ctx->wd_len - 1, centre = B + 4, bytes = ((centre + B + 1 + 12 + 3) / 4) * 4;
        wdc.assign((size_t)bytes, 0);
        for (int sd = -B; sd <= B; sd++) wdc[(size_t)(centre + sd)] = weighted ? wdist[sd < 0 ? -sd : sd] : (uint8_t)1;
        if (ctx->wdc.ensure((size_t)bytes / 4)) {
            (void)hipStreamSynchronize(stream); /* (copies from the caller's arrays are in flight) */
            return 4;
        }
        HIPCHK(hipMemcpyAsync(ctx->wdc.p, wdc.data(), (size_t)bytes, hipMemcpyHostToDevice, stream));
        ctx->wdc_words = bytes / 4;
        ctx->wdc_centre = centre;
    }
    /* The per-sequence device tables are built HERE, not at the first launch: callers may alternate launches between
     * two streams (the scratch slots of gkmhip_set_scratch_slot; gkm_multi.hip and bench.py did until round 5), and a
     * table built by the first launch on one stream was read by the second launch on the other stream before it was
     * complete (found when the host stopped waiting for its uploads: the config-4 stand-in through two contexts differed
     * in a few hundred rows). */
    int rc = 0;
    if (bitslice_serves(ctx)) rc = ensure_sb(ctx, 10, stream, false) || ensure_colpk(ctx, stream, false) || ensure_postab(ctx, stream, false);
    else rc = ensure_lmers(ctx, stream, false);
    /* ONE wait for everything enqueued above (round 5: five before).  The sources of the copies are the caller's (pageable)
     * arrays and a local vector: an asynchronous copy of more than a few KB may still be reading them after this call has
     * returned, so the upload is finished here, error or not (3 MB, once per matrix); and the tables are complete before a
     * launch on any other stream reads them. */
    const hipError_t done = hipStreamSynchronize(stream);
    if (rc) return 4;
    HIPCHK(done);
    return 0;
}

int ensure_lmers(gkmhip_ctx *ctx, hipStream_t stream, bool wait)
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
    if (wait) HIPCHK(hipStreamSynchronize(stream)); /* complete before any OTHER stream may read the table */
    ctx->have_lmers = true;
    return 0;
}

int ensure_colpk(gkmhip_ctx *ctx, hipStream_t stream, bool wait)
{
    if (ctx->have_colpk) return 0;
    /* one word more than the bases need: the hit path reads words q/16 and q/16 + 1 */
    const int pkw = (ctx->maxlen + 15) / 16 + 1;
    if (ctx->colpk.ensure((size_t)ctx->n * 2 * (size_t)pkw)) return 4;
    hipLaunchKernelGGL(k_pack_strands, dim3((unsigned)ctx->n * 2), dim3(64), 0, stream, ctx->codes.p, ctx->off.p, pkw,
                       ctx->colpk.p);
    HIPCHK(hipGetLastError());
    ctx->pkw = pkw;
    if (wait) HIPCHK(hipStreamSynchronize(stream)); /* complete before any OTHER stream may read the table */
    ctx->have_colpk = true;
    return 0;
}

int ensure_postab(gkmhip_ctx *ctx, hipStream_t stream, bool wait)
{
    if (ctx->have_postab) return 0;
    /* (+ 8: the same-length variant reads the bytes as aligned pairs of words around index .. index + 4) */
    const int ptw = ((int)POSTAB_PAD + ctx->L - 1 + ctx->maxlen + (int)POSTAB_PAD + 8 + 3) / 4;
    if (ctx->postab.ensure((size_t)ctx->n * (size_t)ptw)) return 4;
    hipLaunchKernelGGL(k_build_postab, dim3((unsigned)ctx->n), dim3(64), 0, stream, ctx->off.p, ctx->L, ctx->wd.p, ptw,
                       ctx->postab.p);
    HIPCHK(hipGetLastError());
    ctx->ptw = ptw;
    if (wait) HIPCHK(hipStreamSynchronize(stream)); /* complete before any OTHER stream may read the table */
    ctx->have_postab = true;
    return 0;
}

int ensure_sb(gkmhip_ctx *ctx, int W, hipStream_t stream, bool wait)
{
    if (ctx->have_sb && ctx->sb_W == W) return 0;
    const int xw = ((ctx->maxlen + W + 2 * BS_DU + 15) / 16) * 16;
    if (ctx->sb.ensure((size_t)ctx->n * 2 * 2 * (size_t)xw)) return 4;
    hipLaunchKernelGGL(k_build_sb, dim3((unsigned)ctx->n * 2, 2), dim3(256), 0, stream, ctx->codes.p, ctx->off.p,
                       W, ctx->L, xw, ctx->sb.p);
    HIPCHK(hipGetLastError());
    ctx->sb_xw = xw;
    ctx->sb_W = W;
    if (wait) HIPCHK(hipStreamSynchronize(stream)); /* complete before any OTHER stream may read the table */
    ctx->have_sb = true;
    return 0;
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

int gkm_launch_events(gkmhip_ctx *ctx, hipEvent_t *e0, hipEvent_t *e1)
{
    if (ctx->tl_on) {
        if (ctx->tl_used == ctx->tl_pairs.size()) {
            hipEvent_t a = nullptr, b = nullptr;
            HIPCHK(hipEventCreate(&a));
            if (hipEventCreate(&b) != hipSuccess) {
                (void)hipEventDestroy(a);
                return set_err_msg("hipEventCreate failed", 4);
            }
            ctx->tl_pairs.emplace_back(a, b);
        }
        *e0 = ctx->tl_pairs[ctx->tl_used].first;
        *e1 = ctx->tl_pairs[ctx->tl_used].second;
        ctx->tl_used++;
    } else {
        *e0 = ctx->ev0;
        *e1 = ctx->ev1;
    }
    ctx->last_e0 = *e0;
    ctx->last_e1 = *e1;
    return 0;
}

extern "C" int gkmhip_kernel_timeline(gkmhip_ctx *ctx, int on)
{
    if (!ctx) return set_err_msg("gkmhip_kernel_timeline: bad arguments", 2);
    ctx->tl_on = on != 0;
    ctx->tl_used = 0;
    return 0;
}

extern "C" double gkmhip_kernel_timeline_ms(gkmhip_ctx *ctx, int *launches)
{
    if (launches) *launches = 0;
    if (!ctx) return -1.0;
    double sum = 0.0;
    for (size_t i = 0; i < ctx->tl_used; i++) {
        float ms = 0.f;
        if (hipEventSynchronize(ctx->tl_pairs[i].second) != hipSuccess ||
            hipEventElapsedTime(&ms, ctx->tl_pairs[i].first, ctx->tl_pairs[i].second) != hipSuccess)
            return -1.0;
        sum += (double)ms;
    }
    if (launches) *launches = (int)ctx->tl_used;
    return sum;
}

extern "C" int gkmhip_kernel_timeline_spans(gkmhip_ctx *ctx, double *out, int cap)
{
    if (!ctx || !out || (size_t)cap < 2 * ctx->tl_used) return 0;
    for (size_t i = 0; i < ctx->tl_used; i++) {
        float a = 0.f, b = 0.f;
        if (hipEventSynchronize(ctx->tl_pairs[i].second) != hipSuccess ||
            hipEventElapsedTime(&a, ctx->tl_pairs[0].first, ctx->tl_pairs[i].first) != hipSuccess ||
            hipEventElapsedTime(&b, ctx->tl_pairs[0].first, ctx->tl_pairs[i].second) != hipSuccess)
            return 0;
        out[2 * i] = (double)a;
        out[2 * i + 1] = (double)b;
    }
    return (int)(2 * ctx->tl_used);
}

extern "C" double gkmhip_last_kernel_ms(gkmhip_ctx *ctx)
{
    if (!ctx || !ctx->ev_valid || !ctx->last_e1) return -1.0;
    if (hipEventSynchronize(ctx->last_e1) != hipSuccess) return -1.0;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->last_e0, ctx->last_e1) != hipSuccess) return -1.0;
    return (double)ms;
}
extern "C" double gkmhip_last_comparisons(gkmhip_ctx *ctx) { return ctx ? ctx->last_comparisons : 0.0; }
extern "C" const char *gkmhip_last_kernel_name(gkmhip_ctx *ctx) { return ctx ? ctx->last_kernel : "none"; }
