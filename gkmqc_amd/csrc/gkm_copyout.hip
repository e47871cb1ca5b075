/*
 * gkm_copyout.hip -- device matrix -> the caller's (pageable) host rows, the last stage of gkm_main_pywrapper
 * (src/gkmkern_pylib.c:83,187-190,218-221 write the same cells from the row threads): row blocks computed one after the
 * other, each copied in staging-sized pieces on a second stream PROVEN to run beside the compute stream, scattered
 * by the caller's `nthreads` host threads while the next piece travels.
 */
#include "gkm_internal.h"

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

/* Keeps `stream` busy for `microseconds` with ONE wave (k_spin): the multi-GPU path puts it in front of chunk c+1's launch
 * so that the transfer of chunk c, which becomes runnable on another stream at the same moment, has its workgroups on the
 * device before the Gram kernel takes every wave slot (gkm_multi.hip; tools/collective_beside_probe.py). */
extern "C" int gkmhip_pause_stream(void *stream, int microseconds)
{
    if (microseconds <= 0) return 0;
    if (microseconds > 10000) return set_err_msg("gkmhip_pause_stream: more than 10 ms", 2);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)microseconds * 100, (unsigned *)nullptr); /* 100 MHz */
    HIPCHK(hipGetLastError());
    return 0;
}

/* Measurement only (tools/collective_beside_probe.py): a device-to-device copy by `blocks` workgroups of `threads` threads,
 * the launch shape of a collective's kernel (a few large, persistent workgroups), to see on ONE GPU whether such
 * workgroups get onto the device while a Gram kernel holds every wave slot. */
__global__ void k_probe_copy(uint4 *dst, const uint4 *src, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

extern "C" int gkmhip_probe_copy(void *dst, const void *src, size_t bytes, int blocks, int threads, void *stream)
{
    if (!dst || !src || blocks < 1 || threads < 64 || threads > 1024 || (bytes & 15)) return set_err_msg("gkmhip_probe_copy: bad arguments", 2);
    hipLaunchKernelGGL(k_probe_copy, dim3((unsigned)blocks), dim3((unsigned)threads), 0, (hipStream_t)stream, (uint4 *)dst,
                       (const uint4 *)src, bytes / 16);
    HIPCHK(hipGetLastError());
    return 0;
}

/* Measurement only: `blocks` workgroups of `threads` threads that do nothing but watch the clock for `microseconds` -- a
 * stand-in for a collective's workgroups waiting for their peers while they hold their wave slots. */
extern "C" int gkmhip_probe_spin(int blocks, int threads, int microseconds, void *stream)
{
    if (blocks < 1 || threads < 64 || threads > 1024 || microseconds < 1 || microseconds > 100000)
        return set_err_msg("gkmhip_probe_spin: bad arguments", 2);
    hipLaunchKernelGGL(k_spin, dim3((unsigned)blocks), dim3((unsigned)threads), 0, (hipStream_t)stream, (long long)microseconds * 100,
                       (unsigned *)nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}

/* what the drop-in call's copy-out pipeline measured last time (gram_part_to_host_rows cuts its row blocks by it) */
static struct {
    std::mutex m;
    double scatter_Bps_per_thread = 0, cmp_per_s = 0;
} g_ship;

struct PipeStreams {
    hipStream_t compute = nullptr, copy = nullptr;
    int probes = 0;
    bool copy_beside = false; /* proven to run beside `compute` */
    int users = 0;            /* calls that hold the pair right now (gkm_release_pipe_streams leaves those alone) */
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
/* CU mask that leaves `reserve` compute units to other streams, the same number in every XCD whichever way the mask's bits
 * are numbered (XCD = bit / 32 or bit % 8): the j-th reserved CU of XCD k is bit 32 k + (k + 8 j) % 32.  A mask that takes
 * its CUs from ONE XCD slows a machine-filling kernel by 50 % (its workgroups are dealt to the XCDs round robin:
 * profiles/r4_overlap_probe.txt). */
static std::vector<uint32_t> mask_reserving(int reserve)
{
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return {};
    const int cus = prop.multiProcessorCount;
    if (cus != 256 || reserve <= 0 || reserve % 8 || reserve > 32) return {}; /* (laid out for the MI355X's 8 x 32 CUs) */
    std::vector<uint32_t> m((size_t)cus / 32, 0xFFFFFFFFu);
    for (int j = 0; j < reserve / 8; j++)
        for (int k = 0; k < 8; k++) m[(size_t)k] &= ~(1u << ((k + 8 * j) % 32));
    return m;
}

static void *stream_beside(void *const *busy, int nbusy, int *beside, const int *priority, int reserve_cus = 0)
{
    const std::vector<uint32_t> cumask = mask_reserving(reserve_cus);
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
        const hipError_t e = !cumask.empty() ? hipExtStreamCreateWithCUMask(&c, (uint32_t)cumask.size(), cumask.data())
                             : priority      ? hipStreamCreateWithPriority(&c, hipStreamNonBlocking, *priority)
                                             : hipStreamCreateWithFlags(&c, hipStreamNonBlocking);
        if (e != hipSuccess) break;
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

extern "C" void *gkmhip_create_stream_beside(void *const *busy, int nbusy, int *beside)
{
    return stream_beside(busy, nbusy, beside, nullptr);
}

extern "C" void *gkmhip_create_stream_beside_prio(void *const *busy, int nbusy, int *beside, int priority)
{
    return stream_beside(busy, nbusy, beside, &priority);
}

extern "C" void *gkmhip_create_stream_reserving(void *const *busy, int nbusy, int *beside, int reserve_cus)
{
    if (mask_reserving(reserve_cus).empty()) return nullptr;
    return stream_beside(busy, nbusy, beside, nullptr, reserve_cus);
}

/* The device's pair of streams, created and probed at the first call; the caller holds it until pipe_streams_done()
 * (a concurrent gkmhip_release_host_cache() must not destroy streams a copy-out is using). */
static int pipe_streams(int device, PipeStreams **out)
{
    if (device < 0 || device >= 64) return set_err_msg("device ordinal out of range", 2);
    std::lock_guard<std::mutex> lock(g_pipe_mutex);
    PipeStreams &P = g_pipe[device];
    *out = &P;
    if (P.compute && P.copy) {
        P.users++;
        return 0;
    }
    if (P.compute) (void)hipStreamDestroy(P.compute); /* (a half-built entry of an earlier, failed call) */
    P = PipeStreams();
    HIPCHK(hipStreamCreateWithFlags(&P.compute, hipStreamNonBlocking));
    void *busy[1] = {P.compute};
    int beside = 0;
    P.copy = (hipStream_t)gkmhip_create_stream_beside(busy, 1, &beside);
    P.copy_beside = beside != 0;
    P.probes = 1;
    if (!P.copy) { /* nothing half-built stays behind: the next call starts over */
        (void)hipStreamDestroy(P.compute);
        P = PipeStreams();
        return set_err_msg("cannot create the copy-out streams", 4);
    }
    if (getenv("GKM_TRACE"))
        fprintf(stderr, "gkmhip: copy-out streams of device %d: the copy stream %s the compute stream\n", device,
                P.copy_beside ? "runs beside" : "SHARES A QUEUE WITH");
    P.users++;
    return 0;
}

static void pipe_streams_done(PipeStreams *P)
{
    std::lock_guard<std::mutex> lock(g_pipe_mutex);
    if (P->users > 0) P->users--;
}

void gkm_release_pipe_streams()
{
    std::lock_guard<std::mutex> lock(g_pipe_mutex);
    int caller = -1;
    (void)hipGetDevice(&caller);
    for (int d = 0; d < 64; d++) {
        PipeStreams &P = g_pipe[d];
        if (!P.compute || P.users > 0) continue; /* (in use by a copy-out right now: left for the next release) */
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
        const double ship_Bps = std::min(per_thread * nthreads, 50.0e9); /* (the device-to-host copy itself: ~50 GB/s) */
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
    struct Holder { /* hands the pair back on every path out of this function */
        PipeStreams *p;
        ~Holder() { pipe_streams_done(p); }
    } hold{ps};
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
