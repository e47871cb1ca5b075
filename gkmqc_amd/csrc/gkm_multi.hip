/*
 * gkm_multi.hip -- the Gram matrix on several GPUs of one node, driven by ONE host process
 * (SURVEY.md §8(e)): one host thread and one gkmhip_ctx per device, rows sharded by folded row
 * blocks (gkm_shard.h), every rank's row slabs all-gathered over xGMI, then every device
 * un-permutes and normalises its own copy of the whole matrix -- LIBSVM (or the GPU-resident
 * C-SVC of gkm_svm.hip) consumes the whole matrix, which is why a collective is needed at all.
 *
 * Transport of the all-gather:
 *   rccl  ncclCommInitAll + ncclAllGather (RCCL is resolved with dlopen at first use, so that the
 *         single-GPU drop-in call has no dependency on it; inside a PyTorch process the RCCL that
 *         torch already loaded is the one found).  Used whenever the contexts sit on distinct
 *         devices.  Communicators are kept for the life of the process (bin/gkmqc.py calls the
 *         kernel once per peak subset, 20x per run).
 *   p2p   every rank pulls its peers' slabs with hipMemcpyPeerAsync.  Used when several contexts
 *         share a device (the one-GPU rehearsal of the N > 1 path: RCCL refuses duplicate devices)
 *         or when RCCL cannot be loaded.  GKM_ALLGATHER=rccl|p2p forces one.
 * What travels are PACKED slabs (gkm_shard.h): row a as a + 1 doubles -- only j <= a is ever read -- n^2 / (2G)
 * doubles per rank and matrix, half of what full-width rows cost (round 3).
 * The chunks of a rank follow each other on ONE compute stream; the transfer of chunk c runs on a second stream and
 * overlaps the kernel of chunk c+1 (compute_streams() below has the measurement that took the second compute stream away).  Integer profiles are placement-independent,
 * so the assembled matrix is bit-identical to the single-GPU one for any number of devices.
 */
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gkm_hip.h"
#include "gkm_shard.h"

/* ---- the few RCCL entry points used, bound at run time (signatures: rccl/rccl.h) ---- */
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;                  /* ncclSuccess == 0 */
enum { GKM_NCCL_FLOAT64 = 8 };             /* ncclDataType_t: ncclDouble / ncclFloat64 */
typedef ncclResult_t (*fn_CommInitAll)(ncclComm_t *, int, const int *);
typedef ncclResult_t (*fn_CommDestroy)(ncclComm_t);
typedef ncclResult_t (*fn_CommAbort)(ncclComm_t);
typedef ncclResult_t (*fn_AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t);
typedef const char *(*fn_GetErrorString)(ncclResult_t);

namespace {

struct Rccl {
    void *handle = nullptr;
    fn_CommInitAll CommInitAll = nullptr;
    fn_CommDestroy CommDestroy = nullptr;
    fn_CommAbort CommAbort = nullptr; /* optional: frees a communicator whose collective never completed */
    fn_AllGather AllGather = nullptr;
    fn_GetErrorString GetErrorString = nullptr;
    std::vector<int> devs;         /* device list of the cached clique */
    std::vector<ncclComm_t> comms;
    std::string why;               /* why it is unavailable */
};
Rccl g_rccl;
std::mutex g_rccl_mutex;
std::string g_transport = "none";

bool rccl_load()
{
    if (g_rccl.AllGather) return true;
    if (!g_rccl.why.empty()) return false;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        g_rccl.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) { g_rccl.why = std::string("cannot load RCCL: ") + dlerror(); return false; }
    g_rccl.CommInitAll = (fn_CommInitAll)dlsym(g_rccl.handle, "ncclCommInitAll");
    g_rccl.CommDestroy = (fn_CommDestroy)dlsym(g_rccl.handle, "ncclCommDestroy");
    g_rccl.CommAbort = (fn_CommAbort)dlsym(g_rccl.handle, "ncclCommAbort");
    g_rccl.AllGather = (fn_AllGather)dlsym(g_rccl.handle, "ncclAllGather");
    g_rccl.GetErrorString = (fn_GetErrorString)dlsym(g_rccl.handle, "ncclGetErrorString");
    if (!g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.GetErrorString) {
        g_rccl.AllGather = nullptr;
        g_rccl.why = "RCCL library lacks ncclCommInitAll / ncclAllGather";
        return false;
    }
    return true;
}

void rccl_drop_comms(bool abort = false)
{
    for (ncclComm_t c : g_rccl.comms)
        if (c) (void)((abort && g_rccl.CommAbort) ? g_rccl.CommAbort(c) : g_rccl.CommDestroy(c));
    g_rccl.comms.clear();
    g_rccl.devs.clear();
}

/* communicators for this device list (created once, reused while the list stays the same) */
bool rccl_comms_for(const std::vector<int> &devs, std::string &err)
{
    if (!rccl_load()) { err = g_rccl.why; return false; }
    if (g_rccl.devs == devs && g_rccl.comms.size() == devs.size()) return true;
    rccl_drop_comms();
    g_rccl.comms.assign(devs.size(), nullptr);
    const ncclResult_t r = g_rccl.CommInitAll(g_rccl.comms.data(), (int)devs.size(), devs.data());
    if (r != 0) {
        err = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r);
        g_rccl.comms.clear();
        return false;
    }
    g_rccl.devs = devs;
    return true;
}

int fail_with(const std::string &msg, int code)
{
    gkmhip_set_error_message(msg.c_str());
    return code;
}

/* All threads of one call meet here; reusable.  wait(flag) returns the OR of the flags the threads brought
 * to THIS meeting -- the same value in every thread, which is what a decision to enter a collective needs. */
class HostBarrier {
public:
    explicit HostBarrier(int n) : n_(n) {}
    bool wait(bool flag = false)
    {
        std::unique_lock<std::mutex> lk(m_);
        const int gen = gen_;
        acc_ = acc_ || flag;
        if (++count_ == n_) {
            count_ = 0;
            result_ = acc_;
            acc_ = false;
            gen_++;
            cv_.notify_all();
            return result_;
        }
        cv_.wait(lk, [&] { return gen != gen_; });
        return result_;
    }

private:
    std::mutex m_;
    std::condition_variable cv_;
    int n_, count_ = 0, gen_ = 0;
    bool acc_ = false, result_ = false;
};

/* Everything a rank needs besides its context, kept from call to call (bin/gkmqc.py asks for ~20 matrices of the
 * same size per run): the slab of this rank's rows, the gathered slabs of all ranks, the gather index, the self
 * norms, the compute and the transfer stream and the events.  hipMalloc / hipFree synchronise the whole device, so a call that allocated
 * ~1.7 GB and freed it again paid for that beside a ~10 ms kernel on 8 GPUs.  Keyed by (device, n, ranks, chunks);
 * rebuilt when any of them changes, freed by gkmhip_release_comms(). */
struct RankCache {
    int dev = -1, n = 0, G = 0, chunks = 0, nsk = 0;
    int64_t pe = 0; /* doubles per (packed) chunk slab */
    double *slab = nullptr, *gathered = nullptr, *sq = nullptr;
    int64_t *d_slot = nullptr;
    hipStream_t sk[2] = {nullptr, nullptr}, sc = nullptr;
    std::vector<hipEvent_t> ready;          /* [chunk] slab complete (no timing: waited for by peers) */
    std::vector<hipEvent_t> k0, k1, a0, a1; /* [chunk] timing: kernel of the chunk, transfer of the chunk */
    hipEvent_t n0 = nullptr, n1 = nullptr;  /* timing: un-permute + normalise */
};
RankCache g_cache[64];
std::atomic<long> g_allocs{0}; /* hipMalloc calls made by this file (tests: the second call of a kind makes none) */

void cache_release(RankCache &R)
{
    if (R.dev < 0) return;
    (void)hipSetDevice(R.dev);
    if (R.slab) (void)hipFree(R.slab);
    if (R.gathered) (void)hipFree(R.gathered);
    if (R.sq) (void)hipFree(R.sq);
    if (R.d_slot) (void)hipFree(R.d_slot);
    for (auto *v : {&R.ready, &R.k0, &R.k1, &R.a0, &R.a1}) {
        for (hipEvent_t e : *v)
            if (e) (void)hipEventDestroy(e);
        v->clear();
    }
    if (R.n0) (void)hipEventDestroy(R.n0);
    if (R.n1) (void)hipEventDestroy(R.n1);
    for (int i = 0; i < 2; i++)
        if (R.sk[i]) (void)hipStreamDestroy(R.sk[i]);
    if (R.sc) (void)hipStreamDestroy(R.sc);
    R = RankCache();
}

/* How many compute streams a rank's chunk launches alternate between: ONE.  Rounds 2-5 used two ("the kernel of chunk c+1
 * overlaps the drain of chunk c"), and round 5 measured what that does (tools/rank_alone.py prints when each chunk ran,
 * profiles/r5_rank_alone_c2_two_streams.txt, r5_rank_alone_stream_policies.txt): two launches on two streams run CONCURRENTLY, workgroup by workgroup, and complete
 * together -- chunk 0 of 2 ended at 8.63 ms of a rank's 8.88 -- so nothing of chunk 0's transfer hid behind chunk 1's
 * kernel, which is the only reason to cut a rank's rows into chunks.  (Stream priorities do not repair it: the
 * higher-priority launch gets ~60 % of the device, and its small kernels that follow ended with the OTHER launch in
 * most runs.)  On one stream chunk 0 is complete at 4.5 of 9.0 ms; the drain that is no longer overlapped costs 0.2-0.3 ms
 * per chunk boundary.  GKM_MULTI_STREAMS=two brings the old behaviour back for measurements. */
int compute_streams()
{
    const char *e = getenv("GKM_MULTI_STREAMS");
    return (e && !strcmp(e, "two")) ? 2 : 1;
}

struct Call {
    int G = 0, n = 0, chunks = 1, symmetric = 0, nsk = 1;
    int64_t ld = 0, pe = 0;
    bool use_rccl = false;
    gkmhip_ctx **ctxs = nullptr;
    double **K = nullptr;
    std::vector<int> devs;
    std::vector<int64_t> row_offset; /* where matrix row a starts in the gathered slabs (gkm_shard.h) */
    std::vector<std::string> err;
    std::atomic<int> failed{0};
    std::atomic<int> stuck{0}; /* a collective was enqueued by some ranks only: do not wait for it */
    HostBarrier *bar = nullptr;
    /* >= 0: gkmhip_gram_rank_alone -- only this rank runs, ALONE on its device: its own slab goes into its gathered
     * buffer by a device copy, the peers' slabs are what an earlier gkmhip_gram_allgather of the same shape left there */
    int alone = -1;
    double wall_ms = 0; /* (alone) the rank's step on the host clock: first enqueue -> everything complete */
};

/* what the most recent call measured, per rank (gkmhip_allgather_stats) */
struct RankStats {
    double kernel_ms = 0, transfer_ms = 0, assemble_ms = 0, comparisons = 0;
    /* per chunk: its launch group's start and end, its transfer's start and end, all counted from the start of the
     * rank's first launch group (HIP event timestamps; gkmhip_allgather_chunk_times) */
    std::vector<double> chunk_times;
};
std::vector<RankStats> g_stats;
int g_stats_chunks = 0;
long long g_bytes_per_rank = 0; /* received from the peers per matrix by every rank in the most recent call */

#define MCHK(expr)                                                                              \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess && !fail) {                                                        \
            fail = true;                                                                        \
            C.err[(size_t)g] = std::string(#expr) + ": " + hipGetErrorString(e_);               \
        }                                                                                       \
    } while (0)

void rank_thread(Call &C, int g)
{
    bool fail = false;
    const int G = C.G, n = C.n, chunks = C.chunks, dev = C.devs[(size_t)g];
    /* PACKED slabs: row a of a chunk is a + 1 doubles, rows back to back, every chunk padded to the largest one
     * over all ranks (gkm_shard.h) -- n^2 / (2G) doubles per rank instead of n^2 / G */
    const size_t slab_elems = (size_t)C.pe;
    RankCache &R = g_cache[g];
    const std::vector<std::vector<int>> parts = gkmshard::chunked_layout(n, G, g, chunks);

    /* ---- phase 0: buffers, streams, events (kept from the previous call of the same shape) ---- */
    MCHK(hipSetDevice(dev));
    if (!fail && !(R.dev == dev && R.n == n && R.G == G && R.chunks == chunks && R.pe == C.pe && R.nsk == C.nsk)) {
        cache_release(R);
        R.dev = dev; R.n = n; R.G = G; R.chunks = chunks; R.pe = C.pe; R.nsk = C.nsk;
        auto dmalloc = [&](void **p, size_t bytes) { g_allocs++; return hipMalloc(p, bytes); };
        MCHK(dmalloc((void **)&R.slab, (size_t)chunks * slab_elems * sizeof(double)));
        if (!fail) MCHK(dmalloc((void **)&R.gathered, (size_t)chunks * (size_t)G * slab_elems * sizeof(double)));
        if (!fail) MCHK(dmalloc((void **)&R.d_slot, (size_t)n * sizeof(int64_t)));
        if (!fail) MCHK(dmalloc((void **)&R.sq, (size_t)n * sizeof(double)));
        /* The transfer stream must not share a hardware queue with the compute stream (streams on one queue execute in
         * order: the all-gather of chunk c would wait for the kernel of chunk c+1), so it is probed against it. */
        void *busy[3] = {nullptr, nullptr, nullptr};
        for (int i = 0; i < R.nsk && !fail; i++) {
            if (i == 0) MCHK(hipStreamCreateWithFlags(&R.sk[0], hipStreamNonBlocking));
            else R.sk[i] = (hipStream_t)gkmhip_create_stream_beside(busy, i, nullptr);
            if (!fail && !R.sk[i]) { fail = true; C.err[(size_t)g] = "cannot create the rank's streams"; }
            busy[i] = R.sk[i];
        }
        if (!fail) {
            R.sc = (hipStream_t)gkmhip_create_stream_beside(busy, R.nsk, nullptr);
            if (!R.sc) { fail = true; C.err[(size_t)g] = "cannot create the rank's streams"; }
        }
        R.ready.assign((size_t)chunks, nullptr);
        for (auto *v : {&R.k0, &R.k1, &R.a0, &R.a1}) v->assign((size_t)chunks, nullptr);
        for (int c = 0; c < chunks && !fail; c++) {
            MCHK(hipEventCreateWithFlags(&R.ready[(size_t)c], hipEventDisableTiming));
            for (auto *v : {&R.k0, &R.k1, &R.a0, &R.a1})
                if (!fail) MCHK(hipEventCreate(&(*v)[(size_t)c]));
        }
        if (!fail) MCHK(hipEventCreate(&R.n0));
        if (!fail) MCHK(hipEventCreate(&R.n1));
        if (!fail) MCHK(hipMemcpy(R.d_slot, C.row_offset.data(), (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice));
        if (fail) { /* half-built: nothing of it may be taken for a cache hit later */
            const std::string keep = C.err[(size_t)g];
            cache_release(R);
            C.err[(size_t)g] = keep;
        }
    }
    if (fail) C.failed = 1;
    C.bar->wait();
    hipStream_t *sk = R.sk, sc = R.sc;
    RankStats st_out;
    timespec ts_step0;
    clock_gettime(CLOCK_MONOTONIC, &ts_step0);

    /* ---- phase 1: per chunk, the Gram kernel of this rank's rows, then the all-gather of the slab ---- */
    for (int c = 0; c < chunks; c++) {
        const bool go = !C.failed.load();
        double *my_slab = go ? R.slab + (size_t)c * slab_elems : nullptr;
        if (go) {
            hipStream_t st = sk[c % R.nsk];
            const std::vector<int> &rows = parts[(size_t)c];
            /* The transfer of chunk c - 1 becomes runnable (stream sc) when this launch group does, and it must be FIRST on
             * the device: a collective's workgroups of 256-512 threads never start beside a running Gram kernel, which holds
             * 7 of 8 wave slots and 504 of 512 VGPRs of every SIMD and replaces each wave that retires at once (round 5, one
             * GPU, tools/collective_beside_probe.py: a 90-MB copy by 64 x 256 threads enqueued mid-kernel takes 5.3 ms
             * instead of 0.10 and ends when the kernel does; one that waits for the previous kernel's event takes 0.19).
             * What gives it the head start is this group's own table upload and row-plane kernel, ~0.1 ms in front of
             * the Gram kernel: whoever moves them out of the way must put a pause here (gkmhip_pause_stream). */
            MCHK(hipEventRecord(R.k0[(size_t)c], st));
            if (!rows.empty()) {
                const std::vector<int64_t> roff = gkmshard::packed_row_offsets(rows);
                int rc = gkmhip_set_scratch_slot(C.ctxs[g], c & 1);
                if (!rc) rc = gkmhip_gram_rows_packed(C.ctxs[g], rows.data(), (int)rows.size(), my_slab, roff.data(), st);
                if (rc && !fail) { fail = true; C.err[(size_t)g] = gkmhip_last_error(); }
                if (!rc) st_out.comparisons += gkmhip_last_comparisons(C.ctxs[g]);
            }
            MCHK(hipEventRecord(R.k1[(size_t)c], st));
            MCHK(hipEventRecord(R.ready[(size_t)c], st));
            if (fail) C.failed = 1;
        }
        if (C.use_rccl) {
            /* every rank must enter the collective or none: a rank whose launch failed would leave the others
             * waiting in the all-gather for ever, so the ranks agree on the host first */
            if (!C.bar->wait(C.failed.load() != 0)) {
                MCHK(hipStreamWaitEvent(sc, R.ready[(size_t)c], 0));
                MCHK(hipEventRecord(R.a0[(size_t)c], sc));
                if (!fail) {
                    const ncclResult_t r = g_rccl.AllGather(my_slab, R.gathered + (size_t)c * (size_t)G * slab_elems,
                                                            slab_elems, GKM_NCCL_FLOAT64, g_rccl.comms[(size_t)g], sc);
                    if (r != 0) { fail = true; C.err[(size_t)g] = std::string("ncclAllGather: ") + g_rccl.GetErrorString(r); }
                }
                MCHK(hipEventRecord(R.a1[(size_t)c], sc));
                if (fail) C.failed = 1;
                /* ... and agree again that EVERY rank has enqueued it: if one could not, the others' collective will
                 * never complete, and nobody may wait for stream sc */
                if (C.bar->wait(fail)) C.stuck = 1;
            }
        } else {
            /* every rank has RECORDED ready[.][c]: an unrecorded event would not be waited for */
            if (!C.bar->wait(C.failed.load() != 0)) {
                /* first every peer's slab, THEN the start event: transfer_ms brackets the copies only, as the RCCL
                 * branch's brackets the collective only (not the wait for the slowest peer's kernel) */
                const int r_lo = C.alone >= 0 ? g : 0, r_hi = C.alone >= 0 ? g + 1 : G;
                for (int r = r_lo; r < r_hi && !fail; r++) MCHK(hipStreamWaitEvent(sc, g_cache[r].ready[(size_t)c], 0));
                MCHK(hipEventRecord(R.a0[(size_t)c], sc));
                for (int r = r_lo; r < r_hi && !fail; r++) {
                    MCHK(hipMemcpyPeerAsync(R.gathered + ((size_t)c * (size_t)G + (size_t)r) * slab_elems, dev,
                                            g_cache[r].slab + (size_t)c * slab_elems, C.devs[(size_t)r],
                                            slab_elems * sizeof(double), sc));
                }
                MCHK(hipEventRecord(R.a1[(size_t)c], sc));
                if (fail) C.failed = 1;
            }
        }
    }
    (void)gkmhip_set_scratch_slot(C.ctxs[g], 0);

    /* ---- phase 2: un-permute + normalise this device's copy of the whole matrix ---- */
    bool assembled = false;
    if (!C.failed.load()) {
        MCHK(hipEventRecord(R.n0, sc));
        const int rc = gkmhip_assemble_normalize(C.ctxs[g], R.gathered, 1, R.d_slot, C.K[g], C.ld, R.sq, C.symmetric, sc);
        if (rc && !fail) { fail = true; C.err[(size_t)g] = gkmhip_last_error(); }
        MCHK(hipEventRecord(R.n1, sc));
        assembled = !fail;
    }
    for (int i = 0; i < 2; i++)
        if (sk[i]) (void)hipStreamSynchronize(sk[i]);
    if (sc && !C.stuck.load()) {
        hipError_t e = hipStreamSynchronize(sc);
        if (e != hipSuccess && !fail) { fail = true; C.err[(size_t)g] = std::string("hipStreamSynchronize: ") + hipGetErrorString(e); }
    }
    {
        timespec ts_step1;
        clock_gettime(CLOCK_MONOTONIC, &ts_step1);
        if (C.alone >= 0) C.wall_ms = (ts_step1.tv_sec - ts_step0.tv_sec) * 1e3 + (ts_step1.tv_nsec - ts_step0.tv_nsec) * 1e-6;
    }
    if (assembled && !fail) {
        float ms = 0.f;
        for (int c = 0; c < chunks; c++) {
            if (hipEventElapsedTime(&ms, R.k0[(size_t)c], R.k1[(size_t)c]) == hipSuccess) st_out.kernel_ms += ms;
            if (hipEventElapsedTime(&ms, R.a0[(size_t)c], R.a1[(size_t)c]) == hipSuccess) st_out.transfer_ms += ms;
            for (hipEvent_t ev : {R.k0[(size_t)c], R.k1[(size_t)c], R.a0[(size_t)c], R.a1[(size_t)c]})
                st_out.chunk_times.push_back(hipEventElapsedTime(&ms, R.k0[0], ev) == hipSuccess ? (double)ms : -1.0);
        }
        if (hipEventElapsedTime(&ms, R.n0, R.n1) == hipSuccess) st_out.assemble_ms = ms;
        g_stats[(size_t)g] = st_out;
    }
    if (fail) C.failed = 1;
    C.bar->wait(); /* nobody reads this rank's slab any more */
}

} /* namespace */

extern "C" const char *gkmhip_last_transport(void) { return g_transport.c_str(); }

extern "C" void gkmhip_release_comms(void)
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    int caller_device = -1;
    (void)hipGetDevice(&caller_device);
    if (g_rccl.AllGather) rccl_drop_comms();
    for (RankCache &R : g_cache) cache_release(R);
    if (caller_device >= 0) (void)hipSetDevice(caller_device);
}

extern "C" long gkmhip_allgather_alloc_count(void) { return g_allocs.load(); }
extern "C" long long gkmhip_allgather_bytes_per_rank(void) { return g_bytes_per_rank; }

/* out[0] = ranks, out[1] = chunks, out[2] = transport (0 none, 1 p2p, 2 rccl), then per rank
 * {kernel ms (sum over chunks), transfer ms (sum over chunks), assemble ms, l-mer comparisons} */
extern "C" int gkmhip_allgather_stats(double *out, int cap)
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    const int need = 3 + 4 * (int)g_stats.size();
    if (!out || cap < need || g_stats.empty()) return 0;
    out[0] = (double)g_stats.size();
    out[1] = (double)g_stats_chunks;
    out[2] = g_transport == "rccl" ? 2.0 : g_transport == "p2p" ? 1.0 : 0.0;
    for (size_t g = 0; g < g_stats.size(); g++) {
        out[3 + 4 * g] = g_stats[g].kernel_ms;
        out[4 + 4 * g] = g_stats[g].transfer_ms;
        out[5 + 4 * g] = g_stats[g].assemble_ms;
        out[6 + 4 * g] = g_stats[g].comparisons;
    }
    return need;
}

/* out[4 c .. 4 c + 3] = chunk c of rank `rank` in the most recent call: launch group start / end, transfer start / end,
 * ms from the start of the rank's first launch group.  Returns the number of doubles written. */
extern "C" int gkmhip_allgather_chunk_times(int rank, double *out, int cap)
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (!out || rank < 0 || (size_t)rank >= g_stats.size()) return 0;
    const std::vector<double> &t = g_stats[(size_t)rank].chunk_times;
    if ((size_t)cap < t.size()) return 0;
    for (size_t i = 0; i < t.size(); i++) out[i] = t[i];
    return (int)t.size();
}

/* ONE rank of a `ranks`-way gkmhip_gram_allgather, alone on its device: what a rank's step costs without the transfer.
 * No node with several GPUs has been available to any round, and a rehearsal with all ranks on one device makes every
 * rank's kernel `ranks` times too long; this runs rank `rank`'s own chunks (same layout, streams, scratch slots, packed
 * slabs, events as rank_thread above -- it IS rank_thread), puts its slab into its gathered buffer with a device copy
 * and assembles + normalises the whole matrix from that buffer, whose other ranks' slabs must be there from an earlier
 * gkmhip_gram_allgather call with the same contexts' shape (n, ranks, chunks) -- the call fails if they are not.
 * out[0..5] = wall ms (host clock, first enqueue -> all streams complete), kernel ms (sum over the chunks' launches,
 * table uploads / row planes / untile included), copy-in ms, assemble ms, comparisons, chunks. */
extern "C" int gkmhip_gram_rank_alone(gkmhip_ctx *ctx, int rank, int ranks, int chunks, double *K, int64_t ld, int symmetric,
                                      double *out6)
{
    if (!ctx || !K || ranks < 1 || ranks > 64 || rank < 0 || rank >= ranks) return fail_with("gkmhip_gram_rank_alone: bad arguments", 2);
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    Call C;
    C.G = ranks;
    C.n = gkmhip_n_sequences(ctx);
    if (C.n <= 0 || ld < C.n) return fail_with("gkmhip_gram_rank_alone: no sequences uploaded or leading dimension too small", 2);
    C.chunks = chunks > 0 ? chunks : (ranks == 1 ? 1 : gkmshard::auto_chunks(C.n, ranks));
    C.pe = gkmshard::packed_chunk_elems(C.n, ranks, C.chunks);
    const RankCache &R = g_cache[rank];
    C.nsk = compute_streams();
    if (!(R.dev == gkmhip_device_of(ctx) && R.n == C.n && R.G == ranks && R.chunks == C.chunks && R.pe == C.pe && R.nsk == C.nsk))
        return fail_with("gkmhip_gram_rank_alone: no gathered slabs of this shape (run gkmhip_gram_allgather with the same "
                         "number of contexts and chunks first)", 2);
    std::vector<gkmhip_ctx *> ctxs((size_t)ranks, nullptr);
    std::vector<double *> Ks((size_t)ranks, nullptr);
    ctxs[(size_t)rank] = ctx;
    Ks[(size_t)rank] = K;
    C.ctxs = ctxs.data();
    C.K = Ks.data();
    C.ld = ld;
    C.symmetric = symmetric;
    C.devs.assign((size_t)ranks, gkmhip_device_of(ctx));
    C.row_offset = gkmshard::packed_gather_offsets(C.n, ranks, C.chunks);
    C.err.assign((size_t)ranks, std::string());
    C.alone = rank;
    HostBarrier bar(1);
    C.bar = &bar;
    if (g_stats.size() != (size_t)ranks) g_stats.assign((size_t)ranks, RankStats());
    int caller_device = -1;
    (void)hipGetDevice(&caller_device);
    rank_thread(C, rank);
    if (caller_device >= 0) (void)hipSetDevice(caller_device);
    if (C.failed.load()) return fail_with("gkmhip_gram_rank_alone failed: " + C.err[(size_t)rank], 7);
    if (out6) {
        const RankStats &st = g_stats[(size_t)rank];
        out6[0] = C.wall_ms; out6[1] = st.kernel_ms; out6[2] = st.transfer_ms; out6[3] = st.assemble_ms;
        out6[4] = st.comparisons; out6[5] = (double)C.chunks;
    }
    return 0;
}

extern "C" int gkmhip_gram_allgather(gkmhip_ctx **ctxs, int nctx, double **K, int64_t ld, int symmetric, int chunks)
{
    if (!ctxs || !K || nctx < 1 || nctx > 64) { return fail_with("gkmhip_gram_allgather: bad arguments", 2); }
    std::lock_guard<std::mutex> lock(g_rccl_mutex); /* one multi-GPU call at a time per process */
    Call C;
    C.G = nctx;
    C.ctxs = ctxs;
    C.K = K;
    C.ld = ld;
    C.symmetric = symmetric;
    C.nsk = compute_streams();
    C.n = gkmhip_n_sequences(ctxs[0]);
    bool distinct = true;
    for (int g = 0; g < nctx; g++) {
        if (!ctxs[g] || !K[g] || gkmhip_n_sequences(ctxs[g]) != C.n) {
            return fail_with("gkmhip_gram_allgather: every context needs the same sequences uploaded and an output matrix", 2);
        }
        C.devs.push_back(gkmhip_device_of(ctxs[g]));
        for (int h = 0; h < g; h++) distinct = distinct && C.devs[(size_t)h] != C.devs[(size_t)g];
    }
    if (C.n <= 0 || ld < C.n) { return fail_with("gkmhip_gram_allgather: no sequences uploaded or leading dimension too small", 2); }
    /* (one context: one chunk unless the caller asks -- the tests do, to run the chunked sequence of launches, events and
     * collectives through a one-rank RCCL communicator on a box with one GPU) */
    C.chunks = chunks > 0 ? chunks : (nctx == 1 ? 1 : gkmshard::auto_chunks(C.n, nctx));
    C.pe = gkmshard::packed_chunk_elems(C.n, nctx, C.chunks);
    C.row_offset = gkmshard::packed_gather_offsets(C.n, nctx, C.chunks);
    g_bytes_per_rank = (long long)C.chunks * (nctx - 1) * (long long)C.pe * 8;

    const char *force = getenv("GKM_ALLGATHER");
    const bool want_rccl = force ? !strcmp(force, "rccl") : (nctx > 1 && distinct);
    if (force && strcmp(force, "rccl") && strcmp(force, "p2p")) { return fail_with("GKM_ALLGATHER must be rccl or p2p", 2); }
    if (want_rccl) {
        std::string why;
        if (!distinct) { return fail_with("GKM_ALLGATHER=rccl needs the contexts on distinct devices", 2); }
        if (rccl_comms_for(C.devs, why)) C.use_rccl = true;
        else if (force) { return fail_with(why, 6); }
        else fprintf(stderr, "gkmhip_gram_allgather: %s -- falling back to peer copies\n", why.c_str());
    }
    g_transport = C.use_rccl ? "rccl" : (nctx > 1 ? "p2p" : "none");

    g_stats.assign((size_t)nctx, RankStats());
    g_stats_chunks = C.chunks;
    int caller_device = -1;
    (void)hipGetDevice(&caller_device);
    /* buffers of ranks this call does not have (an earlier call used more contexts) would only hold memory */
    for (int g = nctx; g < 64; g++) cache_release(g_cache[g]);
    C.err.assign((size_t)nctx, std::string());
    HostBarrier bar(nctx);
    C.bar = &bar;
    std::vector<std::thread> th;
    for (int g = 1; g < nctx; g++) th.emplace_back(rank_thread, std::ref(C), g);
    rank_thread(C, 0);
    for (auto &t : th) t.join();
    if (C.stuck.load()) {
        /* a collective that some rank never joined: abort the communicators (which ends the waiting kernels) and
         * drop the buffers they were writing; the next call starts from scratch */
        rccl_drop_comms(true);
        for (int g = 0; g < nctx; g++) cache_release(g_cache[g]);
    }
    if (caller_device >= 0) (void)hipSetDevice(caller_device);
    if (C.failed.load()) {
        g_stats.clear();
        std::string msg = "gkmhip_gram_allgather failed";
        for (int g = 0; g < nctx; g++)
            if (!C.err[(size_t)g].empty()) msg += "; rank " + std::to_string(g) + ": " + C.err[(size_t)g];
        return fail_with(msg, 7);
    }
    return 0;
}
