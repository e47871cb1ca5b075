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
 * The chunks of a rank alternate between two compute streams; the transfer of chunk c runs on a
 * third stream and overlaps the kernel of chunk c+1.  Integer profiles are placement-independent,
 * so the assembled matrix is bit-identical to the single-GPU one for any number of devices.
 */
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
typedef ncclResult_t (*fn_AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t);
typedef const char *(*fn_GetErrorString)(ncclResult_t);

namespace {

struct Rccl {
    void *handle = nullptr;
    fn_CommInitAll CommInitAll = nullptr;
    fn_CommDestroy CommDestroy = nullptr;
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
    g_rccl.AllGather = (fn_AllGather)dlsym(g_rccl.handle, "ncclAllGather");
    g_rccl.GetErrorString = (fn_GetErrorString)dlsym(g_rccl.handle, "ncclGetErrorString");
    if (!g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.GetErrorString) {
        g_rccl.AllGather = nullptr;
        g_rccl.why = "RCCL library lacks ncclCommInitAll / ncclAllGather";
        return false;
    }
    return true;
}

void rccl_drop_comms()
{
    for (ncclComm_t c : g_rccl.comms)
        if (c) (void)g_rccl.CommDestroy(c);
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

struct Call {
    int G = 0, n = 0, chunks = 1, pc = 0, symmetric = 0;
    int64_t ld = 0;
    bool use_rccl = false;
    gkmhip_ctx **ctxs = nullptr;
    double **K = nullptr;
    std::vector<int> devs;
    std::vector<double *> slab, gathered;               /* per rank, on its device */
    std::vector<std::vector<hipEvent_t>> ready;         /* [rank][chunk]: that slab is complete */
    std::vector<int64_t> slot_of_row;
    std::vector<std::string> err;
    std::atomic<int> failed{0};
    HostBarrier *bar = nullptr;
};

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
    const int G = C.G, n = C.n, chunks = C.chunks, pc = C.pc, dev = C.devs[(size_t)g];
    const size_t slab_elems = (size_t)pc * (size_t)n;
    hipStream_t sk[2] = {nullptr, nullptr}, sc = nullptr;
    int64_t *d_slot = nullptr;
    double *sq = nullptr;
    const std::vector<std::vector<int>> parts = gkmshard::chunked_layout(n, G, g, chunks);

    /* ---- phase 0: buffers, streams, events ---- */
    MCHK(hipSetDevice(dev));
    if (!fail) MCHK(hipMalloc((void **)&C.slab[(size_t)g], (size_t)chunks * slab_elems * sizeof(double)));
    if (!fail) MCHK(hipMalloc((void **)&C.gathered[(size_t)g], (size_t)chunks * (size_t)G * slab_elems * sizeof(double)));
    if (!fail) MCHK(hipMalloc((void **)&d_slot, (size_t)n * sizeof(int64_t)));
    if (!fail) MCHK(hipMalloc((void **)&sq, (size_t)n * sizeof(double)));
    for (int i = 0; i < 2 && !fail; i++) MCHK(hipStreamCreateWithFlags(&sk[i], hipStreamNonBlocking));
    if (!fail) MCHK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    for (int c = 0; c < chunks && !fail; c++) MCHK(hipEventCreateWithFlags(&C.ready[(size_t)g][(size_t)c], hipEventDisableTiming));
    if (!fail) MCHK(hipMemcpy(d_slot, C.slot_of_row.data(), (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice));
    if (fail) C.failed = 1;
    C.bar->wait();

    /* ---- phase 1: per chunk, the Gram kernel of this rank's rows, then the all-gather of the slab ---- */
    for (int c = 0; c < chunks; c++) {
        const bool go = !C.failed.load();
        double *my_slab = go ? C.slab[(size_t)g] + (size_t)c * slab_elems : nullptr;
        if (go) {
            hipStream_t st = sk[c & 1];
            const std::vector<int> &rows = parts[(size_t)c];
            if (!rows.empty()) {
                int rc = gkmhip_set_scratch_slot(C.ctxs[g], c & 1);
                if (!rc) rc = gkmhip_gram_rows(C.ctxs[g], rows.data(), (int)rows.size(), 1, my_slab, n, nullptr, 0, st);
                if (rc && !fail) { fail = true; C.err[(size_t)g] = gkmhip_last_error(); }
            }
            MCHK(hipEventRecord(C.ready[(size_t)g][(size_t)c], st));
            if (fail) C.failed = 1;
        }
        if (C.use_rccl) {
            /* every rank must enter the collective or none: a rank whose launch failed would leave the others
             * waiting in the all-gather for ever, so the ranks agree on the host first */
            if (!C.bar->wait(C.failed.load() != 0)) {
                MCHK(hipStreamWaitEvent(sc, C.ready[(size_t)g][(size_t)c], 0));
                const ncclResult_t r = g_rccl.AllGather(my_slab, C.gathered[(size_t)g] + (size_t)c * (size_t)G * slab_elems,
                                                        slab_elems, GKM_NCCL_FLOAT64, g_rccl.comms[(size_t)g], sc);
                if (r != 0 && !fail) { fail = true; C.err[(size_t)g] = std::string("ncclAllGather: ") + g_rccl.GetErrorString(r); }
                if (fail) C.failed = 1;
            }
        } else {
            /* every rank has RECORDED ready[.][c]: an unrecorded event would not be waited for */
            if (!C.bar->wait(C.failed.load() != 0)) {
                for (int r = 0; r < G && !fail; r++) {
                    MCHK(hipStreamWaitEvent(sc, C.ready[(size_t)r][(size_t)c], 0));
                    MCHK(hipMemcpyPeerAsync(C.gathered[(size_t)g] + ((size_t)c * (size_t)G + (size_t)r) * slab_elems, dev,
                                            C.slab[(size_t)r] + (size_t)c * slab_elems, C.devs[(size_t)r],
                                            slab_elems * sizeof(double), sc));
                }
                if (fail) C.failed = 1;
            }
        }
    }
    (void)gkmhip_set_scratch_slot(C.ctxs[g], 0);

    /* ---- phase 2: un-permute + normalise this device's copy of the whole matrix ---- */
    if (!C.failed.load()) {
        const int rc = gkmhip_assemble_normalize(C.ctxs[g], C.gathered[(size_t)g], n, d_slot, C.K[g], C.ld, sq, C.symmetric, sc);
        if (rc && !fail) { fail = true; C.err[(size_t)g] = gkmhip_last_error(); }
    }
    for (int i = 0; i < 2; i++)
        if (sk[i]) (void)hipStreamSynchronize(sk[i]);
    if (sc) {
        hipError_t e = hipStreamSynchronize(sc);
        if (e != hipSuccess && !fail) { fail = true; C.err[(size_t)g] = std::string("hipStreamSynchronize: ") + hipGetErrorString(e); }
    }
    if (fail) C.failed = 1;
    C.bar->wait(); /* nobody reads this rank's slab any more */

    if (d_slot) (void)hipFree(d_slot);
    if (sq) (void)hipFree(sq);
    if (C.slab[(size_t)g]) (void)hipFree(C.slab[(size_t)g]);
    if (C.gathered[(size_t)g]) (void)hipFree(C.gathered[(size_t)g]);
    for (int c = 0; c < chunks; c++)
        if (C.ready[(size_t)g][(size_t)c]) (void)hipEventDestroy(C.ready[(size_t)g][(size_t)c]);
    for (int i = 0; i < 2; i++)
        if (sk[i]) (void)hipStreamDestroy(sk[i]);
    if (sc) (void)hipStreamDestroy(sc);
}

} /* namespace */

extern "C" const char *gkmhip_last_transport(void) { return g_transport.c_str(); }

extern "C" void gkmhip_release_comms(void)
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.AllGather) rccl_drop_comms();
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
    C.chunks = nctx == 1 ? 1 : (chunks > 0 ? chunks : 4);
    C.pc = gkmshard::chunk_rows(C.n, nctx, C.chunks);
    C.slot_of_row = gkmshard::chunked_gather_index(C.n, nctx, C.chunks);

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

    C.slab.assign((size_t)nctx, nullptr);
    C.gathered.assign((size_t)nctx, nullptr);
    C.ready.assign((size_t)nctx, std::vector<hipEvent_t>((size_t)C.chunks, nullptr));
    C.err.assign((size_t)nctx, std::string());
    HostBarrier bar(nctx);
    C.bar = &bar;
    int caller_device = -1;
    (void)hipGetDevice(&caller_device);
    std::vector<std::thread> th;
    for (int g = 1; g < nctx; g++) th.emplace_back(rank_thread, std::ref(C), g);
    rank_thread(C, 0);
    for (auto &t : th) t.join();
    if (caller_device >= 0) (void)hipSetDevice(caller_device);
    if (C.failed.load()) {
        std::string msg = "gkmhip_gram_allgather failed";
        for (int g = 0; g < nctx; g++)
            if (!C.err[(size_t)g].empty()) msg += "; rank " + std::to_string(g) + ": " + C.err[(size_t)g];
        return fail_with(msg, 7);
    }
    return 0;
}
