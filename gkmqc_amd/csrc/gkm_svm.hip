/*
 * gkm_svm.hip -- C-SVC on a precomputed kernel resident in HBM (include/gkm_svm.h, SURVEY.md
 * §8(f4)).  One workgroup per problem (cross-validation fold); all folds of a CV run
 * concurrently on different CUs.  The iteration is LIBSVM's SMO with second-order working-set
 * selection and no shrinking (Fan, Chen, Lin 2005; with shrinking: k_smo_general below), restated so that every floating-point
 * operation and every tie break matches the sequential solver: same alpha, same rho.
 *
 * Per iteration (l = training samples, each thread owns up to R of them, strided):
 *   A  i = argmax over I_up of -y G          (ties: the LARGER index, LIBSVM scans with >=)
 *   B  gather Q_i = (float)(y_i y_k K_ik), j = argmin over I_low of -(grad_diff^2)/quad (ties: larger
 *      index, LIBSVM scans with <=), Gmax2; stop when Gmax + Gmax2 < eps
 *   C  every thread updates alpha_i, alpha_j with LIBSVM's clipping (same scalars, same result)
 *   D  gather Q_j, G_k += Q_ik dalpha_i + Q_jk dalpha_j
 * alpha and G live in registers; the kernel matrix is only read (two row gathers per iteration).
 * Measured (MI355X, 5 folds of 8000 samples from the 10 000 x 10 000 headline matrix, 10.4 k
 * iterations each): 0.10 s for the whole cross-validation, ~22 k cycles per iteration, bound by
 * the instruction issue of the one CU a fold runs on (scikit-learn, 5 processes: 0.95 s).
 * -DSVM_PROF prints the per-phase cycle counts.
 */
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gkm_svm.h"

static thread_local std::string g_svm_err;
extern "C" const char *gkmsvm_last_error(void) { return g_svm_err.c_str(); }
static int svm_fail(const char *what, hipError_t e)
{
    g_svm_err = std::string(what) + ": " + hipGetErrorString(e);
    return 100 + (int)e;
}
#define SVMCHK(expr)                                   \
    do {                                               \
        hipError_t e_ = (expr);                        \
        if (e_ != hipSuccess) return svm_fail(#expr, e_); \
    } while (0)

constexpr int SVM_MAX_R = 16;      /* samples per thread at the largest size */
constexpr int SVM_MAX_THREADS = 1024;
constexpr int SVM_MAX_L = SVM_MAX_THREADS * SVM_MAX_R;
constexpr double SVM_TAU = 1e-12;

/* Device scratch of the solver calls, kept for the life of the process: a pipeline solves the folds of one subset on a
 * second stream while the Gram kernel of the next subset runs (gkmqc_amd/gkmsvm.py init_many), and hipMalloc / hipFree
 * wait for the whole device -- i.e. for that Gram kernel.  A pool, because two workers may solve on one device at
 * the same time.  gkmsvm_release_cache() frees what is not in use. */
struct SvmScratch {
    int device = -1;
    char *p = nullptr;
    size_t cap = 0;
    bool in_use = false;
};
static std::mutex g_scratch_mutex;
static std::vector<SvmScratch *> g_scratch;

static SvmScratch *scratch_acquire(int device, size_t bytes)
{
    SvmScratch *b = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        for (SvmScratch *c : g_scratch)
            if (!c->in_use && c->device == device && (!b || (c->cap >= bytes && (b->cap < bytes || c->cap < b->cap)))) b = c;
        if (!b) {
            b = new SvmScratch();
            b->device = device;
            g_scratch.push_back(b);
        }
        b->in_use = true;
    }
    if (b->cap < bytes) { /* (this thread owns b now) */
        if (b->p) (void)hipFree(b->p);
        b->p = nullptr;
        b->cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipMalloc((void **)&b->p, want) != hipSuccess) {
            std::lock_guard<std::mutex> lock(g_scratch_mutex);
            b->in_use = false;
            return nullptr;
        }
        b->cap = want;
    }
    return b;
}
static void scratch_release(SvmScratch *b)
{
    if (!b) return;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    b->in_use = false;
}
extern "C" void gkmsvm_release_cache(void)
{
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    int dev = -1;
    (void)hipGetDevice(&dev);
    for (SvmScratch *b : g_scratch)
        if (!b->in_use && b->p) {
            (void)hipSetDevice(b->device);
            (void)hipFree(b->p);
            b->p = nullptr;
            b->cap = 0;
        }
    if (dev >= 0) (void)hipSetDevice(dev);
}
static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

/* A pointer READ from memory (the fields of SvmProb, GenProb, DecProb) is a generic one to the compiler, and every access through it a flat_
 * instruction -- which counts as an LDS access too, so that each wait for LDS data waits for all of them.  The arrays of
 * the iteration are therefore held as pointers into the global address space. */
#define GKM_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ GKM_GLOBAL T *as_global(T *p)
{
    return (GKM_GLOBAL T *)p;
}

struct SvmProb {
    const int *idx;
    int l, n0;
    double *alpha, *grad, *rho;
    int *iters;
};

/* A candidate of the working-set selection together with everything the other threads need to
 * know about it, so that one LDS record per wave (and one barrier) publishes the winner. */
struct Cand {
    double v;      /* selection value */
    double alpha, G, qd, q; /* of sample k: dual variable, gradient, K_kk, Q_ik (second index only) */
    int k;         /* position in the problem (-1: none) */
    int g;         /* row/column of K */
};

/* LIBSVM scans t = 0..l-1 and replaces the incumbent on `>=` (first index) or `<=` (second
 * index): among equal values the LARGER index wins. */
template <bool MINIMISE>
__device__ __forceinline__ bool better(double v, int k, double bv, int bk)
{
    if (k < 0) return false;
    if (bk < 0) return true;
    return MINIMISE ? (v < bv || (v == bv && k > bk)) : (v > bv || (v == bv && k > bk));
}

/* DPP lane exchange inside a row of 16 lanes (no LDS round trip as with ds_bpermute) */
template <int CTRL>
__device__ __forceinline__ int dpp_i(int x)
{
    /* every lane has a valid source under these controls: no 'old' value, so no register copy */
    return __builtin_amdgcn_mov_dpp(x, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double x)
{
    return __hiloint2double(dpp_i<CTRL>(__double2hiint(x)), dpp_i<CTRL>(__double2loint(x)));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;

/* v_max_f64 / v_min_f64 without the canonicalisation hipcc wraps around fmax()/fmin() (the values
 * compared here are never signalling NaNs) */
template <bool MINIMISE>
__device__ __forceinline__ double extreme2(double a, double b)
{
    double r;
    if (MINIMISE) asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    else asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

/* extreme of each row of 16 lanes, in all its lanes */
template <bool MINIMISE>
__device__ __forceinline__ double row_extreme(double v)
{
    v = extreme2<MINIMISE>(v, dpp_d<DPP_XOR1>(v));
    v = extreme2<MINIMISE>(v, dpp_d<DPP_XOR2>(v));
    v = extreme2<MINIMISE>(v, dpp_d<DPP_HALF_MIRROR>(v));
    v = extreme2<MINIMISE>(v, dpp_d<DPP_MIRROR>(v));
    return v;
}
__device__ __forceinline__ int row_imax(int v)
{
    v = max(v, dpp_i<DPP_XOR1>(v));
    v = max(v, dpp_i<DPP_XOR2>(v));
    v = max(v, dpp_i<DPP_HALF_MIRROR>(v));
    v = max(v, dpp_i<DPP_MIRROR>(v));
    return v;
}

/* extreme of the wave, in all its lanes: rows by DPP, the four rows through SGPRs */
template <bool MINIMISE>
__device__ __forceinline__ double wave_extreme(double v)
{
    v = row_extreme<MINIMISE>(v);
    double m = 0.0;
#pragma unroll
    for (int row = 0; row < 4; row++) {
        const double o = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), row * 16),
                                          __builtin_amdgcn_readlane(__double2loint(v), row * 16));
        m = row == 0 ? o : extreme2<MINIMISE>(m, o);
    }
    return m;
}
__device__ __forceinline__ int wave_imax(int v)
{
    v = row_imax(v);
    int m = 0;
#pragma unroll
    for (int row = 0; row < 4; row++) {
        const int o = __builtin_amdgcn_readlane(v, row * 16);
        m = row == 0 ? o : max(m, o);
    }
    return m;
}

/* best (v, k) of the wave, in all its lanes: the extreme value first, then the largest index among
 * the lanes that hold it (LIBSVM's tie rule).  Lanes without a candidate (k < 0) carry the neutral
 * value (-inf for a maximum, +inf for a minimum). */
template <bool MINIMISE>
__device__ __forceinline__ void wave_select(double &v, int &k)
{
    const double m = wave_extreme<MINIMISE>(v);
    k = wave_imax((k >= 0 && v == m) ? k : -1);
    v = m;
}

/* the same over one row of 16 lanes; also returns the lane (0..15 within the row) that holds the winner */
template <bool MINIMISE>
__device__ __forceinline__ int row_select(double &v, int &k)
{
    const double m = row_extreme<MINIMISE>(v);
    const int kb = row_imax((k >= 0 && v == m) ? k : -1);
    const unsigned long long holders = __ballot(k == kb && v == m);
    const unsigned row_mask = (unsigned)(holders >> ((threadIdx.x & 48))) & 0xFFFFu; /* this lane's row */
    v = m;
    k = kb;
    return row_mask ? __builtin_ctz(row_mask) : 0;
}

__device__ __forceinline__ double row_max(double v) { return row_extreme<false>(v); }
__device__ __forceinline__ double wave_max(double v) { return wave_extreme<false>(v); }

__global__ void k_diag(const double *__restrict__ K, int64_t ld, int n, double *__restrict__ diag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) diag[i] = K[(int64_t)i * ld + i];
}

/*
 * One workgroup of T threads per problem.  Thread t owns samples t, t+T, ...: their alpha, G,
 * matrix index and diagonal live in registers for the whole solve.  Two barriers per iteration:
 * after each of the two selections the winning lane of every wave publishes its candidate with
 * its payload in LDS, every thread then picks the block winner from the T/64 records and does
 * the (scalar) two-variable update redundantly, so no third exchange is needed.
 */
template <int T, int SVM_R, int TABM>
__global__ __launch_bounds__(T) void k_smo(const double *__restrict__ K, int64_t ld, const double *__restrict__ diag,
                                           const SvmProb *probs, double C, double eps, int max_iter)
{
    constexpr int NW = T / 64;
    __shared__ Cand candA[NW], candB[NW];
    __shared__ double g2s[NW];
    __shared__ double chunk[T];
    const SvmProb p = probs[blockIdx.x];
    const auto *const idx_g = as_global(p.idx);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l = p.l, n0 = p.n0;

    /* TAB: the samples' matrix indices and kernel diagonal live in (dynamic) LDS instead of
     * registers -- 48 VGPRs less at 16 samples per thread, which is what keeps alpha and G from
     * spilling to scratch (12 bytes per sample: up to 8192 samples in 96 KB of the CU's 160 KB) */
    /* TABM = 2 (more than 8192 samples: 16 per thread at 4 waves per SIMD = 128 VGPRs): alpha in LDS (128 KB), the
     * matrix indices in registers, the diagonal read from global memory where it is used (beside the gather of
     * the matrix row, whose latency it shares), row i gathered a second time for the gradient update instead of
     * kept.  With everything in registers this size spilt 712 bytes per thread to scratch: 5 folds x 16 000
     * samples of config 3 took 0.86 s, against 0.083 s for 5 x 8 000. */
    constexpr bool TAB = TABM == 1, QDG = TABM == 2, ALS = TABM == 2;
    extern __shared__ double dyn_lds[];
    double *const qd_s = dyn_lds;
    double *const al_s = dyn_lds;
    int *const gidx_s = (int *)(dyn_lds + T * SVM_R);
    int gidx[TAB || TABM == 2 ? 1 : SVM_R];
    double qd[TAB || QDG ? 1 : SVM_R], al[ALS ? 1 : SVM_R], G[SVM_R];
    auto AL = [&](int r) { return ALS ? al_s[tid + r * T] : al[ALS ? 0 : r]; };
#pragma unroll
    for (int r = 0; r < SVM_R; r++) {
        const int k = tid + r * T;
        const int g = k < l ? idx_g[k] : 0;
        const double dg = k < l ? diag[g] : 0.0;
        if (TAB) {
            gidx_s[k] = g;
            qd_s[k] = dg;
        } else if (!QDG) {
            gidx[QDG ? 0 : r] = g;
            qd[QDG ? 0 : r] = dg;
        }
        if (ALS) al_s[k] = 0.0;
        else al[ALS ? 0 : r] = 0.0;  /* LIBSVM: alpha = 0, G = p = -1; stored as S = y G (see below) */
        G[r] = k < n0 ? -1.0 : 1.0;
    }
    if (TAB || ALS) __syncthreads();
    /* The gradient is kept as S_k = y_k G_k and kernel values without LIBSVM's y_i y_k factor: every
     * use of G and Q in the solver carries the matching sign (-y G in the first selection, y G in
     * the second and in rho, Q_ik dalpha_i = y_k (float)K_ik (y_i dalpha_i) in the update), negation
     * and rounding commute, so the arithmetic is LIBSVM's bit for bit with no per-sample sign
     * selects.  G[] below holds S. */
    /* (TABM = 2: the index list itself, 64 KB in L2, read again wherever an index is needed -- kept in registers the
     * 16 indices turn into 16 64-bit row offsets that live through the whole iteration) */
    int tv = tid; /* = tid, re-made opaque every iteration so that hipcc does not hoist 16 64-bit addresses out of the loop (they spilt) */
    auto GI = [&](int r) { return TAB ? gidx_s[tid + r * T] : QDG ? idx_g[min(tv + r * T, l - 1)] : gidx[TAB || QDG ? 0 : r]; };
    auto QD = [&](int r) { return QDG ? diag[GI(r)] : TAB ? qd_s[tid + r * T] : qd[TAB || QDG ? 0 : r]; };

#ifdef SVM_PROF
    long long tp[6] = {0, 0, 0, 0, 0, 0}, t0 = clock64(), t1;
#define PROF(n) t1 = clock64(); tp[n] += t1 - t0; t0 = t1;
#else
#define PROF(n)
#endif
    int iter = 0;
    for (;; iter++) {
        if (iter >= max_iter) { iter = -iter; break; }
        if (QDG) asm volatile("" : "+v"(tv));
        PROF(5)
        /* ---- first index: argmax over I_up of -y G ---- */
        /* (branch-free: k grows with r, so inside a thread "replace on >=" is LIBSVM's tie rule) */
        double ns = INFINITY; /* min S over I_up = -Gmax */
        int bk = -1;
#pragma unroll
        for (int r = 0; r < SVM_R; r++) {
            const int k = tid + r * T;
            const bool pos = k < n0; /* y = +1 */
            const double a_r = AL(r);
            const bool below_C = a_r < C, above_0 = a_r > 0.0; /* (bitwise: no branches) */
            const bool in_up = (k < l) & ((pos & below_C) | (!pos & above_0));
            const bool take = in_up & (G[r] <= ns);
            ns = take ? G[r] : ns;
            bk = take ? k : bk;
            if (ALS && (r & 3) == 3) __builtin_amdgcn_sched_barrier(0); /* (keeps hipcc from fetching all 16 alphas at once) */
        }
        double bv = -ns;
        PROF(0)
        wave_select<false>(bv, bk);
        if (bk < 0) {
            if (lane == 0) candA[wave].k = -1;
        } else { /* k = tid + r T with T a multiple of 64: the owner is lane k % 64 and r = k / T for the whole wave */
            const int rr = __builtin_amdgcn_readfirstlane(bk) / T; /* wave-uniform: one of the blocks below runs */
            const bool owner = (bk & 63) == lane;
#pragma unroll
            for (int r = 0; r < SVM_R; r++)
                if (r == rr && owner) candA[wave] = {bv, AL(r), (bk < n0 ? G[r] : -G[r]), TAB || QDG ? 0.0 : qd[TAB || QDG ? 0 : r], 0.0, bk, TAB ? 0 : QDG ? GI(r) : gidx[TAB || QDG ? 0 : r]};
        }
        __syncthreads();
        /* the NW wave winners: one per lane of a row, DPP selection, then one broadcast read */
        int wk = candA[lane & (NW - 1)].k;
        double wv = wk >= 0 ? candA[lane & (NW - 1)].v : -INFINITY;
        const int ww = row_select<false>(wv, wk) & (NW - 1);
        Cand ci = candA[ww];
        ci.k = wk; /* -1 when no wave has a candidate */
        PROF(1)
        const int i = ci.k;
        if (i < 0) break;
        const double Gmax = ci.v;
        const double yi = i < n0 ? 1.0 : -1.0;
        const double *Ki = K + (int64_t)(TAB ? gidx_s[i] : ci.g) * ld;
        const double QDi = QDG ? diag[ci.g] : TAB ? qd_s[i] : ci.qd;

        /* ---- second index: argmin over I_low of -(grad_diff^2)/quad, and Gmax2 ---- */
        /* (float)K_ik: LIBSVM's Qfloat without its sign y_i y_k; TABM = 2 does not keep the 16 of them but reads
         * row i again for the gradient update (a third gather per iteration is cheaper than the spills) */
        constexpr bool KEEP = !QDG;
        float kfi[KEEP ? SVM_R : 1], mkf = 0.0f;
        double mv = INFINITY, g2max = -INFINITY;
        int mk = -1;
        /* (8 samples at a time: at 16 per thread the 16 doubles of the row and, TABM = 2, of the diagonal in flight
         * together were what spilt) */
        constexpr int BATCH = QDG ? 8 : SVM_R; /* (one batch = the loops as they were for the other shapes) */
        if constexpr (!QDG) { /* the loops as they have always been for these shapes */
            double kik[SVM_R];
#pragma unroll
            for (int r = 0; r < SVM_R; r++) kik[r] = Ki[GI(r)]; /* lanes past l read column 0 */
#pragma unroll
            for (int r = 0; r < SVM_R; r++) {
                const int k = tid + r * T;
                const bool pos = k < n0;
                kfi[KEEP ? r : 0] = (float)kik[r];
                const bool below_C = al[ALS ? 0 : r] < C, above_0 = al[ALS ? 0 : r] > 0.0;
                const bool in_low = (k < l) & ((pos & above_0) | (!pos & below_C));
                const double gs = G[r]; /* y_k G_k */
                g2max = (in_low & (gs > g2max)) ? gs : g2max;
                const double grad_diff = Gmax + gs;
                const double quad = (QDi + QD(r)) - 2.0 * (double)kfi[KEEP ? r : 0];
                const double od = -(grad_diff * grad_diff) / (quad > 0.0 ? quad : SVM_TAU);
                const bool take = in_low & (grad_diff > 0.0) & (od <= mv);
                mv = take ? od : mv;
                mk = take ? k : mk;
            }
        } else {
#pragma unroll
        for (int r0 = 0; r0 < SVM_R; r0 += BATCH) {
        double kik[BATCH], qdr[QDG ? BATCH : 1];
#pragma unroll
        for (int r = r0; r < r0 + BATCH; r++) { /* lanes past l read column 0 */
            kik[r - r0] = Ki[GI(r)];
            if (QDG) qdr[QDG ? r - r0 : 0] = QD(r);
        }
        /* 2 y_i Q_ik = +-2 (float)K_ik with the sign of y_k, so LIBSVM's two quad_coef expressions are
         * both (QD_i + QD_k) - 2 (float)K_ik, bit for bit; the loop is branch-free (selects), one IEEE
         * division per sample */
#pragma unroll
        for (int r = r0; r < r0 + BATCH; r++) {
            const int k = tid + r * T;
            const bool pos = k < n0;
            const float kf = (float)kik[r - r0];
            if (KEEP) kfi[KEEP ? r : 0] = kf;
            const double a_r = AL(r);
            const bool below_C = a_r < C, above_0 = a_r > 0.0;
            const bool in_low = (k < l) & ((pos & above_0) | (!pos & below_C));
            const double gs = G[r]; /* y_k G_k */
            g2max = (in_low & (gs > g2max)) ? gs : g2max;
            const double grad_diff = Gmax + gs;
            const double quad = (QDi + (QDG ? qdr[QDG ? r - r0 : 0] : QD(r))) - 2.0 * (double)kf;
            const double od = -(grad_diff * grad_diff) / (quad > 0.0 ? quad : SVM_TAU);
            const bool take = in_low & (grad_diff > 0.0) & (od <= mv);
            mv = take ? od : mv;
            mk = take ? k : mk;
            if (!KEEP) mkf = take ? kf : mkf;
        }
        if (ALS) __builtin_amdgcn_sched_barrier(0);
        }
        }
        PROF(2)
        g2max = wave_max(g2max);
        wave_select<true>(mv, mk);
        if (lane == 0) g2s[wave] = g2max;
        if (mk < 0) {
            if (lane == 0) candB[wave].k = -1;
        } else {
            const int rr = __builtin_amdgcn_readfirstlane(mk) / T;
            const bool owner = (mk & 63) == lane;
#pragma unroll
            for (int r = 0; r < SVM_R; r++)
                if (r == rr && owner)
                    candB[wave] = {mv, AL(r), (mk < n0 ? G[r] : -G[r]), TAB || QDG ? 0.0 : qd[TAB || QDG ? 0 : r],
                                   (double)((yi > 0.0) == (mk < n0) ? (KEEP ? kfi[KEEP ? r : 0] : mkf) : -(KEEP ? kfi[KEEP ? r : 0] : mkf)), mk, TAB ? 0 : QDG ? GI(r) : gidx[TAB || QDG ? 0 : r]};
        }
        __syncthreads();
        wk = candB[lane & (NW - 1)].k;
        wv = wk >= 0 ? candB[lane & (NW - 1)].v : INFINITY;
        const int wj = row_select<true>(wv, wk) & (NW - 1);
        Cand cj = candB[wj];
        cj.k = wk;
        const double Gmax2 = row_max(g2s[lane & (NW - 1)]);
        PROF(3)
        const int j = cj.k;
        if (Gmax + Gmax2 < eps || j < 0) break;

        /* ---- the two-variable sub-problem, LIBSVM's clipping order (every thread, same result) ---- */
        const double yj = j < n0 ? 1.0 : -1.0;
        const double Qij = cj.q, QDj = QDG ? diag[cj.g] : TAB ? qd_s[j] : cj.qd;
        double ai = ci.alpha, aj = cj.alpha;
        const double old_i = ai, old_j = aj;
        {
            /* LIBSVM's two branches (y_i != y_j / y_i == y_j) differ in signs and in which bound is
             * tested first; written with selects (negation is exact, a - b == a + (-b)), so the wave
             * executes ~60 instructions instead of both branchy variants with their register copies */
            const bool opp = yi != yj;
            const double q2 = 2 * Qij;
            double quad = (QDi + QDj) + (opp ? q2 : -q2);
            quad = quad <= 0 ? SVM_TAU : quad;
            const double delta = ((opp ? -ci.G : ci.G) - cj.G) / quad;
            const double diff = ai - aj, sum = ai + aj;
            ai += opp ? delta : -delta;
            aj += delta;
            const bool c1 = opp ? diff > 0 : sum > C;
            /* first clipping */
            {
                const bool hit = opp ? (c1 ? aj < 0 : ai < 0) : (c1 ? ai > C : aj < 0);
                const double ni = opp ? (c1 ? diff : 0.0) : (c1 ? C : sum);
                const double nj = opp ? (c1 ? 0.0 : -diff) : (c1 ? sum - C : 0.0);
                ai = hit ? ni : ai;
                aj = hit ? nj : aj;
            }
            /* second clipping */
            {
                const bool hit = opp ? (c1 ? ai > C : aj > C) : (c1 ? aj > C : ai < 0);
                const double ni = opp ? (c1 ? C : C + diff) : (c1 ? sum - C : 0.0);
                const double nj = opp ? (c1 ? C - diff : C) : (c1 ? C : sum);
                ai = hit ? ni : ai;
                aj = hit ? nj : aj;
            }
        }
        const double dai = ai - old_i, daj = aj - old_j;

        /* ---- gradient; the owners store the new alpha ---- */
        const double *Kj = K + (int64_t)(TAB ? gidx_s[j] : cj.g) * ld;
        double kja[KEEP ? SVM_R : 1]; /* (the gather of row j goes out before anything else of this phase) */
        if (KEEP) {
#pragma unroll
            for (int r = 0; r < SVM_R; r++) kja[KEEP ? r : 0] = Kj[GI(r)];
        }
        const double ci_ = yi * dai, cj_ = yj * daj;
        const int iu = __builtin_amdgcn_readfirstlane(i), ju = __builtin_amdgcn_readfirstlane(j);
        const int ri = iu / T, rj = ju / T;
        const bool mine_i = (iu & (T - 1)) == tid, mine_j = (ju & (T - 1)) == tid;
#pragma unroll
        for (int r0 = 0; r0 < SVM_R; r0 += BATCH) {
        double kj[KEEP ? 1 : BATCH], ki2[KEEP ? 1 : BATCH];
        if (!KEEP) {
#pragma unroll
            for (int r = r0; r < r0 + BATCH; r++) {
                kj[KEEP ? 0 : r - r0] = Kj[GI(r)];
                ki2[KEEP ? 0 : r - r0] = Ki[GI(r)];
            }
        }
#pragma unroll
        for (int r = r0; r < r0 + BATCH; r++) { /* (lanes past l update a gradient nobody reads) */
            const float kf = KEEP ? kfi[KEEP ? r : 0] : (float)ki2[KEEP ? 0 : r - r0];
            G[r] += (double)kf * ci_ + (double)(float)(KEEP ? kja[KEEP ? r : 0] : kj[KEEP ? 0 : r - r0]) * cj_;
            /* (i and j are wave-uniform: scalar tests pick the one register each of them lives in) */
            if (ALS) {
                if (r == ri && mine_i) al_s[tid + r * T] = ai;
                if (r == rj && mine_j) al_s[tid + r * T] = aj;
            } else {
                if (r == ri) al[ALS ? 0 : r] = mine_i ? ai : al[ALS ? 0 : r];
                if (r == rj) al[ALS ? 0 : r] = mine_j ? aj : al[ALS ? 0 : r];
            }
        }
        if (ALS) __builtin_amdgcn_sched_barrier(0);
        }
        PROF(4)
    }
#ifdef SVM_PROF
    if (tid == 0 && blockIdx.x == 0)
        printf("iters %d cycles/iter: scanA %lld selA %lld gatherB %lld selB %lld updD %lld loop %lld\n", iter,
               tp[0] / iter, tp[1] / iter, tp[2] / iter, tp[3] / iter, tp[4] / iter, tp[5] / iter);
#endif

#pragma unroll
    for (int r = 0; r < SVM_R; r++) {
        const int k = tid + r * T;
        if (k < l) { as_global(p.alpha)[k] = AL(r); as_global(p.grad)[k] = k < n0 ? G[r] : -G[r]; }
    }

    /* rho (LIBSVM calculate_rho).  The mean over the free vectors is summed in index order by one
     * thread (bit-identical to the sequential solver); the values pass through LDS T at a time. */
    int nr_free = 0;
    double ub = INFINITY, lb = -INFINITY, sum_free = 0;
#pragma unroll
    for (int r = 0; r < SVM_R; r++) {
        if (r * T >= l) break;
        const int k = tid + r * T;
        const double y = k < n0 ? 1.0 : -1.0;
        double yG = G[r]; /* S = y G */
        bool is_free = false;
        if (k < l) {
            if (AL(r) >= C) {
                if (y < 0) ub = fmin(ub, yG); else lb = fmax(lb, yG);
            } else if (AL(r) <= 0) {
                if (y > 0) ub = fmin(ub, yG); else lb = fmax(lb, yG);
            } else {
                is_free = true;
            }
        }
        __syncthreads();
        chunk[tid] = is_free ? yG : NAN; /* NaN marks "not free" (a gradient is never NaN here) */
        __syncthreads();
        if (tid == 0) {
            const int m = l - r * T < T ? l - r * T : T;
            for (int t = 0; t < m; t++) {
                const double x = chunk[t];
                if (x == x) { ++nr_free; sum_free += x; }
            }
        }
    }
    /* ub / lb: plain min / max, any order */
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        ub = fmin(ub, __shfl_xor(ub, s));
        lb = fmax(lb, __shfl_xor(lb, s));
    }
    __syncthreads();
    if (lane == 0) { chunk[wave] = ub; chunk[NW + wave] = lb; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 0; w < NW; w++) { ub = fmin(ub, chunk[w]); lb = fmax(lb, chunk[NW + w]); }
        *p.rho = nr_free > 0 ? sum_free / nr_free : (ub + lb) / 2;
        *p.iters = iter;
    }
}


/* ------------------------------------------------------------------------------------------------
 * The general solver: LIBSVM's Solver::Solve as written (sklearn/svm/src/libsvm/svm.cpp), WITH the
 * shrinking heuristic (`--shrinking 1`, reference scripts/gkmsvm.py:110-118 passes it through to SVC)
 * and without the 16 384-sample limit of k_smo.  State lives in global memory (L2-resident: 45 bytes per
 * sample), one workgroup of 1024 threads per fold, positions dealt to the threads round-robin.
 *
 * Shrinking permutes the problem (Solver::swap_index), and LIBSVM's selections scan positions in
 * order and keep the LAST of equal candidates -- so the permutation has to be LIBSVM's own:
 *   do_shrinking   Gmax1/Gmax2 by block reductions (a maximum has no order), be_shrunk() of every
 *                  active position in parallel, then ONE thread walks LIBSVM's two-pointer loop over
 *                  those flags (in LDS) and writes down the swaps, which all threads apply;
 *   reconstruct_gradient   for an inactive sample the sum over the free active samples runs in
 *                  ascending position in both of LIBSVM's loop orders: one thread per inactive
 *                  sample adds them in that order (list of free samples built in position order);
 *   G_bar          updated for all l samples whenever alpha_i or alpha_j reaches or leaves C.
 * Every floating-point expression is LIBSVM's (Qfloat rounding, operation order; -ffp-contract=off).
 * With shrinking = 0 the same kernel is the no-shrinking solver for folds that k_smo cannot hold.
 */
struct GenProb {
    const int *idx;
    int l, n0;
    double *alpha_out, *grad_out, *rho;
    int *iters;
    /* scratch, l entries each (device) */
    int *gidx, *ys, *aset, *swaps; /* swaps: 2 ints per entry */
    double *alpha, *G, *Gbar, *QD, *fa; /* fa: alpha of the free active samples, in position order */
    int *fg, *fy;                       /* their matrix index and label */
    float *Qi, *Qj;
};

/* threads per fold: template parameter GEN_T of k_smo_general -- 1024, or 512 for folds of at most 8 192 samples */
constexpr int GEN_U = 8; /* positions of a thread whose loads are in flight together */
/* LDS_STATE (round 4): for folds of at most GEN_LDS_L samples the state the three scans of an iteration read -- G, Q_i,
 * the matrix index and one byte of "label, alpha at 0, alpha at C" per position -- lives in LDS (18 bytes per sample
 * + the shrinking flag), not in L2: a scan then costs LDS reads and ONE dependent trip to the matrix row.  alpha itself
 * (read at i and j only), QD, G_bar and the rest stay in global memory. */
constexpr int GEN_LDS_L = 8192;
#ifndef GEN_U_1024
#define GEN_U_1024 4 /* the 1 024-thread variant has 128 VGPRs: 8 at a time spilt 24 of them (5 folds x 17 600 samples: 0.93 s, 4: 0.79 s) */
#endif
#ifndef GEN_U_LDS
#define GEN_U_LDS 16 /* 512 threads x 16 = every position of such a fold in ONE round of loads */
#endif
constexpr unsigned ST_LO = 1u, ST_UP = 2u, ST_POS = 4u;
template <bool B, class Ta, class Tb>
__device__ __forceinline__ auto pick(Ta a, Tb b)
{
    if constexpr (B) return a; else return b;
}

struct GenSel {
    double v;
    int k;
};

/* block-wide best (value, position) with LIBSVM's tie rule; every thread returns the same pair */
/* ONE barrier and a DPP row selection over the wave winners (k_smo's way; round 4 -- before: two barriers and a scalar scan
 * of the slots in every thread, 7-14 k cycles per selection, profiles/r4_svm_general_phases.txt).  `slots` must not be
 * the array of the previous selection: every call site has its own, so that whoever still reads the previous one is at
 * least one barrier behind. */
template <bool MINIMISE, int GEN_NW>
__device__ __forceinline__ void block_select(double &v, int &k, GenSel *slots)
{
    static_assert(GEN_NW <= 16 && (GEN_NW & (GEN_NW - 1)) == 0, "the wave winners fit one DPP row of 16 lanes");
    wave_select<MINIMISE>(v, k);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { slots[wave].v = v; slots[wave].k = k; }
    __syncthreads();
    int wk = slots[lane & (GEN_NW - 1)].k;
    double wv = wk >= 0 ? slots[lane & (GEN_NW - 1)].v : (MINIMISE ? INFINITY : -INFINITY);
    (void)row_select<MINIMISE>(wv, wk);
    v = wv;
    k = wk;
}
template <int GEN_NW>
__device__ __forceinline__ double block_max(double v, double *slots)
{
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) slots[wave] = v;
    __syncthreads();
    double m = slots[0];
#pragma unroll
    for (int w = 1; w < GEN_NW; w++) m = fmax(m, slots[w]);
    return m;
}

/* block_select<true> and block_max in ONE exchange (one pair of barriers instead of two): the second selection of
 * select_working_set needs both the best (obj_diff, position) and Gmax2 */
template <int GEN_NW>
__device__ __forceinline__ void block_select_min_and_max(double &v, int &k, double &mx, GenSel *slots, double *mslots)
{
    wave_select<true>(v, k);
    mx = wave_max(mx);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { slots[wave].v = v; slots[wave].k = k; mslots[wave] = mx; }
    __syncthreads();
    int wk = slots[lane & (GEN_NW - 1)].k;
    double wv = wk >= 0 ? slots[lane & (GEN_NW - 1)].v : INFINITY;
    (void)row_select<true>(wv, wk);
    v = wv;
    k = wk;
    mx = row_max(mslots[lane & (GEN_NW - 1)]);
}

template <int GEN_T, bool LDS_STATE>
__global__ __launch_bounds__(GEN_T) void k_smo_general(const double *__restrict__ K, int64_t ld,
                                                       const double *__restrict__ diag, const GenProb *probs, double C,
                                                       double eps, int max_iter, int shrinking, int cap)
{
    constexpr int GEN_NW = GEN_T / 64;
    constexpr int GEN_U = LDS_STATE ? GEN_U_LDS : GEN_T == 1024 ? GEN_U_1024 : ::GEN_U;
    __shared__ GenSel sel_s[GEN_NW], selb_s[GEN_NW]; /* first / second selection: one array each (see block_select) */
    __shared__ double max_s[GEN_NW], maxb_s[GEN_NW];
    __shared__ double chunk[GEN_T];
    __shared__ int chunk_g[GEN_T], chunk_y[GEN_T];
    __shared__ int misc[4];
    /* dynamic LDS, `cap` = the largest fold of the launch rounded up to 16: cap bytes of be_shrunk() flags of the active
     * positions; with LDS_STATE then cap state bytes, cap doubles G, cap floats Q_i, cap ints matrix index */
    extern __shared__ __attribute__((aligned(16))) unsigned char flag_s[];
    unsigned char *const st_s = flag_s + cap;
    const GenProb p = probs[blockIdx.x];
    const int tid = threadIdx.x, l = p.l;
    auto *const gidx = pick<LDS_STATE>((int *)(flag_s + (size_t)14 * cap), as_global(p.gidx));
    auto *const ys = as_global(p.ys), *const aset = as_global(p.aset);
    auto *const alpha = as_global(p.alpha), *const Gbar = as_global(p.Gbar), *const QD = as_global(p.QD);
    auto *const G = pick<LDS_STATE>((double *)(flag_s + (size_t)2 * cap), as_global(p.G));
    auto *const Qi = pick<LDS_STATE>((float *)(flag_s + (size_t)10 * cap), as_global(p.Qi));
    auto *const Qj = as_global(p.Qj);

    auto is_upper = [&](double a) { return a >= C; };
    auto is_lower = [&](double a) { return a <= 0.0; };
    auto encode = [&](double a, int y) -> unsigned { return (a <= 0.0 ? ST_LO : 0u) | (a >= C ? ST_UP : 0u) | (y > 0 ? ST_POS : 0u); };
    /* what the scans need of position k: is_lower_bound, is_upper_bound, the label */
    auto st_load = [&](int k) -> unsigned {
        if constexpr (LDS_STATE) return st_s[k];
        else return encode(alpha[k], ys[k]);
    };

    for (int k = tid; k < l; k += GEN_T) {
        const int g = p.idx[k];
        gidx[k] = g;
        ys[k] = k < p.n0 ? 1 : -1;
        aset[k] = k;
        alpha[k] = 0.0;
        G[k] = -1.0; /* p = -1 */
        Gbar[k] = 0.0;
        QD[k] = diag[g];
        if constexpr (LDS_STATE) st_s[k] = (unsigned char)encode(0.0, k < p.n0 ? 1 : -1);
    }
    __syncthreads();

    /* Q_t[k] = (Qfloat)(y_t y_k K_tk) for k in [k0, k1) */
    auto q_row = [&](int t, auto *out, int k0, int k1) {
        const double *Kt = K + (int64_t)gidx[t] * ld;
        const int yt = ys[t];
        /* GEN_U positions of a thread at a time, the loads of one kind issued together: the state lives in global
         * memory (L2), and a loop that uses each value as soon as it is loaded pays one round trip per position */
        for (int kb = k0 + tid; kb < k1; kb += GEN_T * GEN_U) {
            int gi[GEN_U], yk[GEN_U];
            double kv[GEN_U];
#pragma unroll
            for (int u = 0; u < GEN_U; u++) {
                const int k = kb + u * GEN_T, kc = k < k1 ? k : kb;
                gi[u] = gidx[kc];
                yk[u] = ys[kc];
            }
#pragma unroll
            for (int u = 0; u < GEN_U; u++) kv[u] = Kt[gi[u]];
#pragma unroll
            for (int u = 0; u < GEN_U; u++) {
                const int k = kb + u * GEN_T;
                if (k < k1) out[k] = (float)((double)(yt * yk[u]) * kv[u]);
            }
        }
    };

#ifdef SVM_PROF
    long long gp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, g0 = clock64(), g1;
#define GPROF(n) g1 = clock64(); gp[n] += g1 - g0; g0 = g1;
#else
#define GPROF(n)
#endif
    int active = l;
    /* Solver::reconstruct_gradient */
    auto reconstruct = [&]() {
        if (active == l) return;
        /* the free active samples, in position order (thread 0 appends, a block of positions at a time) */
        __syncthreads();
        if (tid == 0) misc[0] = 0;
        for (int k0 = 0; k0 < active; k0 += GEN_T) {
            const int k = k0 + tid;
            const double a = k < active ? alpha[k] : 0.0;
            __syncthreads();
            chunk[tid] = (k < active && !is_upper(a) && !is_lower(a)) ? a : -1.0;
            chunk_g[tid] = k < active ? gidx[k] : 0;
            chunk_y[tid] = k < active ? ys[k] : 0;
            __syncthreads();
            if (tid == 0) {
                int nf = misc[0];
                const int m = active - k0 < GEN_T ? active - k0 : GEN_T;
                for (int t = 0; t < m; t++)
                    if (chunk[t] >= 0.0) { p.fa[nf] = chunk[t]; p.fg[nf] = chunk_g[t]; p.fy[nf] = chunk_y[t]; nf++; }
                misc[0] = nf;
            }
        }
        __threadfence_block();
        __syncthreads();
        const int nf = misc[0];
        for (int j = active + tid; j < l; j += GEN_T) {
            double g = Gbar[j] + (-1.0);
            const int gj = gidx[j], yj = ys[j];
            for (int f0 = 0; f0 < nf; f0 += 8) { /* 8 kernel values in flight, the additions in order */
                double kv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) kv[u] = K[(int64_t)p.fg[f0 + u < nf ? f0 + u : nf - 1] * ld + gj];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (f0 + u < nf) g += p.fa[f0 + u] * (double)(float)((double)(p.fy[f0 + u] * yj) * kv[u]);
            }
            G[j] = g;
        }
        __syncthreads();
    };

    /* Solver::select_working_set; 0 = pair found.  Leaves Q_i[0..active) in Qi. */
    int sel_yi = 0;                  /* y_i, QD_i, alpha_i of the pair select() found */
    double sel_QDi = 0, sel_ai = 0;
    auto select = [&](int &out_i, int &out_j) -> int {
        double gm = -INFINITY;
        int gi = -1;
        /* LDS_STATE (one round of loads per scan): the diagonal entries the SECOND scan needs are asked for here, a scan
         * and a block selection ahead -- beside the gather of row i they queued behind it in the CU's in-order L1
         * (26.2 -> 24.2 k cycles per iteration, although the 32 registers spill 11) */
        double qd_ahead[LDS_STATE ? GEN_U : 1];
        if constexpr (LDS_STATE) {
#pragma unroll
            for (int u = 0; u < GEN_U; u++) {
                const int k = tid + u * GEN_T;
                qd_ahead[u] = QD[k < active ? k : 0];
            }
        }
        for (int kb = tid; kb < active; kb += GEN_T * GEN_U) {
            double gv[GEN_U];
            unsigned sv[GEN_U];
#pragma unroll
            for (int u = 0; u < GEN_U; u++) {
                const int k = kb + u * GEN_T, kc = k < active ? k : kb;
                gv[u] = G[kc]; sv[u] = st_load(kc);
            }
#pragma unroll
            for (int u = 0; u < GEN_U; u++) { /* ascending k inside the thread: ">=" keeps LIBSVM's last-of-equals */
                const int k = kb + u * GEN_T;
                if (k >= active) continue;
                /* y = +1: not at C, -G;  y = -1: not at 0, +G -- one comparison for both labels */
                const bool pos = (sv[u] & ST_POS) != 0;
                const double val = pos ? -gv[u] : gv[u];
                if (!(sv[u] & (pos ? ST_UP : ST_LO)) && val >= gm) { gm = val; gi = k; }
            }
        }
        GPROF(0)
        block_select<false, GEN_NW>(gm, gi, sel_s);
        GPROF(1)
        const int i = gi;
        if (i < 0) return 1;
        /* Row i is gathered INSIDE the second scan (round 4; q_row() + a second pass that read Q_i back cost a write,
         * a read and their round trip to L2 per iteration): the thread that computes Q_i[k] uses it at once and stores it
         * for the gradient update and for Q_ij.  Same values, same order of the scan. */
        const double *const Ki = K + (int64_t)gidx[i] * ld;
        int yi_;
        if constexpr (LDS_STATE) yi_ = (st_s[i] & ST_POS) ? 1 : -1;
        else yi_ = ys[i];
        const double QDi = QD[i];
        sel_yi = yi_; sel_QDi = QDi; sel_ai = alpha[i]; /* (alpha_i is asked for here: it is back long before the update) */
        double gm2 = -INFINITY, omin = INFINITY;
        int gj = -1;
        for (int kb = tid; kb < active; kb += GEN_T * GEN_U) {
          double gv[GEN_U], qdv[GEN_U], kv[GEN_U];
          float qiv[GEN_U];
          unsigned sv[GEN_U];
          int giv[GEN_U];
#pragma unroll
          for (int u = 0; u < GEN_U; u++) { /* the matrix index first: the trip to row i hangs on it */
              const int k = kb + u * GEN_T, kc = k < active ? k : kb;
              giv[u] = gidx[kc];
          }
#pragma unroll
          for (int u = 0; u < GEN_U; u++) kv[u] = Ki[giv[u]];
#pragma unroll
          for (int u = 0; u < GEN_U; u++) {
              const int k = kb + u * GEN_T, kc = k < active ? k : kb;
              gv[u] = G[kc]; sv[u] = st_load(kc);
              if constexpr (LDS_STATE) qdv[u] = qd_ahead[u];
              else qdv[u] = QD[kc];
          }
          /* Branch-free, one IEEE division per position (k_smo's form of the same arithmetic; a wave holds both labels,
           * and LIBSVM's two branches -- a double division each -- ran one after the other):
           *   Q_ik = (Qfloat)(y_i y_k K_ik) = +-(Qfloat)K_ik (rounding is symmetric);
           *   with s = y_k: gm + G / gm - G is gm + sG, and -G >= Gmax2 / G >= Gmax2 is sG >= Gmax2;
           *   2 y_i Q_ik = s 2 (Qfloat)K_ik exactly, so QD_i + QD_k -/+ 2 y_i Q_ik is (QD_i + QD_k) - 2 (Qfloat)K_ik for
           *   both labels (a - b and a + (-b) are the same IEEE operation);
           *   the quotient by quad_coef > 0 ? quad_coef : TAU is LIBSVM's either way. */
#pragma unroll
          for (int u = 0; u < GEN_U; u++) {
              const int k = kb + u * GEN_T;
              const bool pos = (sv[u] & ST_POS) != 0;
              const float kf = (float)kv[u];
              qiv[u] = (yi_ > 0) == pos ? kf : -kf;
              if (k < active) Qi[k] = qiv[u];
              const bool in_low = (k < active) & !(sv[u] & (pos ? ST_LO : ST_UP));
              const double gs = pos ? gv[u] : -gv[u];
              gm2 = (in_low & (gs >= gm2)) ? gs : gm2;
              const double grad_diff = gm + gs;
              const double quad_coef = (QDi + qdv[u]) - 2.0 * (double)kf;
              const double obj_diff = -(grad_diff * grad_diff) / (quad_coef > 0 ? quad_coef : SVM_TAU);
              const bool take = in_low & (grad_diff > 0) & (obj_diff <= omin);
              omin = take ? obj_diff : omin;
              gj = take ? k : gj;
          }
        }
        GPROF(2)
        double Gmax2 = gm2;
        block_select_min_and_max<GEN_NW>(omin, gj, Gmax2, selb_s, maxb_s);
        GPROF(3)
        if (gm + Gmax2 < eps || gj < 0) return 1;
        out_i = i;
        out_j = gj;
        return 0;
    };

    /* Solver::do_shrinking */
    bool unshrink = false;
    auto do_shrinking = [&]() {
        double g1 = -INFINITY, g2 = -INFINITY;
        for (int k = tid; k < active; k += GEN_T) {
            const double a = alpha[k], g = G[k];
            if (ys[k] == +1) {
                if (!is_upper(a) && -g >= g1) g1 = -g;
                if (!is_lower(a) && g >= g2) g2 = g;
            } else {
                if (!is_upper(a) && -g >= g2) g2 = -g;
                if (!is_lower(a) && g >= g1) g1 = g;
            }
        }
        const double Gmax1 = block_max<GEN_NW>(g1, max_s), Gmax2 = block_max<GEN_NW>(g2, max_s);
        if (!unshrink && Gmax1 + Gmax2 <= eps * 10) {
            unshrink = true;
            reconstruct();
            active = l;
        }
        __syncthreads();
        for (int k = tid; k < active; k += GEN_T) {
            const double a = alpha[k], g = G[k];
            bool sh = false;
            if (is_upper(a)) sh = ys[k] == +1 ? -g > Gmax1 : -g > Gmax2;
            else if (is_lower(a)) sh = ys[k] == +1 ? g > Gmax2 : g > Gmax1;
            flag_s[k] = sh ? 1 : 0;
        }
        __syncthreads();
        if (tid == 0) { /* LIBSVM's loop, on the flags; a swap exchanges the flags too */
            int as = active, ns = 0;
            for (int i = 0; i < as; i++)
                if (flag_s[i]) {
                    as--;
                    while (as > i) {
                        if (!flag_s[as]) {
                            p.swaps[2 * ns] = i; p.swaps[2 * ns + 1] = as; ns++;
                            flag_s[i] = 0; flag_s[as] = 1;
                            break;
                        }
                        as--;
                    }
                }
            misc[1] = as;
            misc[2] = ns;
        }
        __threadfence_block();
        __syncthreads();
        active = misc[1];
        const int ns = misc[2];
        for (int s = tid; s < ns; s += GEN_T) { /* Solver::swap_index: the pairs are disjoint */
            const int a = p.swaps[2 * s], b = p.swaps[2 * s + 1];
            { const int t = gidx[a]; gidx[a] = gidx[b]; gidx[b] = t; }
            { const int t = ys[a]; ys[a] = ys[b]; ys[b] = t; }
            { const int t = aset[a]; aset[a] = aset[b]; aset[b] = t; }
            { const double t = alpha[a]; alpha[a] = alpha[b]; alpha[b] = t; }
            if constexpr (LDS_STATE) { const unsigned char t = st_s[a]; st_s[a] = st_s[b]; st_s[b] = t; }
            { const double t = G[a]; G[a] = G[b]; G[b] = t; }
            { const double t = Gbar[a]; Gbar[a] = Gbar[b]; Gbar[b] = t; }
            { const double t = QD[a]; QD[a] = QD[b]; QD[b] = t; }
        }
        __syncthreads();
    };

    int iter = 0;
    int counter = (l < 1000 ? l : 1000) + 1;
    for (;;) {
        if (iter >= max_iter) {
            if (active < l) { reconstruct(); active = l; }
            iter = -iter;
            break;
        }
        if (--counter == 0) {
            counter = l < 1000 ? l : 1000;
            if (shrinking) do_shrinking();
        }
        int i = -1, j = -1;
        if (select(i, j) != 0) {
            reconstruct();
            active = l;
            if (select(i, j) != 0) break;
            counter = 1; /* do shrinking next iteration */
        }
        ++iter;
        /* the two-variable update: every thread computes the same scalars.  (Row j is gathered inside the gradient
         * update below, not stored and read back: round 4.)  Qi[j], written by another thread in select(), is visible:
         * the barrier of the second selection came after every thread's writes. */
        const int yi = sel_yi;
        int yj;
        if constexpr (LDS_STATE) yj = (st_s[j] & ST_POS) ? 1 : -1;
        else yj = ys[j];
        const double old_ai = sel_ai, old_aj = alpha[j], QD_i = sel_QDi, QD_j = QD[j];
        const double *const Kj = K + (int64_t)gidx[j] * ld;
        /* LDS_STATE: one round of loads covers the fold, so the trip to row j starts HERE, under the scalar update (after
         * the loads of the scalars: loads return in order) */
        static_assert(!LDS_STATE || GEN_T * GEN_U >= GEN_LDS_L, "one round of loads covers an LDS-resident fold");
        double kvj[LDS_STATE ? GEN_U : 1];
        if constexpr (LDS_STATE) {
#pragma unroll
            for (int u = 0; u < GEN_U; u++) {
                const int k = tid + u * GEN_T;
                kvj[u] = Kj[gidx[k < active ? k : 0]];
            }
        }
        double ai = old_ai, aj = old_aj;
        const double Qij = (double)Qi[j];
        if (yi != yj) {
            double quad_coef = QD_i + QD_j + 2 * Qij;
            if (quad_coef <= 0) quad_coef = SVM_TAU;
            const double delta = (-G[i] - G[j]) / quad_coef;
            const double diff = ai - aj;
            ai += delta;
            aj += delta;
            if (diff > 0) {
                if (aj < 0) { aj = 0; ai = diff; }
            } else {
                if (ai < 0) { ai = 0; aj = -diff; }
            }
            if (diff > C - C) {
                if (ai > C) { ai = C; aj = C - diff; }
            } else {
                if (aj > C) { aj = C; ai = C + diff; }
            }
        } else {
            double quad_coef = QD_i + QD_j - 2 * Qij;
            if (quad_coef <= 0) quad_coef = SVM_TAU;
            const double delta = (G[i] - G[j]) / quad_coef;
            const double sum = ai + aj;
            ai -= delta;
            aj += delta;
            if (sum > C) {
                if (ai > C) { ai = C; aj = sum - C; }
            } else {
                if (aj < 0) { aj = 0; ai = sum; }
            }
            if (sum > C) {
                if (aj > C) { aj = C; ai = sum - C; }
            } else {
                if (ai < 0) { ai = 0; aj = sum; }
            }
        }
        const double dai = ai - old_ai, daj = aj - old_aj;
        __syncthreads(); /* everybody has read G[i], G[j], alpha[i], alpha[j] */
        GPROF(4)
        if constexpr (LDS_STATE) {
#pragma unroll
            for (int u = 0; u < GEN_U; u++) {
                const int k = tid + u * GEN_T;
                if (k < active) {
                    const float qj = (float)((double)(yj * ((st_s[k] & ST_POS) ? 1 : -1)) * kvj[u]); /* (Qfloat)(y_j y_k K_jk) */
                    G[k] = G[k] + ((double)Qi[k] * dai + (double)qj * daj);
                }
            }
        } else
        for (int kb = tid; kb < active; kb += GEN_T * GEN_U) {
            double gv[GEN_U], kv[GEN_U];
            float qiv[GEN_U];
            int yv[GEN_U], giv[GEN_U];
#pragma unroll
            for (int u = 0; u < GEN_U; u++) {
                const int k = kb + u * GEN_T, kc = k < active ? k : kb;
                giv[u] = gidx[kc];
            }
#pragma unroll
            for (int u = 0; u < GEN_U; u++) kv[u] = Kj[giv[u]];
#pragma unroll
            for (int u = 0; u < GEN_U; u++) {
                const int k = kb + u * GEN_T, kc = k < active ? k : kb;
                gv[u] = G[kc]; qiv[u] = Qi[kc];
                if constexpr (LDS_STATE) yv[u] = (st_s[kc] & ST_POS) ? 1 : -1;
                else yv[u] = ys[kc];
            }
#pragma unroll
            for (int u = 0; u < GEN_U; u++) {
                const int k = kb + u * GEN_T;
                const float qj = (float)((double)(yj * yv[u]) * kv[u]); /* (Qfloat)(y_j y_k K_jk) */
                if (k < active) G[k] = gv[u] + ((double)qiv[u] * dai + (double)qj * daj);
            }
        }
        /* written by the threads that scan these positions (k = tid + u GEN_T everywhere): see the end of the loop */
        if (tid == (i & (GEN_T - 1))) {
            alpha[i] = ai;
            if constexpr (LDS_STATE) st_s[i] = (unsigned char)encode(ai, yi);
        }
        if (tid == (j & (GEN_T - 1))) {
            alpha[j] = aj;
            if constexpr (LDS_STATE) st_s[j] = (unsigned char)encode(aj, yj);
        }
        GPROF(5)
        if (shrinking) { /* G_bar only matters to reconstruct_gradient */
            const bool ui = is_upper(old_ai), uj = is_upper(old_aj);
            if (ui != is_upper(ai)) {
                __syncthreads();
                /* Q_i of the active positions is in Qi since the second scan: only the shrunk ones are gathered (from
                 * a multiple of GEN_T on, so that position k stays with thread k mod GEN_T and no barrier is needed) */
                q_row(i, Qi, active / GEN_T * GEN_T, l);
                for (int k = tid; k < l; k += GEN_T) Gbar[k] = ui ? Gbar[k] - C * (double)Qi[k] : Gbar[k] + C * (double)Qi[k];
            }
            if (uj != is_upper(aj)) {
                __syncthreads();
                q_row(j, Qj, 0, l);
                for (int k = tid; k < l; k += GEN_T) Gbar[k] = uj ? Gbar[k] - C * (double)Qj[k] : Gbar[k] + C * (double)Qj[k];
            }
        }
        /* Without shrinking no barrier is needed here: every scan and the gradient update deal position k to thread k mod GEN_T,
         * so the first scan of the next iteration reads only what this thread wrote (G, the state byte, alpha); what all
         * threads read of other positions (alpha, G, QD at i and j; Qi[j]) is read after the next selections' barriers. */
        if (shrinking) __syncthreads();
        GPROF(6)
    }

#ifdef SVM_PROF
    if (tid == 0 && blockIdx.x == 0 && iter != 0) {
        const int it = iter < 0 ? -iter : iter;
        printf("k_smo_general<%d,%d> iters %d cycles/iter: scan1 %lld select1 %lld gather_i+scan2 %lld select2 %lld scalars %lld gather_j+update %lld rest %lld\n",
               GEN_T, (int)LDS_STATE, it, gp[0] / it, gp[1] / it, gp[2] / it, gp[3] / it, gp[4] / it, gp[5] / it, gp[6] / it);
    }
#endif
    /* Solver::calculate_rho over all l positions (active == l here), the free samples summed in position order */
    __syncthreads();
    int nr_free = 0;
    double ub = INFINITY, lb = -INFINITY, sum_free = 0;
    for (int k0 = 0; k0 < l; k0 += GEN_T) {
        const int k = k0 + tid;
        bool is_free = false;
        double yG = 0;
        if (k < l) {
            const double a = alpha[k];
            yG = (double)ys[k] * G[k];
            if (is_upper(a)) {
                if (ys[k] == -1) ub = fmin(ub, yG); else lb = fmax(lb, yG);
            } else if (is_lower(a)) {
                if (ys[k] == +1) ub = fmin(ub, yG); else lb = fmax(lb, yG);
            } else {
                is_free = true;
            }
        }
        __syncthreads();
        chunk[tid] = is_free ? yG : NAN;
        __syncthreads();
        if (tid == 0) {
            const int m = l - k0 < GEN_T ? l - k0 : GEN_T;
            for (int t = 0; t < m; t++) {
                const double x = chunk[t];
                if (x == x) { ++nr_free; sum_free += x; }
            }
        }
    }
    ub = -block_max<GEN_NW>(-ub, max_s);
    lb = block_max<GEN_NW>(lb, max_s);
    if (tid == 0) {
        *p.rho = nr_free > 0 ? sum_free / nr_free : (ub + lb) / 2;
        *p.iters = iter;
    }
    for (int k = tid; k < l; k += GEN_T) { /* put back the solution */
        p.alpha_out[aset[k]] = alpha[k];
        p.grad_out[aset[k]] = G[k];
    }
}

struct DecProb {
    const int *idx;
    int l, n0;
    const double *alpha, *rho;
    const int *test;
    int ntest;
    double *dec;
};

/* one thread per test sample; the sum runs over the training samples in LIBSVM's order.  K is
 * bit-symmetric (include/gkm_svm.h), so K(test_t, train_k) is read as K[train_k][test_t]: the
 * threads of a wave then read neighbouring addresses of one matrix row.  The loads of DEC_U
 * training samples are in flight together (the index -> matrix row chain would otherwise cost one
 * memory round trip per sample: the call took 5.6 ms instead of 3.5 ms for 5 folds of the headline
 * matrix, host overhead included); the
 * additions stay sequential, in LIBSVM's order. */
constexpr int DEC_U = 16;
__global__ void k_decision(const double *__restrict__ K, int64_t ld, const DecProb *probs)
{
    const DecProb p = probs[blockIdx.y];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.ntest) return;
    const int col = as_global(p.test)[t];
    const auto *const alpha_g = as_global(p.alpha);
    const auto *const idx_g = as_global(p.idx);
    double sum = 0;
    for (int k0 = 0; k0 < p.l; k0 += DEC_U) {
        double a[DEC_U], kv[DEC_U];
#pragma unroll
        for (int u = 0; u < DEC_U; u++) {
            const int k = min(k0 + u, p.l - 1);
            a[u] = k0 + u < p.l ? alpha_g[k] : 0.0; /* wave-uniform */
            kv[u] = K[(int64_t)idx_g[k] * ld + col];
        }
#pragma unroll
        for (int u = 0; u < DEC_U; u++)
            if (a[u] > 0) sum += (k0 + u < p.n0 ? a[u] : -a[u]) * kv[u];
    }
    p.dec[t] = sum - *p.rho;
}

extern "C" int gkmsvm_train_batch(int device, const double *K, int64_t ld, int n, int nprob, const int *idx,
                                  const int64_t *off, const int *n0, double C, double eps, double *alpha, double *grad,
                                  double *rho, int *iters, void *stream_)
{
    if (!K || n <= 0 || ld < n || nprob <= 0 || !idx || !off || !n0 || !alpha || !grad || !rho || !iters) {
        g_svm_err = "gkmsvm_train_batch: bad arguments";
        return 2;
    }
    hipStream_t stream = (hipStream_t)stream_;
    SVMCHK(hipSetDevice(device));
    std::vector<SvmProb> h((size_t)nprob);
    int64_t maxl = 0;
    for (int p = 0; p < nprob; p++) {
        const int64_t l = off[p + 1] - off[p];
        if (l > maxl) maxl = l;
        if (l <= 0 || l > (int64_t)SVM_MAX_L || n0[p] < 0 || n0[p] > l) {
            g_svm_err = "gkmsvm_train_batch: a problem has no samples or more than 16384";
            return 3;
        }
        h[(size_t)p] = {idx + off[p], (int)l, n0[p], alpha + off[p], grad + off[p], rho + p, iters + p};
    }
    const size_t probs_bytes = align256(sizeof(SvmProb) * (size_t)nprob);
    SvmScratch *const scr = scratch_acquire(device, probs_bytes + sizeof(double) * (size_t)n);
    if (!scr) { g_svm_err = "gkmsvm_train_batch: device scratch"; return 4; }
    /* (error returns included: whatever this call enqueued may still be writing the buffer -- k_diag below -- and another
     * solver on another stream of the device may take it from the pool next) */
    struct Release { SvmScratch *b; hipStream_t s; ~Release() { (void)hipStreamSynchronize(s); scratch_release(b); } } release{scr, stream};
    SvmProb *const dprobs = (SvmProb *)scr->p;
    double *const diag = (double *)(scr->p + probs_bytes);
    SVMCHK(hipMemcpyAsync(dprobs, h.data(), sizeof(SvmProb) * (size_t)nprob, hipMemcpyHostToDevice, stream));
    SVMCHK(hipStreamSynchronize(stream)); /* h is a host temporary */
    hipLaunchKernelGGL(k_diag, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, K, ld, n, diag);
    /* fewest threads and registers that hold the largest problem: narrower reductions, fewer waves
     * per barrier, no spills (GKM_SVM_SHAPE=<threads>x<samples per thread> overrides, for timing) */
    /* LIBSVM's classic cap max(10^7, 100 l) is 10^7 for every size this kernel takes; scikit-learn runs
     * without a cap (max_iter = -1), so a fold that stops here (iters < 0) has NOT converged the way the
     * reference would: the callers re-solve such folds with scikit-learn (gkmqc_amd/svmcv.py).
     * GKM_SVM_MAX_ITER lowers the cap (tests). */
    int max_iter = 10000000;
    if (const char *mi = getenv("GKM_SVM_MAX_ITER")) if (atoi(mi) > 0) max_iter = atoi(mi);
    int T = 0, R = 0;
    const char *force = getenv("GKM_SVM_SHAPE");
    if (force) sscanf(force, "%dx%d", &T, &R);
    if (!force || (int64_t)T * R < maxl) {
        if (maxl <= 256 * 4) { T = 256; R = 4; }
        else if (maxl <= 512 * 4) { T = 512; R = 4; }
        else if (maxl <= 512 * 8) { T = 512; R = 8; }
        else if (maxl <= 1024 * 8) { T = 1024; R = 8; } /* (512x16: 113 ms against 99 ms at 8000 samples) */
        else if (maxl <= 1024 * 10) { T = 1024; R = 10; } /* e.g. 10-fold cross-validation of 10 000 sequences */
        else if (maxl <= 1024 * 12) { T = 1024; R = 12; }
        else { T = 1024; R = 16; }
    }
    /* (The 1024-thread shapes take 80-144 KB of dynamic LDS.  This library carries code objects for gfx950 only -- 160 KB of
     * LDS per workgroup -- so there is no device that could load it and refuse that size; should the attribute call fail
     * all the same, the error is reported, and gkmsvm_train_batch_general (no LDS-resident state) solves any fold.) */
#define SMO_LAUNCH(TT, RR, TB)                                                                                  \
    if (T == TT && R == RR) {                                                                                   \
        const size_t dyn = TB == 1 ? (size_t)TT * RR * 12 : TB == 2 ? (size_t)TT * RR * 8 : 0;                  \
        if (dyn > 0 && hipFuncSetAttribute((const void *)k_smo<TT, RR, TB>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                           (int)dyn) != hipSuccess) {                                           \
            (void)hipGetLastError();                                                                            \
            g_svm_err = "k_smo: the device refuses this launch shape (dynamic LDS)";                            \
            return GKMSVM_RC_SHAPE_REFUSED;                                                                     \
        }                                                                                                       \
        hipLaunchKernelGGL((k_smo<TT, RR, TB>), dim3((unsigned)nprob), dim3(TT), dyn, stream, K, ld, diag, dprobs, C, \
                           eps, max_iter);                                                                      \
    } else
    SMO_LAUNCH(256, 4, 1)
    SMO_LAUNCH(512, 4, 1)
    SMO_LAUNCH(256, 8, 1)
    SMO_LAUNCH(512, 8, 1)
    SMO_LAUNCH(1024, 4, 1)
    SMO_LAUNCH(1024, 8, 1)
    SMO_LAUNCH(1024, 10, 1)
    SMO_LAUNCH(1024, 12, 1)
    SMO_LAUNCH(512, 16, 1)
    SMO_LAUNCH(1024, 16, 2)
    {
        g_svm_err = "GKM_SVM_SHAPE: unsupported shape";
        return GKMSVM_RC_SHAPE_REFUSED;
    }
#undef SMO_LAUNCH
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return svm_fail("k_smo", e);
    return 0;
}

constexpr int GEN_MAX_L = 60000; /* one byte of LDS per sample for the shrinking flags */

extern "C" int gkmsvm_train_batch_general(int device, const double *K, int64_t ld, int n, int nprob, const int *idx,
                                          const int64_t *off, const int *n0, double C, double eps, int shrinking,
                                          double *alpha, double *grad, double *rho, int *iters, void *stream_)
{
    if (!K || n <= 0 || ld < n || nprob <= 0 || !idx || !off || !n0 || !alpha || !grad || !rho || !iters) {
        g_svm_err = "gkmsvm_train_batch_general: bad arguments";
        return 2;
    }
    hipStream_t stream = (hipStream_t)stream_;
    SVMCHK(hipSetDevice(device));
    const int64_t total = off[nprob];
    int64_t maxl = 0;
    for (int p = 0; p < nprob; p++) {
        const int64_t l = off[p + 1] - off[p];
        if (l > maxl) maxl = l;
        if (l <= 0 || l > (int64_t)GEN_MAX_L || n0[p] < 0 || n0[p] > l) {
            g_svm_err = "gkmsvm_train_batch_general: a problem has no samples or more than 60000";
            return 3;
        }
    }
    /* scratch: 6 int, 5 double, 2 float arrays of `total` entries */
    const size_t per = 6 * sizeof(int) + 5 * sizeof(double) + 2 * sizeof(float);
    const size_t state_bytes = align256(per * (size_t)total + 64), probs_bytes = align256(sizeof(GenProb) * (size_t)nprob);
    SvmScratch *const scr = scratch_acquire(device, state_bytes + probs_bytes + sizeof(double) * (size_t)n);
    if (!scr) { g_svm_err = "gkmsvm_train_batch_general: device scratch"; return 4; }
    /* (error returns included: whatever this call enqueued may still be writing the buffer -- k_diag below -- and another
     * solver on another stream of the device may take it from the pool next) */
    struct Release { SvmScratch *b; hipStream_t s; ~Release() { (void)hipStreamSynchronize(s); scratch_release(b); } } release{scr, stream};
    char *const scratch = scr->p;
    double *d0 = (double *)scratch;
    int *i0 = (int *)(d0 + 5 * total);
    float *f0 = (float *)(i0 + 6 * total);
    std::vector<GenProb> h((size_t)nprob);
    for (int p = 0; p < nprob; p++) {
        const int64_t o = off[p];
        GenProb &g = h[(size_t)p];
        g.idx = idx + o; g.l = (int)(off[p + 1] - o); g.n0 = n0[p];
        g.alpha_out = alpha + o; g.grad_out = grad + o; g.rho = rho + p; g.iters = iters + p;
        g.alpha = d0 + o; g.G = d0 + total + o; g.Gbar = d0 + 2 * total + o; g.QD = d0 + 3 * total + o; g.fa = d0 + 4 * total + o;
        g.gidx = i0 + o; g.ys = i0 + total + o; g.aset = i0 + 2 * total + o; g.swaps = i0 + 3 * total + o;
        g.fg = i0 + 4 * total + o; g.fy = i0 + 5 * total + o;
        g.Qi = f0 + o; g.Qj = f0 + total + o;
    }
    GenProb *const dprobs = (GenProb *)(scratch + state_bytes);
    double *const diag = (double *)(scratch + state_bytes + probs_bytes);
    hipError_t e = hipMemcpyAsync(dprobs, h.data(), sizeof(GenProb) * (size_t)nprob, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream); /* h is a host temporary */
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_diag, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, K, ld, n, diag);
        int max_iter = 10000000;
        if (const char *mi = getenv("GKM_SVM_MAX_ITER")) if (atoi(mi) > 0) max_iter = atoi(mi);
        const int cap = (int)(((size_t)maxl + 15) & ~(size_t)15);
        /* 512 threads for folds of at most 8 192 samples (GKM_SVM_GEN_T=512|1024 overrides): half the waves per barrier
         * and per block-wide selection, sixteen positions per thread -- and the scanned state in LDS (GKM_SVM_GEN_LDS=0:
         * in global memory as for larger folds, for timing) */
        int gt = maxl <= GEN_LDS_L ? 512 : 1024;
        if (const char *g = getenv("GKM_SVM_GEN_T")) gt = atoi(g) == 512 ? 512 : 1024;
        bool lds_state = gt == 512 && maxl <= GEN_LDS_L;
        if (const char *g = getenv("GKM_SVM_GEN_LDS")) lds_state = lds_state && atoi(g) != 0;
        size_t dyn = lds_state ? (size_t)18 * cap : (size_t)cap;
        const void *fn = lds_state ? (const void *)k_smo_general<512, true>
                         : gt == 512 ? (const void *)k_smo_general<512, false> : (const void *)k_smo_general<1024, false>;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        if (e != hipSuccess && lds_state) { /* a device that refuses 147 KB of LDS: the state stays in global memory */
            (void)hipGetLastError();
            lds_state = false;
            dyn = (size_t)cap;
            fn = (const void *)k_smo_general<512, false>;
            e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        }
        if (e == hipSuccess) {
            if (lds_state)
                hipLaunchKernelGGL((k_smo_general<512, true>), dim3((unsigned)nprob), dim3(512), dyn, stream, K, ld, diag, dprobs, C,
                                   eps, max_iter, shrinking ? 1 : 0, cap);
            else if (gt == 512)
                hipLaunchKernelGGL((k_smo_general<512, false>), dim3((unsigned)nprob), dim3(512), dyn, stream, K, ld, diag, dprobs, C,
                                   eps, max_iter, shrinking ? 1 : 0, cap);
            else
                hipLaunchKernelGGL((k_smo_general<1024, false>), dim3((unsigned)nprob), dim3(1024), dyn, stream, K, ld, diag, dprobs,
                                   C, eps, max_iter, shrinking ? 1 : 0, cap);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
    }
    if (e != hipSuccess) return svm_fail("k_smo_general", e);
    return 0;
}

extern "C" int gkmsvm_decision_batch(int device, const double *K, int64_t ld, int nprob, const int *idx,
                                     const int64_t *off, const int *n0, const double *alpha, const double *rho,
                                     const int *test_idx, const int64_t *test_off, double *dec, void *stream_)
{
    if (!K || nprob <= 0 || !idx || !off || !n0 || !alpha || !rho || !test_idx || !test_off || !dec) {
        g_svm_err = "gkmsvm_decision_batch: bad arguments";
        return 2;
    }
    hipStream_t stream = (hipStream_t)stream_;
    SVMCHK(hipSetDevice(device));
    std::vector<DecProb> h((size_t)nprob);
    int maxtest = 0;
    for (int p = 0; p < nprob; p++) {
        const int nt = (int)(test_off[p + 1] - test_off[p]);
        h[(size_t)p] = {idx + off[p], (int)(off[p + 1] - off[p]), n0[p], alpha + off[p], rho + p,
                        test_idx + test_off[p], nt, dec + test_off[p]};
        if (nt > maxtest) maxtest = nt;
    }
    if (maxtest == 0) return 0;
    SvmScratch *const scr = scratch_acquire(device, sizeof(DecProb) * (size_t)nprob);
    if (!scr) { g_svm_err = "gkmsvm_decision_batch: device scratch"; return 4; }
    /* (error returns included: whatever this call enqueued may still be writing the buffer -- k_diag below -- and another
     * solver on another stream of the device may take it from the pool next) */
    struct Release { SvmScratch *b; hipStream_t s; ~Release() { (void)hipStreamSynchronize(s); scratch_release(b); } } release{scr, stream};
    DecProb *const dprobs = (DecProb *)scr->p;
    SVMCHK(hipMemcpyAsync(dprobs, h.data(), sizeof(DecProb) * (size_t)nprob, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_decision, dim3((unsigned)((maxtest + 127) / 128), (unsigned)nprob), dim3(128), 0, stream, K, ld,
                       dprobs);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return svm_fail("k_decision", e);
    return 0;
}
