/*
 * bitslice_cpu_probe.cpp -- runs the per-lane bit-sliced program of gkm_bitslice.h on
 * the CPU for one (row sequence, column sequence) pair.  UNIT-TEST HARNESS for the
 * product's kernel logic (tests/test_bitslice_core.py); it is not linked into
 * gkmkern_pylib.so and is not a fallback path.
 */
#include <stdint.h>
#include <stdlib.h>
#include <vector>

#include "gkm_bitslice.h"

using namespace gkmbs;

template <int W, int L, int D>
static void run_pair(const uint8_t *A, int lenA, const uint8_t *B, int lenB, const uint8_t *wd,
                     uint32_t *acc /* [1<<NB] */)
{
    constexpr int NB = planes_for(D);
    constexpr int CAP = segment_capacity(W, L);
    const int nA = lenA - L + 1, nB = lenB - L + 1, T = lenB;
    for (int k = 0; k <= D; k++) acc[k] = 0;
    (void)NB;
    std::vector<uint32_t> sb[2][3];
    for (int st = 0; st < 2; st++)
        for (int pl = 0; pl < 3; pl++) {
            sb[st][pl].resize((size_t)T + W);
            for (int x = 0; x < T + W; x++) sb[st][pl][(size_t)x] = sb_word(B, T, st, x, W, L, pl);
        }
    const uint32_t rcpT = mod_magic((uint32_t)T);
    auto weight = [&](int n, int p) { return wd ? (uint32_t)wd[n / 2 > p ? n / 2 - p : p - n / 2] : 1u; };
    std::vector<uint32_t> lmA((size_t)nA), lmB[2];
    for (int p = 0; p < nA; p++) lmA[(size_t)p] = lmer_entry(A, lenA, L, 0, p, weight(nA, p));
    for (int st = 0; st < 2; st++) {
        lmB[st].resize((size_t)nB);
        for (int q = 0; q < nB; q++) lmB[st][(size_t)q] = lmer_entry(B, lenB, L, st, q, weight(nB, st ? nB - 1 - q : q));
    }
    for (int s0 = 0; s0 < nA; s0 += CAP) {
        uint32_t Ahi[W], Alo[W], AV[W];
        for (int w = 0; w < W; w++) {
            Ahi[w] = row_plane_word(A, lenA, s0, w, W, L, 0);
            Alo[w] = row_plane_word(A, lenA, s0, w, W, L, 1);
            AV[w] = row_plane_word(A, lenA, s0, w, W, L, 2);
        }
        auto rl = [&](int i0) { return lmA[(size_t)(s0 + i0)]; };
        auto cl = [&](int st, int q) { return lmB[st][(size_t)q]; };
        for (int st = 0; st < 2; st++)
            for (int delta = 0; delta < T; delta++) {
                uint32_t hit[W];
                /* odd shifts exercise the variant that leaves the column-side validity to resolve_hit */
                window_hits<W, L, D>(Ahi, Alo, AV, &sb[st][0][(size_t)delta], &sb[st][1][(size_t)delta],
                                     (delta & 1) ? (const uint32_t *)nullptr : &sb[st][2][(size_t)delta], hit);
                for (int w = 0; w < W; w++) {
                    uint32_t h = hit[w];
                    while (h) {
                        const int bit = __builtin_ctz(h);
                        h &= h - 1u;
                        const HitValue hv = resolve_hit<W>(bit, w, delta, st, (uint32_t)T, rcpT, nB, rl, cl);
                        acc[hv.m] += hv.v;
                    }
                }
            }
    }
}

#define CASE(WW, LL, DD) \
    if (W == WW && L == LL && d == DD) { run_pair<WW, LL, DD>(A, lenA, B, lenB, wd, acc); ok = 1; }

/* wd: distance-indexed positional weight table (NULL = unweighted) */
extern "C" int bsprobe_profile(int W, int L, int d, const uint8_t *A, int lenA, const uint8_t *B, int lenB,
                               const uint8_t *wd, int32_t *P)
{
    uint32_t acc[16] = {0};
    int ok = 0;
    CASE(10, 11, 3) CASE(10, 10, 3) CASE(10, 12, 4) CASE(10, 8, 4) CASE(10, 9, 4) CASE(10, 12, 6)
    CASE(10, 4, 2) CASE(10, 2, 1) CASE(10, 3, 0) CASE(10, 5, 2) CASE(10, 6, 3) CASE(10, 7, 3)
    CASE(5, 11, 3) CASE(16, 11, 3) CASE(3, 12, 4) CASE(10, 12, 8) CASE(10, 12, 12) CASE(10, 11, 1)
    if (!ok) return 1;
    for (int m = 0; m <= d; m++) P[m] = (int32_t)acc[m];
    return 0;
}
