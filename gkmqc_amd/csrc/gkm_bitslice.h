/*
 * gkm_bitslice.h -- the bit-sliced "diagonal" formulation of the gkm mismatch profile.
 *
 * What the reference computes with a k-mer tree DFS (src/libgkm.c:315-387) is, for a
 * pair of sequences (a, b), the histogram over all l-mer pairs (p, q) of their Hamming
 * distance m <= d, weighted by w_a[p] * w_b[q].
 *
 * Observation: mm(p, q) = sum_{i<L} [a[p+i] != b[q+i]] is a sliding-window sum along a
 * DIAGONAL (q - p = const) of the base-level mismatch matrix.  So instead of comparing
 * l-mers one by one (>= 5 integer ops per comparison) we
 *   1. hold a row SEGMENT of up to 32*W bases as two bit planes (hi/lo bit of the 2-bit
 *      base code) in a STRIDED layout: base i of the segment is bit (i / W) of word (i % W);
 *   2. for a cyclic shift `delta` of the column sequence b (period T = len(b)) take the
 *      same strided layout of b[(i + delta) mod T]  -- these words are precomputed once per
 *      column sequence ("SB table"), are uniform across a wavefront and live in SGPRs;
 *   3. mismatch bits Z[w] = (Ahi[w]^Bhi[w]) | (Alo[w]^Blo[w]): 32 base comparisons in 3 ops;
 *   4. in the strided layout "next base" = "next word", so the L-window sum over bases
 *      i..i+L-1 is a sum over L CONSECUTIVE WORDS (no funnel shifts); it is evaluated
 *      bit-sliced (one bit plane per binary digit of the count, 32 windows per op) by a
 *      sliding power-of-two tree  w2[x]=Z[x]+Z[x+1], w4[x]=w2[x]+w2[x+2], w8[x]=w4[x]+w4[x+4];
 *   5. counts are kept in NB planes + a sticky overflow plane; windows with count <= d and
 *      a valid start on both sides are the HITS (about 0.1-0.4 % of all windows on iid
 *      sequences), the only places where weights are touched.
 * Cost: ~1.1-1.4 integer VALU ops per l-mer comparison instead of ~5-6.
 *
 * Everything here is plain C++ on uint32_t so that the very same code runs per lane on
 * the GPU and in the CPU unit test (tests/test_bitslice_core.py via csrc/bitslice_cpu_probe.cpp).
 */
#ifndef GKM_BITSLICE_H
#define GKM_BITSLICE_H

#include <stdint.h>

#if defined(__HIPCC__)
#define GKM_HD __host__ __device__ __forceinline__
#else
#define GKM_HD inline
#endif

namespace gkmbs {

/* number of count planes needed to represent 0..d exactly */
constexpr int planes_for(int d) { return d <= 1 ? 1 : d <= 3 ? 2 : d <= 7 ? 3 : 4; }

/* window starts a full row segment of W words contributes (its last L-1 bases only
 * complete windows of the segment's own starts; the next segment begins here) */
constexpr int segment_capacity(int W, int L) { return 32 * W - (L - 1); }

template <int NB>
struct Count {
    uint32_t b[NB]; /* binary digits of the per-bit-position count */
    uint32_t ovf;   /* sticky: count exceeded 2^NB - 1 somewhere on the way */
};

template <int NB>
GKM_HD Count<NB> count_from_bit(uint32_t z)
{
    Count<NB> r;
    r.b[0] = z;
#pragma unroll
    for (int i = 1; i < NB; i++) r.b[i] = 0u;
    r.ovf = 0u;
    return r;
}

/* ripple-carry add of two bit-sliced counts (+ optional 1-bit carry-in plane).
 * sum = x ^ y ^ c ; carry = majority(x, y, c) = (t & c) | (~t & x) with t = x ^ y
 * (one v_bfi_b32).  Planes known to be zero fold away after full unrolling. */
template <int NB>
GKM_HD Count<NB> count_add(const Count<NB> &x, const Count<NB> &y, uint32_t cin = 0u)
{
    Count<NB> r;
    uint32_t c = cin;
#pragma unroll
    for (int i = 0; i < NB; i++) {
        const uint32_t t = x.b[i] ^ y.b[i];
        r.b[i] = t ^ c;
        c = (t & c) | (~t & x.b[i]);
    }
    r.ovf = x.ovf | y.ovf | c;
    return r;
}

/* bit mask of positions whose count is <= D (D compile-time, 0 <= D < 2^NB) */
template <int NB, int D>
GKM_HD uint32_t count_le(const Count<NB> &v)
{
    uint32_t less = 0u, eq = ~0u;
#pragma unroll
    for (int i = NB - 1; i >= 0; i--) {
        if ((D >> i) & 1) {
            less |= eq & ~v.b[i];
            eq &= v.b[i];
        } else {
            eq &= ~v.b[i];
        }
    }
    return (less | eq) & ~v.ovf;
}

struct Parts {
    int n;
    int size[4];
    int off[4];
};
constexpr Parts make_parts(int L)
{
    Parts p{};
    int o = 0;
    for (int sz = 8; sz >= 1; sz >>= 1)
        if (L & sz) {
            p.size[p.n] = sz;
            p.off[p.n] = o;
            o += sz;
            p.n++;
        }
    return p;
}

/*
 * One shift `delta` of one column strand against one row segment.
 *   Ahi/Alo : the segment's planes, W words each (per lane)
 *   Bhi/Blo : SB words x = delta .. delta+W-1 of the column strand (uniform)
 * Produces for w in [0,W): cnt[w] = per-bit mismatch count of the L-window starting at
 * segment base b*W + w, saturating (ovf) above 2^NB-1.
 */
template <int W, int L, int NB>
GKM_HD void window_counts(const uint32_t *Ahi, const uint32_t *Alo, const uint32_t *Bhi,
                          const uint32_t *Blo, Count<NB> *cnt)
{
    constexpr int NX = W + L - 1;
    uint32_t Z[NX];
#pragma unroll
    for (int w = 0; w < W; w++) Z[w] = (Ahi[w] ^ Bhi[w]) | (Alo[w] ^ Blo[w]);
#pragma unroll
    for (int x = W; x < NX; x++) Z[x] = Z[x - W] >> 1; /* word x == word x-W one bit up */

    /* sliding power-of-two window sums; unused levels are dead code */
    Count<NB> s1[NX], s2[NX], s4[NX], s8[NX];
#pragma unroll
    for (int x = 0; x < NX; x++) s1[x] = count_from_bit<NB>(Z[x]);
#pragma unroll
    for (int x = 0; x + 1 < NX; x++) s2[x] = count_add<NB>(s1[x], s1[x + 1]);
#pragma unroll
    for (int x = 0; x + 3 < NX; x++) s4[x] = count_add<NB>(s2[x], s2[x + 2]);
#pragma unroll
    for (int x = 0; x + 7 < NX; x++) s8[x] = count_add<NB>(s4[x], s4[x + 4]);

    constexpr Parts P = make_parts(L);
#pragma unroll
    for (int w = 0; w < W; w++) {
        /* L = 8*e8 + 4*e4 + 2*e2 + e1: chain the blocks left to right; a trailing
         * single base rides as the carry-in of the last two-operand add */
        Count<NB> part[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            part[k] = count_from_bit<NB>(0u);
            if (k < P.n) {
                const int x = w + P.off[k];
                if (P.size[k] == 8) part[k] = s8[x];
                else if (P.size[k] == 4) part[k] = s4[x];
                else if (P.size[k] == 2) part[k] = s2[x];
                else part[k] = s1[x];
            }
        }
        Count<NB> acc = part[0];
        if (P.n == 2) {
            acc = count_add<NB>(acc, part[1]);
        } else if (P.n == 3) {
            if (P.size[2] == 1) acc = count_add<NB>(acc, part[1], Z[w + P.off[2]]);
            else acc = count_add<NB>(count_add<NB>(acc, part[1]), part[2]);
        } else if (P.n == 4) {
            acc = count_add<NB>(count_add<NB>(acc, part[1]), part[2], Z[w + P.off[3]]);
        }
        cnt[w] = acc;
    }
}

/* ------------------------------------------------------------------ tables */
/* Word w of a ROW SEGMENT plane.  Segment base i = b*W + w is sequence position s0 + i.
 * plane 0/1: hi/lo bit of the base code (0 beyond the end of the sequence);
 * plane 2: window-start validity: the l-mer starting at s0+i exists (s0+i <= len-L) and
 * belongs to this segment (i < segment_capacity). */
GKM_HD uint32_t row_plane_word(const uint8_t *codes, int len, int s0, int w, int W, int L, int plane)
{
    uint32_t v = 0u;
    const int cap = 32 * W - (L - 1);
    for (int b = 0; b < 32; b++) {
        const int i = b * W + w, pos = s0 + i;
        uint32_t bit;
        if (plane == 2) bit = (pos <= len - L && i < cap) ? 1u : 0u;
        else bit = (pos < len) ? ((uint32_t)(codes[pos] >> (1 - plane)) & 1u) : 0u;
        v |= bit << b;
    }
    return v;
}

/* Word x of a COLUMN STRAND table ("SB"): bit b describes strand base (b*W + x) mod T,
 * i.e. what segment base i = b*W + w meets under the cyclic shift delta = x - w.
 * strand 0 = the sequence, strand 1 = its reverse complement (libgkm.c:877-888).
 * plane 2: that base starts an l-mer that does not wrap (q <= T-L). */
GKM_HD uint32_t sb_word(const uint8_t *codes, int T, int strand, int x, int W, int L, int plane)
{
    uint32_t v = 0u;
    int q = x % T;
    const int step = W % T;
    for (int b = 0; b < 32; b++) {
        uint32_t bit;
        if (plane == 2) bit = (q <= T - L) ? 1u : 0u;
        else {
            const uint32_t code = strand ? (3u - codes[T - 1 - q]) : codes[q];
            bit = (code >> (1 - plane)) & 1u;
        }
        v |= bit << b;
        q += step;
        if (q >= T) q -= T;
    }
    return v;
}

/* ------------------------------------------------------------------- hits */
/* Consume the hit bits of one word: for each set bit b of h the l-mer pair
 *   row window start  p = s0 + b*W + w           (weight wtA[p])
 *   column l-mer      q = (b*W + w + delta) mod T on `strand`
 *                     (weight wtB[q] forward, wtB[nB-1-q] reverse: libgkm.c:924)
 * has m = count bits (c[0..NB-1] at bit b) mismatches, m <= d.  acc[m] += wa*wb in
 * wrapping 32-bit arithmetic (the reference's int, libgkm.c:338).
 * wtA/wtB may be NULL for unweighted kernels (all weights 1). */
template <int W, int NB>
GKM_HD void consume_hits(uint32_t h, const uint32_t *c, int delta, int w, int strand, int s0, int T,
                         int nB, const uint8_t *wtA, const uint8_t *wtB, uint32_t *acc)
{
    while (h) {
        const int b = __builtin_ctz(h);
        h &= h - 1u;
        int m = 0;
#pragma unroll
        for (int i = 0; i < NB; i++) m |= (int)((c[i] >> b) & 1u) << i;
        const int i0 = b * W + w;
        const int q = (i0 + delta) % T;
        uint32_t v = 1u;
        if (wtA) v = (uint32_t)wtA[s0 + i0] * (uint32_t)wtB[strand ? (nB - 1 - q) : q];
#pragma unroll
        for (int k = 0; k < (1 << NB); k++) acc[k] += (k == m) ? v : 0u;
    }
}

/* queue entry meta word: delta | w << 11 | strand << 17 */
GKM_HD uint32_t pack_meta(int delta, int w, int strand)
{
    return (uint32_t)delta | ((uint32_t)w << 11) | ((uint32_t)strand << 17);
}

} /* namespace gkmbs */
#endif
