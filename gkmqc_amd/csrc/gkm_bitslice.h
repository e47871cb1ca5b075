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

/* ------------------------------------------------------------------ lop3 */
/* Any 3-input bitwise function in one full-rate VALU op: gfx950's v_bitop3_b32.  TT is the
 * truth table f(0xF0, 0xCC, 0xAA).  (v_and_or_b32 / v_or3_b32 issue at HALF rate on MI355X,
 * v_bitop3_b32 at full rate -- tools/valu_peak.hip -- so every 3-input op below is a lop3.) */
template <int TT>
GKM_HD uint32_t lop3(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
#else
    uint32_t r = 0u;
    for (int k = 0; k < 8; k++)
        if ((TT >> k) & 1) r |= ((k & 4) ? a : ~a) & ((k & 2) ? b : ~b) & ((k & 1) ? c : ~c);
    return r;
#endif
}
constexpr int TT_XOR3 = 0x96;     /* a ^ b ^ c */
constexpr int TT_MAJ = 0xE8;      /* majority(a, b, c) */
constexpr int TT_OR3 = 0xFE;      /* a | b | c */
constexpr int TT_A_OR_BXC = 0xF6; /* a | (b ^ c) */
constexpr int TT_NA_B_C = 0x08;   /* ~a & b & c */

constexpr int bitlen(int v)
{
    int n = 0;
    while (v) { n++; v >>= 1; }
    return n;
}

/* Bit-sliced count whose value is known (at compile time) to lie in 0..MX.  Only the planes
 * that can be non-zero exist; `ovf` exists (is meaningful) iff MX does not fit NB planes. */
template <int NB, int MX>
struct Cnt {
    static constexpr int P = bitlen(MX) < NB ? bitlen(MX) : NB;
    static constexpr bool OV = MX >= (1 << NB);
    uint32_t b[P > 0 ? P : 1];
    uint32_t ovf;
};

template <int PX, int PY, bool CIN>
constexpr bool carry_into(int i)
{
    return i == 0 ? CIN : (((i - 1 < PX) ? 1 : 0) + ((i - 1 < PY) ? 1 : 0) + (carry_into<PX, PY, CIN>(i - 1) ? 1 : 0)) >= 2;
}

template <int I, int NB, int M1, int M2, bool CIN>
GKM_HD void add_plane(const Cnt<NB, M1> &x, const Cnt<NB, M2> &y, uint32_t &c, Cnt<NB, M1 + M2 + (CIN ? 1 : 0)> &r)
{
    using X = Cnt<NB, M1>;
    using Y = Cnt<NB, M2>;
    using R = Cnt<NB, M1 + M2 + (CIN ? 1 : 0)>;
    if constexpr (I < R::P) {
        constexpr bool hx = I < X::P, hy = I < Y::P, hc = carry_into<X::P, Y::P, CIN>(I);
        constexpr int nin = (hx ? 1 : 0) + (hy ? 1 : 0) + (hc ? 1 : 0);
        /* the carry out of this plane is wanted by the next plane, or by ovf if this is plane NB-1 */
        constexpr bool want_c = nin >= 2 && ((I + 1 < R::P) || (I + 1 == NB && R::OV));
        if constexpr (nin == 3) {
            const uint32_t xi = x.b[I], yi = y.b[I], ci = c;
            r.b[I] = lop3<TT_XOR3>(xi, yi, ci);
            if constexpr (want_c) c = lop3<TT_MAJ>(xi, yi, ci);
        } else if constexpr (nin == 2) {
            const uint32_t u = hx ? x.b[hx ? I : 0] : y.b[hy ? I : 0];
            const uint32_t v = hc ? c : y.b[hy ? I : 0];
            r.b[I] = u ^ v;
            if constexpr (want_c) c = u & v;
        } else if constexpr (nin == 1) {
            r.b[I] = hx ? x.b[hx ? I : 0] : (hy ? y.b[hy ? I : 0] : c);
        } else {
            r.b[I] = 0u;
        }
    }
}

/* x + y (+ a one-bit carry-in plane): 2 ops per full-adder plane, 2 per half-adder plane */
template <bool CIN, int NB, int M1, int M2>
GKM_HD Cnt<NB, M1 + M2 + (CIN ? 1 : 0)> cnt_add(const Cnt<NB, M1> &x, const Cnt<NB, M2> &y, uint32_t cin = 0u)
{
    using X = Cnt<NB, M1>;
    using Y = Cnt<NB, M2>;
    using R = Cnt<NB, M1 + M2 + (CIN ? 1 : 0)>;
    R r;
    uint32_t c = cin;
    add_plane<0, NB, M1, M2, CIN>(x, y, c, r);
    add_plane<1, NB, M1, M2, CIN>(x, y, c, r);
    add_plane<2, NB, M1, M2, CIN>(x, y, c, r);
    add_plane<3, NB, M1, M2, CIN>(x, y, c, r);
    r.ovf = 0u;
    if constexpr (R::OV) {
        /* carry out of plane NB-1 exists iff that plane had >= 2 inputs */
        constexpr bool hc = R::P == NB && (((NB - 1 < X::P) ? 1 : 0) + ((NB - 1 < Y::P) ? 1 : 0) +
                                           (carry_into<X::P, Y::P, CIN>(NB - 1) ? 1 : 0)) >= 2;
        constexpr int terms = (X::OV ? 1 : 0) + (Y::OV ? 1 : 0) + (hc ? 1 : 0);
        if constexpr (terms == 3) r.ovf = lop3<TT_OR3>(x.ovf, y.ovf, c);
        else if constexpr (terms == 2) r.ovf = X::OV ? (x.ovf | (Y::OV ? y.ovf : c)) : (y.ovf | c);
        else if constexpr (terms == 1) r.ovf = X::OV ? x.ovf : (Y::OV ? y.ovf : c);
    }
    return r;
}

/* truth table of [value(b2 b1 b0) > D] for a lop3 over three count planes */
constexpr int tt_gt3(int D)
{
    int tt = 0;
    for (int k = 0; k < 8; k++)
        if (k > D) tt |= 1 << k;
    return tt;
}
/* truth table of [ovf | value(b1 b0) > D] with inputs (b1, b0, ovf) */
constexpr int tt_gt2_or(int D)
{
    int tt = 0;
    for (int k = 0; k < 8; k++)
        if ((k & 1) || ((k >> 1) > D)) tt |= 1 << k;
    return tt;
}

/* mask of positions where the count is NOT <= D (too many mismatches or overflowed) */
template <int D, int NB, int MX>
GKM_HD uint32_t cnt_exceeds(const Cnt<NB, MX> &v)
{
    using V = Cnt<NB, MX>;
    if constexpr (MX <= D) {
        return 0u;
    } else if constexpr (V::P <= 2) {
        const uint32_t b1 = V::P > 1 ? v.b[V::P > 1 ? 1 : 0] : 0u, b0 = V::P > 0 ? v.b[0] : 0u;
        if constexpr (D >= 3) return V::OV ? v.ovf : 0u;
        else return lop3<tt_gt2_or(D)>(b1, b0, V::OV ? v.ovf : 0u);
    } else if constexpr (V::P == 3) {
        if constexpr (D >= 7) return V::OV ? v.ovf : 0u;
        else {
            const uint32_t gt = lop3<tt_gt3(D)>(v.b[2], v.b[1], v.b[0]);
            return V::OV ? (gt | v.ovf) : gt;
        }
    } else if constexpr (((D + 1) & D) == 0 && D < 8) {
        /* D + 1 a power of two: "count > D" is the OR of the planes from log2(D+1) up */
        uint32_t gt;
        if constexpr (D == 7) gt = v.b[3];
        else if constexpr (D == 3) gt = v.b[3] | v.b[2];
        else if constexpr (D == 1) gt = lop3<TT_OR3>(v.b[3], v.b[2], v.b[1]);
        else gt = lop3<TT_OR3>(v.b[3], v.b[2], v.b[1]) | v.b[0];
        return V::OV ? (gt | v.ovf) : gt;
    } else {
        uint32_t gt;
        if constexpr (D >= 15) gt = 0u;
        else if constexpr (D >= 8) gt = v.b[3] & lop3<tt_gt3(D - 8)>(v.b[2], v.b[1], v.b[0]);
        else gt = v.b[3] | lop3<tt_gt3(D)>(v.b[2], v.b[1], v.b[0]);
        return V::OV ? (gt | v.ovf) : gt;
    }
}

constexpr int TT_A_AND_BXC = 0x60; /* a & (b ^ c) */
constexpr int TT_BXC_AND_AXC = 0x42; /* (b ^ c) & (a ^ c) */

/* Sum of N one-bit planes as an exact bit-sliced count, used once per shift for the first
 * window.  Column compression: the N planes of weight 1 are folded three at a time by full
 * adders (x ^ y ^ z stays in the column, maj(x, y, z) moves to the next), a half adder takes a
 * leftover pair; the N/2 carries are the next column.  A full adder takes one plane off the
 * total, so this is the minimum number of them: L = 11 -> 7 full adders + 1 half adder = 16
 * ops (the balanced tree of two-operand additions it replaces took 22). */
template <int N>
struct ColumnSum {
    static constexpr int NC = N / 2; /* carries out of a column of N planes */
    static GKM_HD uint32_t run(const uint32_t *x, uint32_t *carry)
    {
        uint32_t s = x[0];
        int nc = 0;
#pragma unroll
        for (int i = 1; i + 1 < N; i += 2) {
            carry[nc++] = lop3<TT_MAJ>(s, x[i], x[i + 1]);
            s = lop3<TT_XOR3>(s, x[i], x[i + 1]);
        }
        if constexpr (N % 2 == 0) {
            carry[nc++] = s & x[N - 1];
            s ^= x[N - 1];
        }
        return s;
    }
};

template <int I, int NB, int MX, int N>
GKM_HD void plane_sum_columns(const uint32_t *x, Cnt<NB, MX> &r)
{
    if constexpr (N >= 1 && I < Cnt<NB, MX>::P) {
        uint32_t carry[ColumnSum<N>::NC > 0 ? ColumnSum<N>::NC : 1];
        r.b[I] = ColumnSum<N>::run(x, carry);
        plane_sum_columns<I + 1, NB, MX, ColumnSum<N>::NC>(carry, r);
    }
}

template <int NB, int N>
struct PlaneSum {
    static_assert(bitlen(N) <= NB, "PlaneSum: the count must fit its planes");
    static GKM_HD Cnt<NB, N> run(const uint32_t *z)
    {
        Cnt<NB, N> r;
        r.ovf = 0u;
        plane_sum_columns<0, NB, N, N>(z, r);
        return r;
    }
};

/*
 * One cyclic shift of one column strand against one row segment.
 *   Ahi/Alo/AV : the segment's planes, W words each (per lane)
 *   Bhi/Blo/Bv : SB words x = delta .. delta+W-1 of the column strand (wave-uniform)
 * For w in [0,W): hit[w] = windows (bit b <-> segment base b*W + w) with <= D mismatches
 * whose start is valid on both sides.  With Bv == nullptr the column-side validity (the
 * window must not wrap around the end of the strand) is NOT applied here: the ~L/T of
 * windows concerned then yield a few candidate hits more, which resolve_hit() rejects --
 * one operand and one scalar-operand instruction less per word in the hot loop.
 * The exact count of a hit is recomputed from the l-mer tables when the hit is consumed
 * (cheaper than carrying the count planes along).
 *
 * Window counts: the first window (w = 0) is summed with an adder tree; every further window
 * differs from its left neighbour by one base entering and one leaving, so the exact
 * bit-sliced count is stepped by an up/down counter, 2 ops per count plane (1 for the last):
 *     t_0 = Zin ^ Zout;  b_i ^= t_i;  t_{i+1} = t_i & (b_i_old ^ Zout)
 * (carry when counting up through a 1, borrow when counting down through a 0).
 */
template <int W, int L, int D>
GKM_HD void window_hits(const uint32_t *Ahi, const uint32_t *Alo, const uint32_t *AV, const uint32_t *Bhi,
                        const uint32_t *Blo, const uint32_t *Bv, uint32_t *hit)
{
    static_assert(L >= 2 && L <= 12, "L out of range");
    constexpr int P = bitlen(L); /* planes of an exact count 0..L */
    constexpr int NX = W + L - 1;
    uint32_t Z[NX];
#pragma unroll
    for (int w = 0; w < W; w++) Z[w] = lop3<TT_A_OR_BXC>(Ahi[w] ^ Bhi[w], Alo[w], Blo[w]);
#pragma unroll
    for (int x = W; x < NX; x++) Z[x] = Z[x - W] >> 1; /* word x == word x-W one bit up */

    Cnt<P, L> cnt = PlaneSum<P, L>::run(Z);
#pragma unroll
    for (int w = 0; w < W; w++) {
        if (w > 0) {
            const uint32_t zout = Z[w - 1], zin = Z[w + L - 1];
            /* plane 0 straight from (b_0, Zin, Zout): b_0 ^ Zin ^ Zout and the carry/borrow
             * (Zin ^ Zout) & (b_0 ^ Zout), one lop3 each -- no separate Zin ^ Zout */
            const uint32_t old0 = cnt.b[0];
            cnt.b[0] = lop3<TT_XOR3>(old0, zin, zout);
            uint32_t t = lop3<TT_BXC_AND_AXC>(old0, zin, zout);
#pragma unroll
            for (int i = 1; i < P; i++) {
                const uint32_t old = cnt.b[i];
                cnt.b[i] = old ^ t;
                if (i + 1 < P) t = lop3<TT_A_AND_BXC>(t, old, zout);
            }
        }
        hit[w] = Bv ? lop3<TT_NA_B_C>(cnt_exceeds<D>(cnt), AV[w], Bv[w]) : (~cnt_exceeds<D>(cnt) & AV[w]);
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

/* Packed lanes (gkm_pack.h): the bit that one PIECE -- bit rows [b0, b0+nb) of a lane holding
 * sequence positions p0.. with cnt owned window starts -- contributes to bit row b, word w of
 * plane 0/1 (hi/lo bit of the base code, 0 beyond the sequence) or plane 2 (window start owned
 * by the piece).  Bit rows outside the piece contribute 0. */
GKM_HD uint32_t piece_bit(const uint8_t *codes, int len, int b0, int nb, int p0, int cnt, int b, int w, int W,
                          int plane)
{
    if (b < b0 || b >= b0 + nb) return 0u;
    const int li = (b - b0) * W + w;
    if (plane == 2) return li < cnt ? 1u : 0u;
    const int pos = p0 + li;
    return pos < len ? ((uint32_t)(codes[pos] >> (1 - plane)) & 1u) : 0u;
}
/* piece k of a lane owns bit row b: k = (number of piece starts at or below b) - 1, from the
 * lane's mask of piece-start bit rows */
GKM_HD int piece_of_bitrow(uint32_t start_mask, int b)
{
    return __builtin_popcount(start_mask & (0xFFFFFFFFu >> (31 - b))) - 1;
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
/* exact x mod T for x < 2^32 / T, with rcp = ceil(2^32 / T) (T >= 2) */
GKM_HD uint32_t mod_magic(uint32_t T) { return 0xFFFFFFFFu / T + 1u; }
GKM_HD uint32_t mod_small(uint32_t x, uint32_t T, uint32_t rcp)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return x - __umulhi(x, rcp) * T;
#else
    return x - (uint32_t)(((uint64_t)x * rcp) >> 32) * T;
#endif
}

/* Positional weights as a function of the distance to the sequence's centre l-mer:
 * w(n, p) = wd[|n/2 - p|] (libgkm.c:912-925 depends on n and p only through that
 * distance), so ONE small table serves every sequence length. */
GKM_HD uint32_t dist_weight(const uint8_t *wd, int center, int p)
{
    const int dd = center - p;
    return wd[dd < 0 ? -dd : dd];
}

/* l-mer table entry: the L bases as 2 bits each (first base in the highest pair, as a
 * number in base 4) in bits 0..23, the positional weight of that l-mer in bits 24..31 */
GKM_HD uint32_t lmer_entry(const uint8_t *codes, int len, int L, int strand, int p, uint32_t weight)
{
    uint32_t v = 0u;
    for (int i = 0; i < L; i++) {
        const uint32_t c = strand ? (3u - codes[len - 1 - (p + i)]) : codes[p + i];
        v = (v << 2) | c;
    }
    return v | (weight << 24);
}
GKM_HD int lmer_mismatch(uint32_t x, uint32_t y)
{
    uint32_t t = (x ^ y) & 0x00FFFFFFu;
    t = (t | (t >> 1)) & 0x00555555u;
    return __builtin_popcount(t);
}

/* Resolve one hit: bit b of the hit word of (delta, w, strand) against a row segment.
 *   row window start  p = s0 + b*W + w,  column l-mer q = (b*W + w + delta) mod T on `strand`
 * Returns m = Hamming distance of the two l-mers (<= d for a true hit) and v = wa*wb, the
 * amount the reference adds to mmprofile[m] (libgkm.c:338); the weights ride in the top
 * byte of the l-mer table entries (forward: wt[q], reverse strand: wt_rc[q] = wt[n-1-q],
 * libgkm.c:924).
 *   rowlm(i0)        table entry of the row's l-mer at segment offset i0
 *   collm(strand, q) table entry of the column strand's l-mer q */
struct HitValue {
    int m;
    uint32_t v;
};
template <int W, class RowLm, class ColLm>
GKM_HD HitValue resolve_hit(int b, int w, int delta, int strand, uint32_t T, uint32_t rcpT, int nB, RowLm rowlm,
                            ColLm collm)
{
    HitValue r;
    const int i0 = b * W + w;
    const uint32_t x = (uint32_t)(i0 + delta);
    int q;
    if (T >= (uint32_t)(32 * W)) { /* x < 2T: one conditional subtraction (uniform test) */
        const uint32_t y = x - T;
        q = (int)(y < x ? y : x);
    } else {
        q = (int)mod_small(x, T, rcpT);
    }
    if (q >= nB) { /* window wraps around the end of the strand: not an l-mer (see window_hits) */
        r.m = 0;
        r.v = 0u;
        return r;
    }
    const uint32_t ea = rowlm(i0), eb = collm(strand, q);
    r.m = lmer_mismatch(ea, eb);
    r.v = (ea >> 24) * (eb >> 24);
    return r;
}

/* ---- 2-bit packed strands: what the hit path reads -------------------------------------------
 * 16 bases per 32-bit word, base i of a strand in bits 2(i%16)..2(i%16)+1 of word i/16, zeros beyond
 * the end.  The l-mer starting at base pos is the low 2L bits of the 64-bit value (word[pos/16 + 1] :
 * word[pos/16]) >> 2(pos%16) -- one funnel shift (v_alignbit_b32) on two neighbouring words.  A hit
 * gathers two words of the row lane's packed positions (64 lanes x 21 words per tile: a few cache
 * lines, L1 resident) and two words of the column strand from LDS, instead of one 4-byte table entry
 * per l-mer and side: the per-l-mer tables made every gather touch 64 different cache lines, and the
 * vector memory pipeline, not the VALU, set the pace of the hit path (config 2: 18 of 91 ms). */
GKM_HD uint32_t pk_word(const uint8_t *codes, int len, int strand, int x)
{
    uint32_t v = 0u;
    for (int k = 0; k < 16; k++) {
        const int i = x * 16 + k;
        if (i >= len) break;
        const uint32_t c = strand ? (3u - codes[len - 1 - i]) : codes[i];
        v |= c << (2 * k);
    }
    return v;
}
/* the same with the strand continued cyclically behind its end (base i = base i mod len) */
GKM_HD uint32_t pk_word_cyclic(const uint8_t *codes, int len, int strand, int x)
{
    uint32_t v = 0u;
    int i = (x * 16) % len;
    for (int k = 0; k < 16; k++) {
        const uint32_t c = strand ? (3u - codes[len - 1 - i]) : codes[i];
        v |= c << (2 * k);
        if (++i == len) i = 0;
    }
    return v;
}
GKM_HD uint32_t pk_window(uint32_t lo, uint32_t hi, int pos)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(2 * (pos & 15)));
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (2 * (pos & 15)));
#endif
}
/* Hamming distance of the first L bases of two packed windows */
GKM_HD int pk_mismatch(uint32_t a, uint32_t b, int L)
{
    uint32_t t = (a ^ b) & ((1u << (2 * L)) - 1u);
    t = (t | (t >> 1)) & 0x55555555u;
    return __builtin_popcount(t);
}

/* Resolve one hit from packed data: bit b of the hit word of (delta, w, strand) against a row lane.
 *   lane position i0 = b*W + w; column l-mer q = (i0 + delta) mod T on `strand` (rejected if it wraps)
 *   roww(i0)        32-bit packed window of the row LANE's positions starting at i0
 *   colw(strand, q) 32-bit packed window of the column strand starting at base q
 *   wdist(D)        positional weight at distance D from the centre l-mer (libgkm.c:912-925); 1 if unweighted
 *   c0              (l-mers of the row)/2 - p0 + b0*W of the piece that owns the bit row, so that the
 *                   row l-mer's distance to its sequence's centre l-mer is |c0 - i0|
 * The reverse strand's weights are the forward ones mirrored: wt_rc[q] = wt[nB-1-q] (libgkm.c:924). */
template <int W, class RowWin, class ColWin, class Wdist>
GKM_HD HitValue resolve_hit_packed(int b, int w, int delta, int strand, uint32_t T, uint32_t rcpT, int nB, int L, int c0,
                                   RowWin roww, ColWin colw, Wdist wdist)
{
    HitValue r;
    r.m = 0;
    r.v = 0u;
    const int i0 = b * W + w;
    const uint32_t x = (uint32_t)(i0 + delta);
    int q;
    if (T >= (uint32_t)(32 * W)) { /* x < 2T: one conditional subtraction (uniform test) */
        const uint32_t y = x - T;
        q = (int)(y < x ? y : x);
    } else {
        q = (int)mod_small(x, T, rcpT);
    }
    if (q >= nB) return r; /* window wraps around the end of the strand: not an l-mer (see window_hits) */
    r.m = pk_mismatch(roww(i0), colw(strand, q), L);
    const int da = c0 - i0, qd = strand ? nB - 1 - q : q, db = nB / 2 - qd;
    r.v = wdist(da < 0 ? -da : da) * wdist(db < 0 ? -db : db);
    return r;
}

/* Origin word of a hit record.  The layout is chosen for the instruction count of the trip that consumes it, priced
 * with the issue rates measured on gfx950 (tools/valu_ops.hip, profiles/r3_valu_ops.txt: v_and / v_or / v_add / v_sub /
 * v_xor / v_not / v_mov / v_lshrrev / v_ashrrev / v_bitop3 issue every ~2.2 cycles per SIMD; everything else -- v_lshlrev
 * (also by a constant: 4.16, which is why the kernel doubles with v_add_u32 x, x), v_bfe, v_mad_u32_u24, v_min, v_sad,
 * v_alignbit, v_ffbl, v_bcnt, compares, SDWA -- every ~4.2; an SGPR operand costs a stream made of nothing else 4.2 too,
 * but nothing in a mix with VGPR-only instructions: 2.2 is what tools/issue_model.py prices it at):
 *   bits  0..3   word index w within the shift (0..W-1; a trip adds the word's offset in its group)       ms & 15
 *   bit   4      strand (0 forward, 1 reverse complement), bit 5: strand & [the column has an even number of l-mers]
 *                (the reverse strand's weights are the forward ones mirrored, wt_rc[q] = wt[nB-1-q] = wd[|q + even -
 *                nB/2|], libgkm.c:924) -- used by the CPU model of the hit path (bitslice_cpu_probe.cpp) only: since
 *                round 5 the kernel empties its hit list between the two strands, the strand is wave-uniform inside a
 *                trip and the kernel leaves both bits 0
 *   bits  7..12  source lane; ms & 0x1F80 is the byte offset of that lane's packed positions (128 bytes per lane)
 *   bits 21..31  shift delta (0..2046): ms >> 21
 *   (k_gram_bitslice's same-length variant also keeps, set once per wave by the source lane:
 *   bits  4..6   piece index of the lane within its row (the lane holds sequence positions pi * capacity ..)
 *   bits 13..18  row slot of the lane; (ms >> 11) & 0xFC is its byte offset in a profile row) */
constexpr uint32_t META_LANE_SHIFT = 7;
constexpr uint32_t META_PIECE_SHIFT = 4;
constexpr uint32_t META_SLOT_SHIFT = 13;
GKM_HD uint32_t pack_meta(int delta, int w, int strand, int even_adj = 0)
{
    return (uint32_t)w | ((uint32_t)strand << 4) | ((uint32_t)(strand & even_adj) << 5) | ((uint32_t)delta << 21);
}
GKM_HD int rec_w(uint32_t r) { return (int)(r & 15u); }
GKM_HD int rec_delta(uint32_t r) { return (int)(r >> 21); }
GKM_HD int rec_strand(uint32_t r) { return (int)((r >> 4) & 1u); }
GKM_HD int rec_lane(uint32_t r) { return (int)((r >> META_LANE_SHIFT) & 63u); }

} /* namespace gkmbs */
#endif
