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
#include "gkm_pack.h"

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
    auto wdist = [&](int dd) { return wd ? (uint32_t)wd[dd] : 1u; };
    /* packed column strands, as the kernel holds them in LDS */
    const int pkw = (T + 15) / 16 + 1;
    std::vector<uint32_t> pkB[2];
    for (int st = 0; st < 2; st++) {
        pkB[st].resize((size_t)pkw);
        for (int x = 0; x < pkw; x++) pkB[st][(size_t)x] = pk_word(B, T, st, x);
    }
    for (int s0 = 0; s0 < nA; s0 += CAP) {
        uint32_t Ahi[W], Alo[W], AV[W];
        for (int w = 0; w < W; w++) {
            Ahi[w] = row_plane_word(A, lenA, s0, w, W, L, 0);
            Alo[w] = row_plane_word(A, lenA, s0, w, W, L, 1);
            AV[w] = row_plane_word(A, lenA, s0, w, W, L, 2);
        }
        /* the lane's packed positions: base i of the segment is sequence position s0 + i */
        uint32_t lanepk[2 * W + 2];
        for (int x = 0; x < 2 * W + 2; x++) {
            uint32_t v = 0u;
            for (int k = 0; k < 16; k++) {
                const int pos = s0 + x * 16 + k;
                if (x * 16 + k < 32 * W && pos < lenA) v |= (uint32_t)A[pos] << (2 * k);
            }
            lanepk[x] = v;
        }
        auto rl = [&](int i0) { return pk_window(lanepk[i0 >> 4], lanepk[(i0 >> 4) + 1], i0); };
        auto cl = [&](int st, int q) { return pk_window(pkB[st][(size_t)(q >> 4)], pkB[st][(size_t)(q >> 4) + 1], q); };
        const int c0 = nA / 2 - s0;
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
                        const HitValue hv = resolve_hit_packed<W>(bit, w, delta, st, (uint32_t)T, rcpT, nB, L, c0, rl, cl, wdist);
                        if (hv.m <= D) acc[hv.m] += hv.v; /* (a wrapped window comes back as m = 0, v = 0) */
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
    CASE(10, 11, 3) CASE(10, 10, 3) CASE(10, 12, 4) CASE(10, 8, 4) CASE(10, 9, 4) CASE(10, 12, 6) CASE(10, 11, 5) CASE(10, 12, 5)
    CASE(10, 4, 2) CASE(10, 2, 1) CASE(10, 3, 0) CASE(10, 5, 2) CASE(10, 6, 3) CASE(10, 7, 3)
    CASE(5, 11, 3) CASE(16, 11, 3) CASE(3, 12, 4) CASE(10, 12, 8) CASE(10, 12, 12) CASE(10, 11, 1)
    CASE(10, 9, 5) CASE(10, 10, 5) CASE(10, 10, 6) CASE(10, 11, 6) CASE(10, 11, 7) CASE(10, 12, 7) /* more d > 4 pairs (the device table holds (10,5), (11,5), (12,5), (12,6) of them) */
    if (!ok) return 1;
    for (int m = 0; m <= d; m++) P[m] = (int32_t)acc[m];
    return 0;
}

/* ---- packed lanes: several row sequences against one column sequence ------------------------
 * Packs the rows with gkmpack::pack_rows, checks the packing invariants, builds every lane's
 * planes from its pieces, runs the lane program and attributes each hit to its piece's row.
 * P_out[i*(d+1)+m] for row i.  Returns 0, or a negative code naming the violated invariant. */
template <int W, int L, int D>
static int run_packed(const uint8_t *codes, const int64_t *off, const int *rows, int nrows, int col,
                      const uint8_t *wd, int32_t *P_out, int *lanes_used)
{
    using namespace gkmpack;
    std::vector<int> nwin((size_t)nrows);
    for (int i = 0; i < nrows; i++) nwin[(size_t)i] = (int)(off[rows[i] + 1] - off[rows[i]]) - L + 1;
    const Packing P = pack_rows(rows, nwin.data(), nrows, W, L);
    *lanes_used = (int)P.lanes_used;
    /* invariants: every window of every row owned exactly once; pieces inside their lane; all
     * pieces of a row in one tile; limits respected */
    std::vector<std::vector<int>> owned((size_t)nrows);
    for (int i = 0; i < nrows; i++) owned[(size_t)i].assign((size_t)nwin[(size_t)i], 0);
    std::vector<int> row_tile((size_t)nrows, -1);
    std::vector<int> lane_bits((size_t)P.ntiles * LANES, 0), lane_np((size_t)P.ntiles * LANES, 0);
    for (const Piece &pc : P.pieces) {
        if (pc.b0 < 0 || pc.nb <= 0 || pc.b0 + pc.nb > 32) return -1;
        if (pc.cnt <= 0 || pc.cnt > pc.nb * W - (L - 1)) return -2;
        if (pc.b0 != lane_bits[(size_t)pc.lane]) return -3; /* contiguous, in order */
        lane_bits[(size_t)pc.lane] += pc.nb;
        if (++lane_np[(size_t)pc.lane] > MAX_PIECES) return -4;
        const int t = pc.lane / LANES;
        const int i = P.tile_out[(size_t)t * MAX_ROWS + pc.slot];
        if (i < 0 || i >= nrows || rows[i] != pc.row || P.tile_row[(size_t)t * MAX_ROWS + pc.slot] != pc.row) return -5;
        if (row_tile[(size_t)i] >= 0 && row_tile[(size_t)i] != t) return -6;
        row_tile[(size_t)i] = t;
        for (int k = 0; k < pc.cnt; k++) {
            if (pc.p0 + k >= nwin[(size_t)i]) return -7;
            owned[(size_t)i][(size_t)(pc.p0 + k)]++;
        }
    }
    for (int i = 0; i < nrows; i++)
        for (int v : owned[(size_t)i])
            if (v != 1) return -8;
    for (int t = 0; t < P.ntiles; t++)
        if (P.tile_nrows[(size_t)t] > MAX_ROWS) return -9;

    /* the computation */
    constexpr int NB = planes_for(D);
    (void)NB;
    const uint8_t *B = codes + off[col];
    const int T = (int)(off[col + 1] - off[col]), nB = T - L + 1;
    std::vector<uint32_t> sb[2][2];
    for (int st = 0; st < 2; st++)
        for (int pl = 0; pl < 2; pl++) {
            sb[st][pl].resize((size_t)T + W);
            for (int x = 0; x < T + W; x++) sb[st][pl][(size_t)x] = sb_word(B, T, st, x, W, L, pl);
        }
    auto wdist = [&](int Dd) { return wd ? (uint32_t)wd[Dd] : 1u; };
    const int pkw = (T + 15) / 16 + 1;
    std::vector<uint32_t> pkB[2];
    for (int st = 0; st < 2; st++) {
        pkB[st].resize((size_t)pkw);
        for (int x = 0; x < pkw; x++) pkB[st][(size_t)x] = pk_word(B, T, st, x);
    }
    const uint32_t rcpT = mod_magic((uint32_t)T);
    std::vector<uint32_t> acc((size_t)nrows * (D + 1), 0u);
    size_t pi = 0;
    while (pi < P.pieces.size()) {
        size_t pj = pi;
        while (pj < P.pieces.size() && P.pieces[pj].lane == P.pieces[pi].lane) pj++;
        uint32_t Ahi[W], Alo[W], AV[W], start_mask = 0u;
        uint32_t lanepk[2 * W + 2]; /* the lane's packed positions (what k_build_rowplanes writes as plane 3) */
        for (int w = 0; w < W; w++) Ahi[w] = Alo[w] = AV[w] = 0u;
        for (int x = 0; x < 2 * W + 2; x++) lanepk[x] = 0u;
        for (size_t k = pi; k < pj; k++) {
            const Piece &pc = P.pieces[k];
            start_mask |= 1u << pc.b0;
            const uint8_t *seq = codes + off[pc.row];
            const int len = (int)(off[pc.row + 1] - off[pc.row]);
            for (int w = 0; w < W; w++)
                for (int b = 0; b < 32; b++) {
                    Ahi[w] |= piece_bit(seq, len, pc.b0, pc.nb, pc.p0, pc.cnt, b, w, W, 0) << b;
                    Alo[w] |= piece_bit(seq, len, pc.b0, pc.nb, pc.p0, pc.cnt, b, w, W, 1) << b;
                    AV[w] |= piece_bit(seq, len, pc.b0, pc.nb, pc.p0, pc.cnt, b, w, W, 2) << b;
                    const int i = b * W + w;
                    lanepk[i >> 4] |= ((piece_bit(seq, len, pc.b0, pc.nb, pc.p0, pc.cnt, b, w, W, 0) << 1) |
                                       piece_bit(seq, len, pc.b0, pc.nb, pc.p0, pc.cnt, b, w, W, 1)) << (2 * (i & 15));
                }
        }
        for (int st = 0; st < 2; st++)
            for (int delta = 0; delta < T; delta++) {
                uint32_t hit[W];
                window_hits<W, L, D>(Ahi, Alo, AV, &sb[st][0][(size_t)delta], &sb[st][1][(size_t)delta],
                                     (const uint32_t *)nullptr, hit);
                for (int w = 0; w < W; w++) {
                    uint32_t h = hit[w];
                    while (h) {
                        const int bit = __builtin_ctz(h);
                        h &= h - 1u;
                        const Piece &pc = P.pieces[pi + (size_t)piece_of_bitrow(start_mask, bit)];
                        const uint8_t *seq = codes + off[pc.row];
                        const int len = (int)(off[pc.row + 1] - off[pc.row]), nA = len - L + 1;
                        /* the device keeps c0 = nA/2 - p0 + b0*W per piece: the row l-mer of lane position i0 is
                         * l-mer p = p0 + i0 - b0*W of its sequence, |c0 - i0| away from the centre l-mer */
                        auto rl = [&](int i0) { return pk_window(lanepk[i0 >> 4], lanepk[(i0 >> 4) + 1], i0); };
                        auto cl = [&](int s2, int q) { return pk_window(pkB[s2][(size_t)(q >> 4)], pkB[s2][(size_t)(q >> 4) + 1], q); };
                        const int c0 = nA / 2 - pc.p0 + pc.b0 * W;
                        (void)seq;
                        const HitValue hv = resolve_hit_packed<W>(bit, w, delta, st, (uint32_t)T, rcpT, nB, L, c0, rl, cl, wdist);
                        const int i = P.tile_out[(size_t)(pc.lane / LANES) * MAX_ROWS + pc.slot];
                        if (hv.m <= D) acc[(size_t)i * (D + 1) + hv.m] += hv.v;
                        else if (hv.v != 0u) return -20; /* a true hit can never exceed D */
                    }
                }
            }
        pi = pj;
    }
    for (size_t k = 0; k < acc.size(); k++) P_out[k] = (int32_t)acc[k];
    return 0;
}

#define PCASE(WW, LL, DD) \
    if (W == WW && L == LL && d == DD) return run_packed<WW, LL, DD>(codes, off, rows, nrows, col, wd, P, lanes_used);

extern "C" int bsprobe_profile_packed(int W, int L, int d, const uint8_t *codes, const int64_t *off, const int *rows,
                                      int nrows, int col, const uint8_t *wd, int32_t *P, int *lanes_used)
{
    PCASE(10, 11, 3) PCASE(20, 11, 3) PCASE(10, 12, 4) PCASE(20, 12, 4) PCASE(20, 10, 3) PCASE(10, 6, 2) PCASE(20, 6, 2)
    PCASE(5, 11, 3)
    return 1;
}

/* ---- row sharding layout (gkm_shard.h) for tests/test_sharding.py ---- */
#include "gkm_shard.h"

extern "C" int shardprobe_chunk_rows(int n, int world, int chunks) { return gkmshard::chunk_rows(n, world, chunks); }

extern "C" int shardprobe_part(int n, int world, int rank, int chunks, int chunk, int *rows_out)
{
    const std::vector<std::vector<int>> parts = gkmshard::chunked_layout(n, world, rank, chunks);
    const std::vector<int> &p = parts[(size_t)chunk];
    for (size_t i = 0; i < p.size(); i++) rows_out[i] = p[i];
    return (int)p.size();
}

extern "C" void shardprobe_gather_index(int n, int world, int chunks, int64_t *slot_out)
{
    const std::vector<int64_t> s = gkmshard::chunked_gather_index(n, world, chunks);
    for (int i = 0; i < n; i++) slot_out[i] = s[(size_t)i];
}

extern "C" long long shardprobe_packed_chunk_elems(int n, int world, int chunks)
{
    return (long long)gkmshard::packed_chunk_elems(n, world, chunks);
}

extern "C" void shardprobe_packed_gather_offsets(int n, int world, int chunks, int64_t *off_out)
{
    const std::vector<int64_t> s = gkmshard::packed_gather_offsets(n, world, chunks);
    for (int i = 0; i < n; i++) off_out[i] = s[(size_t)i];
}

extern "C" int shardprobe_auto_chunks(int n, int world) { return gkmshard::auto_chunks(n, world); }

/* work items (tile, column) of a triangular launch over `rows` packed with / without closing tiles at jumps in the row
 * list (gkm_pack.h pack_rows split_jump), and the number of tiles; every row must have found a slot */
extern "C" long long packprobe_triangle_items(const int *rows, const int *nwin, int nrows, int W, int L, int max_rows,
                                              int split_jump, int *ntiles_out)
{
    const gkmpack::Packing P = gkmpack::pack_rows(rows, nwin, nrows, W, L, max_rows, split_jump);
    long placed = 0;
    for (int t = 0; t < P.ntiles; t++) placed += P.tile_nrows[(size_t)t];
    if (placed != nrows) return -1;
    if (ntiles_out) *ntiles_out = P.ntiles;
    return gkmpack::triangle_items(P);
}
