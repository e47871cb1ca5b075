/*
 * gkm_gram.hip -- one launch of the Gram kernel for a set of rows: which kernel, the row packing (gkm_pack.h), the order
 * of the work items, the per-launch tables; plus the small kernels either side of the hot one and the general kernel.
 *
 * Kernels
 *   k_build_rowplanes row-segment bit planes + the lanes' packed positions for one set of rows
 *   k_untile          tile-transposed values -> matrix rows (128-byte lines in, 512-byte runs out; one-wave workgroups)
 *   k_gram_direct     general fallback: per-l-mer tables, l-mer by l-mer XOR/popcount (any L <= 12, d <= 12)
 * (k_gram_bitslice, the hot kernel: gkm_gram_bitslice.hip)
 */
#include "gkm_gram_bitslice.h"

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

/* S (tile-transposed, see BsArgs) -> rows of G.  Block = 64 columns x UT_SLOTS row slots of one tile, moved through LDS
 * so that the reads are whole 128-byte lines (16 slots of one column) and the writes 512-byte runs (64 columns of one
 * row).  ONE wave and 8.5 KB of LDS per workgroup, on purpose: rounds 2-5 used 256 threads and 33 KB (64 x 64 values), and
 * such a workgroup cannot START beside a Gram kernel -- whose 28 one-wave workgroups per CU leave 17-20 KB of LDS and no
 * four free wave slots at one time -- so the untile of a multi-GPU rank's chunk c, on its own stream, ended when the Gram
 * kernel of chunk c+1 did and the transfer of chunk c hid behind nothing (round 5, tools/rank_alone.py,
 * profiles/r5_rank_alone_stream_policies.txt).  grid (column blocks, tiles * NSLOT / UT_SLOTS). */
constexpr int UT_SLOTS = 16;
template <int NSLOT>
__global__ __launch_bounds__(64) void k_untile(const double *__restrict__ S, const int64_t *__restrict__ tile_soff,
                                               const int *__restrict__ tile_cbeg, const int *__restrict__ tile_cend,
                                               const int *__restrict__ tile_nrows, const int *__restrict__ tile_row,
                                               const int *__restrict__ tile_out, GramOut out)
{
    __shared__ double buf[64][UT_SLOTS + 1];
    constexpr int PARTS = NSLOT / UT_SLOTS;
    const int tile = blockIdx.y / PARTS, part = blockIdx.y % PARTS;
    const int cbeg = tile_cbeg[tile], cend = tile_cend[tile], nrows = tile_nrows[tile];
    const int jb = cbeg + (int)blockIdx.x * 64;
    if (jb >= cend || part * UT_SLOTS >= nrows) return;
    const int lane = threadIdx.x, ts = lane % UT_SLOTS, tc = lane / UT_SLOTS;
    const double *src = S + (tile_soff[tile] + (jb - cbeg)) * NSLOT + part * UT_SLOTS;
#pragma unroll 4
    for (int c = tc; c < 64; c += 64 / UT_SLOTS)
        if (jb + c < cend && part * UT_SLOTS + ts < nrows) buf[c][ts] = src[(int64_t)c * NSLOT + ts];
    __syncthreads();
    const int j = jb + lane;
    for (int rl = 0; rl < UT_SLOTS; rl++) {
        const int rs = part * UT_SLOTS + rl;
        if (rs >= nrows) break;
        const int row = tile_row[tile * gkmpack::MAX_ROWS + rs];
        if (j >= cend || (j > row && !out.write_all)) continue;
        const int64_t r = out.local_rows ? tile_out[tile * gkmpack::MAX_ROWS + rs] : row;
        *gram_cell(out, r, j) = buf[lane][rl];
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
    static_assert(GKM_MAXD1 >= 12 + 1, "acc has a row for every possible mismatch count m <= L <= 12 (no test for m <= d below)");
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
/* The share `auto` goes by: the iid one, or what set_sequences sampled on the uploaded sequences when that is HIGHER by more
 * than three standard deviations of its sampling noise (8 192 pairs: 0.28 % at the threshold) -- repeats and
 * low-complexity input push real data above the iid rate, never far below it. */
static double decisive_hit_share(const gkmhip_ctx *ctx)
{
    const double iid = iid_hit_share(ctx->L, ctx->d);
    const double noise = 3.0 * sqrt(iid * (1.0 - iid) / 8192.0);
    return ctx->sampled_hit_share > iid + noise ? ctx->sampled_hit_share : iid;
}
static bool auto_takes_bitslice(const gkmhip_ctx *ctx) { return decisive_hit_share(ctx) <= GKM_BITSLICE_MAX_HIT_SHARE; }

bool bitslice_serves(const gkmhip_ctx *ctx)
{
    if (ctx->kernel_pref == GKMHIP_KERNEL_DIRECT || !gkm_pick_bitslice(2, ctx->L, ctx->d)) return false;
    return ctx->kernel_pref == GKMHIP_KERNEL_BITSLICE || auto_takes_bitslice(ctx);
}


/* ---- the bit-sliced launch, host side: what a launch needs that depends on the rows and not on the device ---- */
struct BitslicePlan {
    gkmpack::Packing pk;              /* rows -> lanes (gkm_pack.h) */
    int slots = 64;                   /* row slots per tile: 64 or 128 */
    bool packed = false, same_length = false;
    bs_kernel_t kernel = nullptr;
    const char *name = "";
    size_t dyn_lds = 0;
    std::vector<int> desc;            /* [lane][piece][5] row, b0, nb, p0, cnt: what k_build_rowplanes reads */
    std::vector<uint32_t> lane_mask, lane_piece;
    std::vector<int> cbeg, cend;      /* columns [cbeg, cend) per tile */
    std::vector<int64_t> soff;        /* first work item of each tile in the plain order; soff[ntiles] = items */
    std::vector<int64_t> ent_off;     /* (column chunk, tile) entries: the other order of the work items (BsArgs) */
    std::vector<int> ent_tile, ent_j0, ent_j1;
    int64_t n_items = 0;
};

/* Pack the rows, choose the kernel variant, fill the per-lane tables, the tiles' column ranges and the work-item order.
 * mode says which columns every tile of rows visits: COLS_TRIANGLE j <= largest row of the tile (the path of
 * gkm_main_pywrapper), COLS_FULL every sequence, COLS_DIAGONAL only the band of the tile's own rows (self norms). */
static int plan_bitslice(const gkmhip_ctx *ctx, const int *rows, int nrows, int mode, BitslicePlan &P)
{
    const int L = ctx->L, d = ctx->d, n = ctx->n;
    std::vector<int> nwin((size_t)nrows);
    for (int i = 0; i < nrows; i++) nwin[(size_t)i] = ctx->h_len[(size_t)rows[i]] - L + 1;
    /* Every variant resolves hits by GROUPS of five lane positions (k_gram_bitslice): a piece that does not finish its row
     * owns a multiple of five windows.  Same-length problems take the variant that needs neither piece table nor permute
     * (PK = 4); GKM_FORCE_PACKED=1|128 (tests, A/B runs) puts them on the several-pieces variants. */
    P.same_length = ctx->minlen == ctx->maxlen;
    const bool unif = P.same_length && getenv("GKM_FORCE_PACKED") == nullptr && gkm_pick_bitslice(4, L, d) != nullptr;
    const int own_mult = 5;
    /* (a jump in the row list -- a multi-GPU rank's two folded row blocks -- closes the tile where that means fewer work
     * items: gkm_pack.h) */
    auto pack = [&](int max_rows) {
        gkmpack::Packing a = gkmpack::pack_rows(rows, nwin.data(), nrows, 10, L, max_rows, 0, own_mult);
        if (mode == COLS_FULL || rows[nrows - 1] - rows[0] + 1 == nrows) return a; /* (no jump, or every tile visits all columns) */
        gkmpack::Packing b = gkmpack::pack_rows(rows, nwin.data(), nrows, 10, L, max_rows, gkmpack::LANES, own_mult);
        return gkmpack::triangle_items(b) < gkmpack::triangle_items(a) ? b : a;
    };
    /* At most 64 rows per tile unless that leaves lanes empty (rows shorter than half a lane): the 64-slot kernels keep a
     * wave more per SIMD (k_gram_bitslice) */
    P.pk = pack(64);
    P.slots = 64;
    {
        gkmpack::Packing wide = pack(gkmpack::MAX_ROWS);
        if (getenv("GKM_FORCE_PACKED") ? !strcmp(getenv("GKM_FORCE_PACKED"), "128")
                                       : (double)P.pk.ntiles > 1.04 * (double)wide.ntiles) {
            P.pk = std::move(wide);
            P.slots = gkmpack::MAX_ROWS;
        }
    }
    const gkmpack::Packing &pk = P.pk;
    const int W = pk.W, ntiles = pk.ntiles, slots = P.slots;
    /* everything but a same-length problem whose rows fill whole lanes -- ragged one-piece data too -- takes the
     * several-pieces variants */
    bool packed = !unif || slots != 64;
    for (size_t k = 1; k < pk.pieces.size() && !packed; k++) packed = pk.pieces[k].lane == pk.pieces[k - 1].lane;
    P.packed = packed;
    P.kernel = gkm_pick_bitslice(!packed ? 4 : slots == 64 ? 1 : 2, L, d);
    P.name = !packed ? "k_gram_bitslice<same length>" : slots == 64 ? "k_gram_bitslice<packed>" : "k_gram_bitslice<packed,128>";
    if (!P.kernel) return set_err_msg("bit-sliced kernel not instantiated for this (L, d)", 5);
    const int NP = packed ? gkmpack::MAX_PIECES : 1, LPW = packed ? NP : 2;
    /* GKM_LDS_PAD=<bytes> (experiments): extra dynamic LDS per wave, i.e. fewer waves per CU -- how much does the
     * kernel depend on its occupancy? */
    const size_t lds_pad = getenv("GKM_LDS_PAD") ? (size_t)atoi(getenv("GKM_LDS_PAD")) : 0;
    /* dynamic LDS of a wave: the column's two packed strands + the column's weights by position with zeros either side
     * (ctx->ptw words) + the centred distance table for the row side in the several-pieces variants (the same-length
     * variant's rows read the column's table by position) */
    P.dyn_lds = (size_t)(2 * ctx->pkw) * sizeof(uint32_t) + lds_pad + (size_t)ctx->ptw * 4 + (packed ? (size_t)ctx->wdc_words * 4 : 0);
    static_assert(GKM_MAXLEN / (32 * 10 - 11) <= 7, "a row's piece index fits the 3 bits of the origin word");

    const size_t nl = (size_t)ntiles * 64;
    P.desc.assign(nl * gkmpack::MAX_PIECES * 5, 0);
    P.lane_mask.assign(nl, 0u);
    P.lane_piece.assign(nl * (size_t)LPW, 0u);
    std::vector<int> fill(nl, 0);
    for (const gkmpack::Piece &pc : pk.pieces) {
        const int k = fill[(size_t)pc.lane]++;
        int *dd = &P.desc[((size_t)pc.lane * gkmpack::MAX_PIECES + k) * 5];
        dd[0] = pc.row; dd[1] = pc.b0; dd[2] = pc.nb; dd[3] = pc.p0; dd[4] = pc.cnt;
        P.lane_mask[(size_t)pc.lane] |= 1u << pc.b0;
        /* second profile copy (k_gram_bitslice two_copies): odd lanes of a tile with at most slots / 2 rows */
        const bool second = 2 * pk.tile_nrows[(size_t)(pc.lane / 64)] <= slots && (pc.lane & 1);
        const uint32_t slot = (uint32_t)pc.slot + (second ? (uint32_t)slots / 2u : 0u);
        const int row_windows = ctx->h_len[(size_t)pc.row] - L + 1;
        if (packed) {
            /* slot | centre offset << 7 | owned windows << 20: the row l-mer at lane position i0 is i0 + 2048 - c0b l-mers
             * away from its sequence's centre l-mer (signed; the bias keeps c0b unsigned) */
            const uint32_t c0b = (uint32_t)(row_windows / 2 - pc.p0 + pc.b0 * W + 2048);
            if (slot > 127 || c0b > 0x1FFFu || pc.cnt > 511) return set_err_msg("gram: piece entry out of range", 2);
            P.lane_piece[(size_t)pc.lane * NP + k] = slot | (c0b << 7) | ((uint32_t)pc.cnt << 20);
        } else {
            /* row slot and piece index where the record's origin word wants them (gkm_bitslice.h); the piece index is the
             * piece's first position over the lane capacity: pieces of a same-length problem fill whole lanes */
            const int cap = gkmbs::segment_capacity(W, L) / own_mult * own_mult;
            if (pc.b0 != 0 || pc.p0 % cap != 0 || pc.p0 / cap > 7 || slot > 63 || (pc.cnt % own_mult != 0 && pc.p0 + pc.cnt != row_windows))
                return set_err_msg("gram: same-length packing broke its own rule", 2);
            P.lane_piece[(size_t)pc.lane * 2] = (slot << gkmbs::META_SLOT_SHIFT) | ((uint32_t)(pc.p0 / cap) << gkmbs::META_PIECE_SHIFT);
        }
    }
    P.cbeg.assign((size_t)ntiles, 0);
    P.cend.assign((size_t)ntiles, 0);
    P.soff.assign((size_t)ntiles + 1, 0);
    for (int t = 0; t < ntiles; t++) {
        int amin = n;
        for (int rs = 0; rs < pk.tile_nrows[(size_t)t]; rs++) amin = std::min(amin, pk.tile_row[(size_t)t * gkmpack::MAX_ROWS + rs]);
        P.cbeg[(size_t)t] = mode == COLS_DIAGONAL ? amin : 0;
        P.cend[(size_t)t] = mode == COLS_FULL ? n : pk.tile_amax[(size_t)t] + 1;
        P.soff[(size_t)t + 1] = P.soff[(size_t)t] + (P.cend[(size_t)t] - P.cbeg[(size_t)t]);
    }
    if (P.soff[(size_t)ntiles] <= 0 || P.soff[(size_t)ntiles] > 0x7fffffffLL) return set_err_msg("gram: bad work item count", 2);
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
    P.n_items = P.soff[(size_t)ntiles];
    const double mean_len = ctx->h_cum_n[(size_t)n] / n + (L - 1);
    const double col_bytes = (4.0 * (mean_len + W) + 2.0 * (mean_len / 16.0 + 1.0)) * sizeof(uint32_t);
    long chunk = col_bytes * 4096.0 / 8.0 <= 3.0 * 1048576.0 ? 4096 : 0;
    if (const char *cc = getenv("GKM_COL_CHUNK")) chunk = atol(cc) & ~7L;
    if (chunk >= 8 && chunk < n) {
        int64_t at = 0;
        for (long c0 = 0; c0 < n; c0 += chunk)
            for (int t = 0; t < ntiles; t++) {
                const int j0 = std::max<long>(P.cbeg[(size_t)t], c0), j1 = (int)std::min<long>(P.cend[(size_t)t], c0 + chunk);
                if (j0 >= j1) continue;
                P.ent_off.push_back(at);
                P.ent_tile.push_back(t);
                P.ent_j0.push_back(j0);
                P.ent_j1.push_back(j1);
                at += (j1 - j0 + 7) & ~7;
            }
        P.ent_off.push_back(at);
        if (at > 0x7fffffffLL) { P.ent_off.clear(); P.ent_tile.clear(); P.ent_j0.clear(); P.ent_j1.clear(); } /* (too many items with the padding: plain order) */
        else P.n_items = at;
    }
    return 0;
}

/* ---- the bit-sliced launch, device side: tables up in one copy, row planes, the Gram kernel, untile ---- */
static int enqueue_bitslice(gkmhip_ctx *ctx, const BitslicePlan &P, int nrows, GramOut out, hipStream_t stream)
{
    const gkmpack::Packing &pk = P.pk;
    const int W = pk.W, L = ctx->L, ntiles = pk.ntiles;
    const size_t nl = (size_t)ntiles * 64;
    /* every per-launch table goes to the device in ONE copy (the boundary call issues a few launches, a multi-GPU rank
     * one per chunk) */
    std::vector<char> blob;
    auto put = [&](const void *src, size_t bytes) {
        const size_t at = (blob.size() + 255) & ~(size_t)255;
        blob.resize(at + bytes);
        memcpy(blob.data() + at, src, bytes);
        return at;
    };
    const size_t o_desc = put(P.desc.data(), P.desc.size() * sizeof(int));
    const size_t o_mask = put(P.lane_mask.data(), nl * sizeof(uint32_t));
    const size_t o_piece = put(P.lane_piece.data(), P.lane_piece.size() * sizeof(uint32_t));
    const size_t o_trow = put(pk.tile_row.data(), pk.tile_row.size() * sizeof(int));
    const size_t o_tout = put(pk.tile_out.data(), pk.tile_out.size() * sizeof(int));
    const size_t o_tn = put(pk.tile_nrows.data(), (size_t)ntiles * sizeof(int));
    const size_t o_cbeg = put(P.cbeg.data(), (size_t)ntiles * sizeof(int));
    const size_t o_cend = put(P.cend.data(), (size_t)ntiles * sizeof(int));
    const size_t o_soff = put(P.soff.data(), P.soff.size() * sizeof(int64_t));
    const size_t o_roff = out.row_off ? put(out.row_off, (size_t)nrows * sizeof(int64_t)) : 0;
    const int nent = (int)P.ent_tile.size();
    const size_t o_eoff = nent ? put(P.ent_off.data(), P.ent_off.size() * sizeof(int64_t)) : 0;
    const size_t o_etile = nent ? put(P.ent_tile.data(), (size_t)nent * sizeof(int)) : 0;
    const size_t o_ej0 = nent ? put(P.ent_j0.data(), (size_t)nent * sizeof(int)) : 0;
    const size_t o_ej1 = nent ? put(P.ent_j1.data(), (size_t)nent * sizeof(int)) : 0;
    auto &scr = ctx->scratch[ctx->sel];
    /* words of a lane's packed positions: 32 W / 16 + 1 are used (the hit path reads two); the stride is 128 bytes,
     * so that the lane field of a record's origin word is the lane's byte offset (gkm_bitslice.h pack_meta) */
    const int rpw = 32;
    static_assert(32 * 10 / 16 + 1 <= 32, "a lane's packed positions fit 128 bytes");
    PinBuf *hb = pin_acquire(blob.size());
    if (!hb) return set_err_msg("gram: pinned host buffer for the launch tables", 4);
    if (scr.tables.ensure(blob.size(), true) || scr.rowplanes.ensure(nl * 3 * W, true) || scr.rowpk.ensure(nl * (size_t)rpw, true) ||
        (out.G && scr.S.ensure((size_t)P.soff[(size_t)ntiles] * (size_t)P.slots, true))) {
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
    /* (This upload and the row-plane kernel are the ~0.1 ms between two Gram kernels of one stream.  A multi-GPU rank's
     * transfer of the previous chunk needs exactly that head start to get onto the device: gkm_multi.hip rank_thread.) */
    char *tb = scr.tables.p;
    static_assert(10 * 4 >= 32, "plane 3: ten parts of four packed words cover the lane's 32");
    hipLaunchKernelGGL(k_build_rowplanes, dim3((unsigned)ntiles, 4, (unsigned)W), dim3(64), 0, stream, ctx->codes.p, ctx->off.p,
                       (const int *)(tb + o_desc), W, scr.rowplanes.p, scr.rowpk.p, rpw);
    HIPCHK(hipGetLastError());

    BsArgs A;
    A.rowplanes = scr.rowplanes.p; A.lane_mask = (const uint32_t *)(tb + o_mask); A.lane_piece = (const uint32_t *)(tb + o_piece);
    A.tile_row = (const int *)(tb + o_trow); A.tile_out = (const int *)(tb + o_tout); A.tile_nrows = (const int *)(tb + o_tn);
    A.tile_cbeg = (const int *)(tb + o_cbeg); A.tile_cend = (const int *)(tb + o_cend);
    A.rowpk = scr.rowpk.p; A.colpk = ctx->colpk.p;
    A.rpw = rpw; A.pkw = ctx->pkw; A.postab = ctx->postab.p; A.ptw = ctx->ptw; A.ptw_stride = P.same_length ? 0 : ctx->ptw;
    A.cap = gkmbs::segment_capacity(W, L) / 5 * 5;
    A.wdc = ctx->wdc.p; A.wdc_words = ctx->wdc_words; A.wdc_centre = ctx->wdc_centre;
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
    /* One column sequence per work item: a wave of the full-size problem lives ~0.6 ms, which is what the drain at the end
     * of every launch costs -- nothing for one big launch, 0.2-0.3 ms for each of a multi-GPU rank's chunks. */
    hipEvent_t e0, e1;
    if (gkm_launch_events(ctx, &e0, &e1)) return 4;
    HIPCHK(hipEventRecord(e0, stream));
    hipLaunchKernelGGL(P.kernel, dim3((unsigned)P.n_items), dim3(64), P.dyn_lds, stream, A);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e1, stream));
    if (out.G) {
        int span = 0;
        for (int t = 0; t < ntiles; t++) span = std::max(span, P.cend[(size_t)t] - P.cbeg[(size_t)t]);
        const dim3 ug((unsigned)((span + 63) / 64), (unsigned)(ntiles * (P.slots / UT_SLOTS)));
        if (P.slots != 64)
            hipLaunchKernelGGL(k_untile<gkmpack::MAX_ROWS>, ug, dim3(64), 0, stream, scr.S.p, A.tile_soff, A.tile_cbeg, A.tile_cend,
                               A.tile_nrows, A.tile_row, A.tile_out, A.out);
        else
            hipLaunchKernelGGL(k_untile<64>, ug, dim3(64), 0, stream, scr.S.p, A.tile_soff, A.tile_cbeg, A.tile_cend, A.tile_nrows,
                               A.tile_row, A.tile_out, A.out);
        HIPCHK(hipGetLastError());
    }
    if (getenv("GKM_TRACE")) { /* the occupancy the runtime grants this instantiation with this much dynamic LDS */
        int per_cu = 0;
        hipFuncAttributes fa;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)P.kernel, 64, P.dyn_lds) == hipSuccess &&
            hipFuncGetAttributes(&fa, (const void *)P.kernel) == hipSuccess)
            fprintf(stderr, "gkmhip: hot kernel: %d VGPRs, %zu + %zu bytes of LDS per wave, %d one-wave workgroups per CU\n",
                    fa.numRegs, (size_t)fa.sharedSizeBytes, P.dyn_lds, per_cu);
    }
    ctx->last_kernel = P.name;
    return 0;
}

/* ---- the general kernel's launch ---- */
static int enqueue_direct(gkmhip_ctx *ctx, const int *rows, int nrows, int mode, GramOut out, hipStream_t stream)
{
    const int n = ctx->n;
    auto &scr = ctx->scratch[ctx->sel];
    if (ensure_lmers(ctx, stream)) return 4;
    if (scr.rows.ensure((size_t)nrows)) return 4;
    HIPCHK(hipMemcpyAsync(scr.rows.p, rows, (size_t)nrows * sizeof(int), hipMemcpyHostToDevice, stream));
    if (out.row_off) {
        if (scr.rowoff.ensure((size_t)nrows)) return 4;
        HIPCHK(hipMemcpyAsync(scr.rowoff.p, out.row_off, (size_t)nrows * sizeof(int64_t), hipMemcpyHostToDevice, stream));
        out.row_off = scr.rowoff.p;
    }
    HIPCHK(hipStreamSynchronize(stream)); /* `rows` is the caller's: see gkmhip_set_sequences */
    DirectArgs A;
    A.rows = scr.rows.p; A.nrows = nrows;
    A.len = ctx->len.p; A.lmoff = ctx->lmoff.p; A.lmf = ctx->lmf.p; A.lmr = ctx->lmf.p + ctx->lm_stride;
    for (int m = 0; m < GKM_MAXD1; m++) A.c[m] = ctx->c[m];
    A.out = out;
    A.L = ctx->L; A.d = ctx->d; A.mode = mode; A.n = n;
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
    hipEvent_t e0, e1;
    if (gkm_launch_events(ctx, &e0, &e1)) return 4;
    HIPCHK(hipEventRecord(e0, stream));
    hipLaunchKernelGGL(k_gram_direct, dim3(nchunks, ntiles), dim3(64), 0, stream, A);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e1, stream));
    ctx->last_kernel = "k_gram_direct";
    return 0;
}

/* One launch of the Gram kernel for a set of rows (mode: see plan_bitslice). */
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
    /* (W = 10 words per lane; W = 20 was measured too -- config 2: 121 vs 118 ms, 150 bp: 56 vs 31 ms in round 1: the longer
     * per-shift chain does not pay for the registers it costs) */
    if (ctx->kernel_pref == GKMHIP_KERNEL_BITSLICE && !gkm_pick_bitslice(2, L, d))
        return set_err_msg("bit-sliced kernel not instantiated for this (L, d)", 5);
    int rc;
    if (bitslice_serves(ctx)) { /* (auto: the general kernel where it is the faster one) */
        /* the per-sequence tables (normally built by gkmhip_set_sequences); ctx->pkw and ctx->ptw size the plan's dynamic LDS */
        if (ensure_sb(ctx, 10, stream) || ensure_colpk(ctx, stream) || ensure_postab(ctx, stream)) return 4;
        BitslicePlan P;
        rc = plan_bitslice(ctx, rows, nrows, mode, P);
        if (!rc) rc = enqueue_bitslice(ctx, P, nrows, out, stream);
    } else {
        rc = enqueue_direct(ctx, rows, nrows, mode, out, stream);
    }
    if (rc) return rc;
    ctx->ev_valid = true;
    ctx->last_comparisons = comparisons;
    if (getenv("GKM_TRACE")) /* which kernel served this launch, and the rule's input: a regression on data far from iid shows here */
        fprintf(stderr, "gkmhip: %d rows -> %s (preference %d; hit share of (L=%d, d=%d): iid %.4f, sampled on these sequences %.4f; "
                        "bit-sliced up to %.3f)\n", nrows, ctx->last_kernel, ctx->kernel_pref, L, d, iid_hit_share(L, d),
                ctx->sampled_hit_share, (double)GKM_BITSLICE_MAX_HIT_SHARE);
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
