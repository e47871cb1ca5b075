/*
 * gkm_pack.h -- host-side packing of row sequences into the lanes of the bit-sliced kernel.
 *
 * A lane of the kernel holds 32 "bit rows" of W consecutive sequence positions each (bit b of
 * word w <-> lane position b*W + w, gkm_bitslice.h).  A PIECE is a run of bit rows of one lane
 * given to one row sequence: bit rows [b0, b0+nb) hold sequence positions p0 .. p0+nb*W-1 and
 * own the `cnt` l-mer windows that start at p0 .. p0+cnt-1 (a window must lie inside its
 * piece, so cnt <= nb*W - (L-1); the next piece of the same sequence starts at p0+cnt and
 * therefore overlaps by L-1 bases).  Packing several pieces per lane keeps the lanes full for
 * any length distribution (one 300-bp row needs 15 of the 32 bit rows at W = 20).
 *
 * All pieces of a row live in ONE tile (64 lanes) so that the kernel can accumulate a row's
 * mismatch profile in LDS by "row slot".  Plain C++ (no HIP) so the packing is unit-tested on
 * the CPU (bitslice_cpu_probe.cpp).
 */
#ifndef GKM_PACK_H
#define GKM_PACK_H

#include <stdint.h>

#include <vector>

namespace gkmpack {

constexpr int LANES = 64;
constexpr int MAX_PIECES = 4;   /* pieces per lane */
constexpr int MAX_ROWS = 128;   /* row slots per tile (a caller may ask pack_rows for fewer) */

struct Piece {
    int32_t lane;  /* global lane index = tile * 64 + lane in tile */
    int32_t b0;    /* first bit row */
    int32_t nb;    /* number of bit rows */
    int32_t slot;  /* row slot inside the tile */
    int32_t row;   /* sequence index */
    int32_t p0;    /* first sequence position (= first window start) of the piece */
    int32_t cnt;   /* window starts owned by the piece */
};

struct Packing {
    int W = 0, L = 0, ntiles = 0;
    std::vector<Piece> pieces;          /* sorted by lane, then b0 */
    std::vector<int32_t> tile_row;      /* [ntiles * MAX_ROWS] sequence index of each row slot (-1 = unused) */
    std::vector<int32_t> tile_out;      /* [ntiles * MAX_ROWS] output row (position in the caller's row list) */
    std::vector<int32_t> tile_nrows;    /* [ntiles] */
    std::vector<int32_t> tile_amax;     /* [ntiles] largest sequence index in the tile */
    long lanes_used = 0;
};

/* rows: ascending sequence indices; nwin[i]: l-mer windows of rows[i] (>= 1).
 * Greedy first-fit in row order (the order matters: a tile only visits columns j <= its largest
 * row, so tiles should hold neighbouring rows).
 * split_jump: a JUMP of at least that many rows in the list (the folded row blocks of a multi-GPU rank: rows 0..624, then
 * 9375..9999) closes the current tile.  Every work item of a tile is 64 lanes x one column for ALL columns up to the
 * tile's largest row, so low rows that share a tile with high ones ride along through thousands of columns they do not
 * need: rank 0 of an 8-way split of config 2 ran 109 660 work items instead of 100 760 (round 5,
 * profiles/r5_small_launch_blocks.txt).  Whether splitting pays depends on the list (it can also cost a tile); the caller
 * packs both ways and keeps the one with fewer work items (triangle_items).
 * own_mult: a piece that does not finish its row owns a multiple of that many windows (the same-length kernel variant
 * evaluates GROUPS of five consecutive lane positions at once and needs every group owned whole or not at all). */
inline Packing pack_rows(const int *rows, const int *nwin, int nrows, int W, int L, int max_rows = MAX_ROWS, int split_jump = 0,
                         int own_mult = 1)
{
    Packing P;
    P.W = W;
    P.L = L;
    const int min_bits = (L + W - 1) / W;       /* bit rows needed for a single window */
    int tile = 0, lane = 0, freeb = 32, npl = 0; /* cursor: lane in tile, free bit rows, pieces in lane */
    int tile_rows = 0;
    auto new_lane = [&]() { lane++; freeb = 32; npl = 0; };
    auto close_tile = [&]() {
        P.tile_nrows.push_back(tile_rows);
        tile++;
        lane = 0; freeb = 32; npl = 0; tile_rows = 0;
    };
    auto open_tile_storage = [&]() {
        if ((int)P.tile_row.size() < (tile + 1) * MAX_ROWS) {
            P.tile_row.resize((size_t)(tile + 1) * MAX_ROWS, -1);
            P.tile_out.resize((size_t)(tile + 1) * MAX_ROWS, 0);
        }
    };
    for (int i = 0; i < nrows; i++) {
        /* split_jump > 0: a jump of at least that many rows in the list closes the tile (see the function's header) */
        if (split_jump > 0 && i > 0 && tile_rows > 0 && rows[i] - rows[i - 1] >= split_jump) close_tile();
        for (int attempt = 0; attempt < 2; attempt++) {
            if (tile_rows >= max_rows) close_tile();
            open_tile_storage();
            /* remember the cursor so the row can be undone if it does not fit in this tile */
            const size_t mark = P.pieces.size();
            const int s_lane = lane, s_free = freeb, s_npl = npl;
            int remaining = nwin[i], p0 = 0;
            bool fits = true;
            while (remaining > 0) {
                const int need = (remaining + L - 1 + W - 1) / W;
                /* do not split off a sliver: every split costs L-1 overlapping bases and makes the
                 * tile take the several-pieces-per-lane path; a partial piece must hold >= 3W windows
                 * (config 2 with 10-window slivers in the 2 spare bit rows: 119 ms instead of 113) */
                const int want = need < freeb ? need : freeb;
                const int room = want * W - (L - 1); /* windows a piece of `want` bit rows can own */
                const int cnt_here = room >= remaining ? remaining : room / own_mult * own_mult;
                if (freeb < min_bits || npl >= MAX_PIECES || (want < need && cnt_here < 3 * W)) {
                    new_lane();
                    if (lane >= LANES) { fits = false; break; }
                    continue;
                }
                Piece pc;
                pc.lane = tile * LANES + lane;
                pc.b0 = 32 - freeb;
                pc.nb = want;
                pc.slot = tile_rows;
                pc.row = rows[i];
                pc.p0 = p0;
                pc.cnt = cnt_here;
                P.pieces.push_back(pc);
                freeb -= want;
                npl++;
                p0 += cnt_here;
                remaining -= cnt_here;
            }
            if (fits) {
                P.tile_row[(size_t)tile * MAX_ROWS + tile_rows] = rows[i];
                P.tile_out[(size_t)tile * MAX_ROWS + tile_rows] = i;
                tile_rows++;
                break;
            }
            /* undo and retry in a fresh tile (a row needs at most 7 lanes, it always fits there) */
            P.pieces.resize(mark);
            lane = s_lane; freeb = s_free; npl = s_npl;
            close_tile();
        }
    }
    if (tile_rows > 0 || P.tile_nrows.empty()) close_tile();
    P.ntiles = (int)P.tile_nrows.size();
    P.tile_row.resize((size_t)P.ntiles * MAX_ROWS, -1);
    P.tile_out.resize((size_t)P.ntiles * MAX_ROWS, 0);
    P.tile_amax.assign((size_t)P.ntiles, -1);
    long lanes = 0;
    int last_lane = -1;
    for (const Piece &pc : P.pieces) {
        int &am = P.tile_amax[(size_t)(pc.lane / LANES)];
        if (pc.row > am) am = pc.row;
        if (pc.lane != last_lane) { lanes++; last_lane = pc.lane; }
    }
    P.lanes_used = lanes;
    return P;
}

/* work items of a launch that visits, for every tile, the columns 0 .. its largest row (the triangle) */
inline long long triangle_items(const Packing &P)
{
    long long items = 0;
    for (int t = 0; t < P.ntiles; t++) items += (long long)P.tile_amax[(size_t)t] + 1;
    return items;
}

/* relative cost of running the kernel with this packing: lanes x (per-word cost x W + fixed
 * per-shift cost), in VALU instructions per shift (18 per word, ~25 per shift: DESIGN.md §5) */
inline double packing_cost(const Packing &P) { return (double)P.ntiles * LANES * (18.0 * P.W + 25.0); }

} /* namespace gkmpack */
#endif
