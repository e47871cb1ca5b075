/*
 * gkm_gram_bitslice.h -- launch interface of the hot kernel k_gram_bitslice (gkm_gram_bitslice.hip): its argument block,
 * the tunables that the host side has to agree with, and the instantiation table.
 */
#ifndef GKM_GRAM_BITSLICE_H
#define GKM_GRAM_BITSLICE_H

#include "gkm_internal.h"

struct BsArgs {
    const uint32_t *rowplanes;  /* [tile][plane 3][W][64] */
    const uint32_t *lane_mask;  /* [tile*64] bit rows at which a piece starts */
    const uint32_t *lane_piece; /* several pieces: [tile*64][MAX_PIECES] row slot | centre offset << 7 | owned windows << 20;
                                 * same length: [tile*64][2], word 0 = row slot and piece index as the origin word holds them */
    const int *tile_row, *tile_out, *tile_nrows, *tile_cbeg, *tile_cend; /* columns [cbeg, cend) per tile */
    const uint32_t *rowpk;      /* [tile*64 + lane][rpw] the lanes' positions, 2-bit packed (k_build_rowplanes) */
    const uint32_t *colpk;      /* [seq][pkw][strand] 2-bit packed strands, the two strands interleaved        */
    int rpw, pkw;               /* words per lane of rowpk (32: 128 bytes); words per strand of colpk */
    const uint32_t *postab;     /* [seq][ptw] every sequence's weights BY POSITION, L - 1 zero bytes either side and five wrap
                                 * bytes outside those (k_build_postab; POSTAB_PAD below): the column's weights in LDS */
    const uint32_t *wdc;        /* several-pieces variants: the distance weights CENTRED, byte wdc_centre + s = wd[|s|] (the
                                 * row side's weights) */
    int wdc_words, wdc_centre;
    int cap;                    /* same-length variant: l-mer windows a full lane owns (a multiple of 5: gkm_pack.h own_mult) */
    int ptw, ptw_stride;        /* ptw_stride = ptw, or 0 when every sequence has the same length: one table, which then stays
                                 * in the CUs' L1 instead of 0.3-0.6 KB per column coming from L2 (config 2: 1 % of the kernel) */
    const uint32_t *sb;
    int xw;
    const int *len;
    double c[GKM_MAXD1];
    GramOut out;
    /* Work items = (tile, column) pairs, one wavefront each, as a 1-D grid of exactly the pairs inside the
     * visited region (the 2-D (column, tile) grid launched as many empty blocks as real ones): item =
     * tile_soff[tile] + (j - cbeg[tile]), columns fastest.  Neighbouring blocks -- the waves resident on
     * a CU at the same time -- therefore work on the SAME row tile (its packed rows stay in the CU's L1)
     * and on DIFFERENT columns.  The opposite order (all tiles of a column next to each other on one XCD,
     * so that the column tables come from that XCD's L2) was measured: 91 instead of 85 ms on config 2
     * and 1000 instead of 509 ms on the peak-like set -- the waves of a CU then hit the same dense column
     * regions at the same moment and all wait on the hit path together. */
    int ntiles;
    /* raw Gram values leave the kernel tile-transposed: S[(tile_soff[tile] + j - cbeg) * NSLOT + row slot],
     * 64 consecutive doubles per store instruction; k_untile turns them into rows of G */
    double *S;
    const int64_t *tile_soff;
    /* The ORDER of the work items (round 4): entries (column chunk, tile), chunks outermost -- all tiles take the columns
     * [c C, (c + 1) C) before any takes the next chunk, so that a chunk's column tables (SB planes: 5-10 KB per column)
     * are streamed from HBM once per chunk and then come from the XCD's L2 for every further tile, instead of once per
     * tile.  Entry e covers columns [ent_j0[e], ent_j1[e]) of tile ent_tile[e] and starts at work item ent_off[e]; every
     * entry's item count is rounded up to a multiple of 8 (padding items return at once), so that column j of a chunk
     * has the same index mod 8 -- the same XCD under round-robin placement -- for every tile.  Inside an entry the
     * order is what it always was: one tile, consecutive columns.  nent = 0: the plain tile-major order. */
    int nent;
    const int64_t *ent_off;
    const int *ent_tile, *ent_j0, *ent_j1;
};

/* wave-uniform read-only words: address space 4 makes hipcc fetch them with scalar loads (s_load_dwordx*) into SGPRs */
typedef const uint32_t __attribute__((address_space(4))) * sgpr_words;

#ifndef GKM_BS_DU
#define GKM_BS_DU 4 /* shifts per refill of the column words.  Round 2's final kernel, same-run A/B, config 2 / gkmQC's
                       defaults / config 5: 1 -> 78.2 / 437.2 / 177.9 ms, 2 -> 77.0-77.5 / 436.8-437.6 / 176.7, 3 -> 76.7 / 437.0 /
                       176.2, 4 -> 76.4-76.8 / 436.7-438.6 / 175.6-175.8, 5, 6, 8 -> 79.1-79.4 / 452-454 / 176.5 */
#endif
constexpr int BS_DU = GKM_BS_DU; /* shifts per SB register refill (the SB tables are padded by it: ensure_sb) */

/* The positional weight table of a sequence with n l-mers (k_build_postab), bytes:
 *   [POSTAB_PAD: wt[n-5 .. n-1]] [L - 1 zeros] [wt[0 .. n-1]] [L - 1 zeros] [POSTAB_PAD: wt[0 .. 4]] [zeros to the end]
 * byte POSTAB_PAD + L - 1 + p = wt[p].  The zeros serve the windows that run over the strand's end (no l-mers); the five
 * bytes behind them the windows PAST the end -- position T + j is l-mer j again -- and the five bytes in front the same on
 * the reverse strand, whose weights are read downwards (wt_rc[q] = wt[n-1-q], libgkm.c:924). */
constexpr uint32_t POSTAB_PAD = 5;

typedef void (*bs_kernel_t)(const BsArgs);
/* the instantiation for W = 10 words per lane and packing variant pk (k_gram_bitslice's PK), or nullptr */
bs_kernel_t gkm_pick_bitslice(int pk, int L, int d);

#endif
