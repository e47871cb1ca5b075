/*
 * gkm_shard.h -- which rank computes which rows of the lower-triangular Gram matrix
 * (SURVEY.md §8(e)); plain C++ so that the layout is unit-tested on the CPU against its Python
 * twin gkmqc_amd/sharding.py (tests/test_sharding.py), which the one-process-per-GPU path
 * of bench.py uses.
 *
 * Row a of the lower triangle costs ~ (a + 1) column sequences, so equal row blocks are unbalanced.
 * Folded pairing: the rows are cut into 2 G contiguous blocks of `blk` rows, rank g owns blocks g and
 * 2G-1-g: equal row counts (an all-gather needs equal send counts) and equal area within ~1 %.
 * The rows of each of a rank's two blocks are dealt to `chunks` sub-lists in groups of 64 consecutive rows (one tile of
 * the Gram kernel; a group never reaches across the two blocks), round robin, so that the all-gather of chunk c can
 * overlap the kernel of chunk c+1 and every chunk carries the same mix of cheap and expensive rows.
 *
 * PACKED slabs (round 4).  Only the cells j <= a of row a are ever computed or read (the consumer is the
 * un-permute + normalise pass, k_assemble_normalize), so a chunk's slab holds row a as a + 1 doubles, the rows of
 * the chunk back to back in ascending order -- not n doubles per row.  The folded pairing makes the packed size of
 * every rank's rows the same, blk^2 (2G - 1) + blk (blk + 1) doubles when n = 2 G blk, and chunk by chunk within a
 * few rows of each other; every chunk slab is padded to the largest one over all (rank, chunk), which every rank
 * computes alike, so the equal-count rule of an all-gather holds.  The collective then moves n^2 / (2G) doubles per
 * rank instead of n^2 / G: half (config 2 on 8 ranks: ~350 MB received per GPU per matrix instead of 700).
 */
#ifndef GKM_SHARD_H
#define GKM_SHARD_H

#include <stdint.h>

#include <vector>

namespace gkmshard {

constexpr int CHUNK_GROUP = 64;

inline int block_rows(int n, int world) { return (n + 2 * world - 1) / (2 * world); }
inline int slab_rows(int n, int world) { return 2 * block_rows(n, world); }

/* ascending rows of `rank` */
inline std::vector<int> folded_rows(int n, int world, int rank)
{
    const int blk = block_rows(n, world);
    std::vector<int> rows;
    auto add = [&](int b) {
        const long lo = (long)b * blk, hi = (long)(b + 1) * blk;
        for (long r = lo; r < hi && r < n; r++) rows.push_back((int)r);
    };
    add(rank);
    if (2 * world - 1 - rank != rank) add(2 * world - 1 - rank);
    return rows;
}

/* chunk of each position of a row list made of blocks of `lens` rows: 64-row groups dealt round robin, a group never
 * reaching across two blocks.  (Round 5: a group that held the last 49 rows of a rank's low block and the first 15 of its
 * high block became ONE tile of the Gram kernel, and the 49 low rows rode along through the thousands of columns only the
 * high rows need: 8.8 % more work items for rank 0 of an 8-way split of config 2, profiles/r5_small_launch_blocks.txt.) */
inline std::vector<int> chunk_of_position(const std::vector<int> &lens, int chunks)
{
    std::vector<int> which;
    int g0 = 0;
    for (int ln : lens) {
        for (int i = 0; i < ln; i++) which.push_back((g0 + i / CHUNK_GROUP) % chunks);
        g0 += (ln + CHUNK_GROUP - 1) / CHUNK_GROUP;
    }
    return which;
}

/* rows per chunk slab: the largest chunk of a full slab (two whole blocks) */
inline int chunk_rows(int n, int world, int chunks)
{
    const int blk = block_rows(n, world);
    std::vector<int> cnt((size_t)chunks, 0);
    for (int c : chunk_of_position({blk, blk}, chunks)) cnt[(size_t)c]++;
    int pc = 0;
    for (int c : cnt) pc = c > pc ? c : pc;
    return pc;
}

inline std::vector<std::vector<int>> chunked_layout(int n, int world, int rank, int chunks)
{
    const std::vector<int> rows = folded_rows(n, world, rank);
    const int blk = block_rows(n, world), hi = 2 * world - 1 - rank;
    auto len_of = [&](int b) {
        const long lo = (long)b * blk, up = (long)(b + 1) * blk;
        return (int)((up < n ? up : n) - (lo < n ? lo : n));
    };
    std::vector<int> lens = {len_of(rank)};
    if (hi != rank) lens.push_back(len_of(hi));
    const std::vector<int> which = chunk_of_position(lens, chunks);
    std::vector<std::vector<int>> parts((size_t)chunks);
    for (size_t i = 0; i < rows.size(); i++) parts[(size_t)which[i]].push_back(rows[i]);
    return parts;
}

/* slot_of_row[a]: row of matrix row a inside the concatenation over chunks c of the all-gathered
 * [world * pc, n] slabs */
inline std::vector<int64_t> chunked_gather_index(int n, int world, int chunks)
{
    const int pc = chunk_rows(n, world, chunks);
    std::vector<int64_t> slot((size_t)n, -1);
    for (int g = 0; g < world; g++) {
        const std::vector<std::vector<int>> parts = chunked_layout(n, world, g, chunks);
        for (int c = 0; c < chunks; c++)
            for (size_t i = 0; i < parts[(size_t)c].size(); i++)
                slot[(size_t)parts[(size_t)c][i]] = ((int64_t)c * world + g) * pc + (int64_t)i;
    }
    return slot;
}

/* ---- packed slabs ---- */
/* offset of each row of one chunk inside its packed slab (rows back to back, row a = a + 1 doubles); the entry
 * behind the last row is the packed size of the chunk */
inline std::vector<int64_t> packed_row_offsets(const std::vector<int> &rows)
{
    std::vector<int64_t> off(rows.size() + 1, 0);
    for (size_t i = 0; i < rows.size(); i++) off[i + 1] = off[i] + (int64_t)rows[i] + 1;
    return off;
}

/* doubles per chunk slab: the largest packed chunk over all ranks and chunks (the all-gather's send count) */
inline int64_t packed_chunk_elems(int n, int world, int chunks)
{
    int64_t pe = 1;
    for (int g = 0; g < world; g++) {
        const std::vector<std::vector<int>> parts = chunked_layout(n, world, g, chunks);
        for (const std::vector<int> &p : parts) {
            int64_t e = 0;
            for (int a : p) e += (int64_t)a + 1;
            pe = e > pe ? e : pe;
        }
    }
    return pe;
}

/* Chunks per rank when the caller does not say.  A chunk more costs a launch more -- measured in round 5 with one rank
 * alone on a GPU (tools/rank_alone.py, 8-way split of n = 10 000): 10.58 / 10.54 / 10.85 / 11.05 ms per rank with 1 / 2 /
 * 3 / 4 chunks, i.e. ~0.25 ms per further launch beside a kernel of ~9 ms -- and hides 1 / chunks more of the transfer
 * (~0.36 GB received per rank, 1-5 ms over xGMI): 2, at most 3, for 2 to 8 ranks.  Second concern: 64-row groups dealt
 * round robin leave some chunk with one expensive group more than the others, and every chunk slab is padded to the
 * largest (21 % of the bytes with 4 chunks at n = 10 000 on 8 ranks, 2 % with 2).  So: 2, unless 3 pads at least 3 % less,
 * else 4 likewise.  (GKM_BENCH_CHUNKS and the `chunks` argument override.) */
inline int auto_chunks(int n, int world)
{
    if (world <= 1) return 1;
    const int cand[3] = {2, 3, 4};
    double padded[3];
    for (int i = 0; i < 3; i++) padded[i] = (double)cand[i] * (double)packed_chunk_elems(n, world, cand[i]);
    int best = 0;
    for (int i = 1; i < 3; i++)
        if (padded[i] < 0.97 * padded[best]) best = i;
    return cand[best];
}

/* offset[a]: where matrix row a starts inside the concatenation over chunks c of the all-gathered
 * [world][packed_chunk_elems] slabs */
inline std::vector<int64_t> packed_gather_offsets(int n, int world, int chunks)
{
    const int64_t pe = packed_chunk_elems(n, world, chunks);
    std::vector<int64_t> off((size_t)n, -1);
    for (int g = 0; g < world; g++) {
        const std::vector<std::vector<int>> parts = chunked_layout(n, world, g, chunks);
        for (int c = 0; c < chunks; c++) {
            const std::vector<int64_t> ro = packed_row_offsets(parts[(size_t)c]);
            for (size_t i = 0; i < parts[(size_t)c].size(); i++)
                off[(size_t)parts[(size_t)c][i]] = ((int64_t)c * world + g) * pe + ro[i];
        }
    }
    return off;
}

} /* namespace gkmshard */
#endif
