/*
 * gkm_shard.h -- which rank computes which rows of the lower-triangular Gram matrix
 * (SURVEY.md §8(e)); plain C++ so that the layout is unit-tested on the CPU against its Python
 * twin gkmqc_amd/sharding.py (tests/test_sharding.py), which the one-process-per-GPU path
 * of bench.py uses.
 *
 * Row a of the lower triangle costs ~ (a + 1) column sequences, so equal row blocks are unbalanced.
 * Folded pairing: the rows are cut into 2 G contiguous blocks of `blk` rows, rank g owns blocks g and
 * 2G-1-g: equal row counts (an all-gather needs equal send counts) and equal area within ~1 %.
 * A rank's rows are dealt to `chunks` sub-lists in groups of 64 consecutive rows (one tile of the
 * Gram kernel), round robin, so that the all-gather of chunk c can overlap the kernel of chunk c+1
 * and every chunk carries the same mix of cheap and expensive rows.
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

/* rows per chunk slab: the largest chunk of a full slab */
inline int chunk_rows(int n, int world, int chunks)
{
    const int per = slab_rows(n, world);
    std::vector<int> cnt((size_t)chunks, 0);
    for (int i = 0; i < per; i++) cnt[(size_t)((i / CHUNK_GROUP) % chunks)]++;
    int pc = 0;
    for (int c : cnt) pc = c > pc ? c : pc;
    return pc;
}

inline std::vector<std::vector<int>> chunked_layout(int n, int world, int rank, int chunks)
{
    const std::vector<int> rows = folded_rows(n, world, rank);
    std::vector<std::vector<int>> parts((size_t)chunks);
    for (size_t i = 0; i < rows.size(); i++) parts[(i / CHUNK_GROUP) % (size_t)chunks].push_back(rows[i]);
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

} /* namespace gkmshard */
#endif
