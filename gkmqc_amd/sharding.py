"""Row sharding of the lower-triangular Gram computation across ranks.

Row a costs ~ (a+1) column sequences, so equal row blocks are unbalanced.  Folded pairing:
cut the rows into 2*G contiguous blocks; rank g owns blocks g and 2G-1-g -- equal row count
(so equal all-gather send counts) and near-equal area.  The assembled matrix is obtained by
an all-gather of the per-rank slabs followed by an index permutation.

Packed slabs (round 4): only the cells j <= a of row a are computed and read, so a chunk's slab holds row a as
a + 1 doubles, rows back to back; the folded pairing makes the packed size the same on every rank (and, chunk by
chunk, within a few rows), every chunk slab is padded to the largest one -- the all-gather moves n^2 / (2G)
doubles per rank instead of n^2 / G.  Twin of gkmqc_amd/csrc/gkm_shard.h (tests/test_sharding.py).
"""
import numpy as np


def folded_rows(n, world_size, rank):
    """Ascending row indices owned by `rank`; every rank gets exactly ceil(n / (2G)) * 2
    slots, the surplus ones are returned as -1 padding at the end (never computed)."""
    g2 = 2 * world_size
    blk = -(-n // g2)
    lo = np.arange(rank * blk, min((rank + 1) * blk, n))
    hi_blk = g2 - 1 - rank
    hi = np.arange(min(hi_blk * blk, n), min((hi_blk + 1) * blk, n))
    rows = np.concatenate([lo, hi]).astype(np.int32)
    pad = 2 * blk - len(rows)
    return rows, pad


def slab_rows(n, world_size):
    """rows_per_rank used for the all-gather slab of every rank."""
    return 2 * (-(-n // (2 * world_size)))


def gather_index(n, world_size):
    """perm such that full[perm[r]] = gathered[r] for the valid slots; returns
    (slot_index_of_row[n]) : row a of the matrix lives at gathered slot slot_of_row[a]."""
    per = slab_rows(n, world_size)
    slot_of_row = np.full(n, -1, dtype=np.int64)
    for g in range(world_size):
        rows, _ = folded_rows(n, world_size, g)
        slot_of_row[rows] = g * per + np.arange(len(rows))
    assert (slot_of_row >= 0).all()
    return slot_of_row


CHUNK_GROUP = 64    # rows dealt to a chunk at a time: one 64-lane tile of the Gram kernel for fixed-length data


def _chunk_of_position(block_lens, chunks):
    """Chunk of each position of a rank's row list, which is made of blocks of `block_lens` rows: 64-row groups dealt
    round robin, a group never reaching across two blocks.  (Round 5: a group that held the last 49 rows of the rank's low
    block and the first 15 of its high block became ONE tile of the Gram kernel, and the 49 low rows rode along through
    the thousands of columns only the high rows need -- 8.8 % more work items for rank 0 of an 8-way split of config 2.)"""
    which, g0 = [], 0
    for ln in block_lens:
        which.append((g0 + np.arange(ln) // CHUNK_GROUP) % chunks)
        g0 += -(-ln // CHUNK_GROUP)
    return np.concatenate(which) if which else np.zeros(0, dtype=np.int64)


def _block_lens(n, world_size, rank):
    g2 = 2 * world_size
    blk = -(-n // g2)
    hi_blk = g2 - 1 - rank
    return [max(0, min((rank + 1) * blk, n) - rank * blk), max(0, min((hi_blk + 1) * blk, n) - min(hi_blk * blk, n))]


def chunked_layout(n, world_size, rank, chunks):
    """Split this rank's rows into `chunks` sub-lists.  The rows of each of the rank's two blocks are dealt to the chunks
    in groups of CHUNK_GROUP CONSECUTIVE rows, round robin: every chunk carries the same mix of cheap
    and expensive rows, and the rows a tile of the kernel holds stay neighbours (a tile visits all
    columns up to its largest row, so a tile of rows 4 apart would do 3.8 % more work on the
    headline problem).  The Gram kernel runs once per chunk and each chunk's slab is all-gathered on
    its own, which lets the collective of chunk c overlap the kernel of chunk c+1.
    Returns (list of ascending row arrays, rows_per_chunk = slab height of every chunk)."""
    rows, _ = folded_rows(n, world_size, rank)
    blk = -(-n // (2 * world_size))
    which = _chunk_of_position(_block_lens(n, world_size, rank), chunks)
    pc = int(np.bincount(_chunk_of_position([blk, blk], chunks), minlength=chunks).max())
    return [rows[which == c] for c in range(chunks)], pc


def chunked_gather_index(n, world_size, chunks):
    """slot_of_row[a] = row index of matrix row a inside the concatenation over chunks c of the
    all-gathered tensors [world_size * rows_per_chunk, n]."""
    slot_of_row = np.full(n, -1, dtype=np.int64)
    for g in range(world_size):
        parts, pc = chunked_layout(n, world_size, g, chunks)
        for c, r in enumerate(parts):
            slot_of_row[r] = (c * world_size + g) * pc + np.arange(len(r))
    assert (slot_of_row >= 0).all()
    return slot_of_row


def packed_row_offsets(rows):
    """Offset of each row of one chunk inside its packed slab (row a = a + 1 doubles, rows back to back); one
    entry more than rows: the packed size of the chunk."""
    off = np.zeros(len(rows) + 1, dtype=np.int64)
    np.cumsum(np.asarray(rows, dtype=np.int64) + 1, out=off[1:])
    return off


def packed_chunk_elems(n, world_size, chunks):
    """Doubles per chunk slab: the largest packed chunk over all ranks and chunks (the all-gather's send count)."""
    pe = 1
    for g in range(world_size):
        parts, _ = chunked_layout(n, world_size, g, chunks)
        for p in parts:
            pe = max(pe, int((p.astype(np.int64) + 1).sum()))
    return pe


def auto_chunks(n, world_size):
    """Chunks per rank when the caller does not say (twin of gkm_shard.h auto_chunks): 2, unless 3 (or then 4) pads the
    slabs at least 3 % less.  A chunk more costs a launch more (~0.7 ms of ramp and drain beside a kernel of 75 ms /
    ranks) and hides 1 / chunks more of a transfer of 1-5 ms; 64-row groups dealt round robin leave some chunk with one
    expensive group more than the others, and every slab is padded to the largest."""
    if world_size <= 1:
        return 1
    cand = (2, 3, 4)
    padded = [c * packed_chunk_elems(n, world_size, c) for c in cand]
    best = 0
    for i in (1, 2):
        if padded[i] < 0.97 * padded[best]:
            best = i
    return cand[best]


def packed_gather_offsets(n, world_size, chunks):
    """offset[a]: where matrix row a starts inside the concatenation over chunks c of the all-gathered
    [world_size][packed_chunk_elems] slabs."""
    pe = packed_chunk_elems(n, world_size, chunks)
    off = np.full(n, -1, dtype=np.int64)
    for g in range(world_size):
        parts, _ = chunked_layout(n, world_size, g, chunks)
        for c, p in enumerate(parts):
            off[p] = (c * world_size + g) * pe + packed_row_offsets(p)[:-1]
    assert (off >= 0).all()
    return off


def allgather_bytes_per_rank(n, world_size, chunks, packed=True):
    """Bytes one rank RECEIVES from its peers per matrix (what crosses xGMI into each GPU)."""
    if world_size <= 1:
        return 0
    if packed:
        per_chunk = packed_chunk_elems(n, world_size, chunks)
    else:
        per_chunk = chunked_layout(n, world_size, 0, chunks)[1] * n
    return int(chunks * (world_size - 1) * per_chunk * 8)
