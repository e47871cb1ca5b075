"""Row sharding of the lower-triangular Gram computation across ranks.

Row a costs ~ (a+1) column sequences, so equal row blocks are unbalanced.  Folded pairing:
cut the rows into 2*G contiguous blocks; rank g owns blocks g and 2G-1-g -- equal row count
(so equal all-gather send counts) and near-equal area.  The assembled matrix is obtained by
an all-gather of the per-rank [rows_per_rank, N] slabs followed by an index permutation.
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
