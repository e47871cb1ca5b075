"""The bit-sliced lane program (gkmqc_amd/csrc/gkm_bitslice.h) run on the CPU through
bitslice_cpu_probe.so, against the oracle's brute-force profile.  This exercises the same
templates the HIP kernel instantiates: strided planes, SB tables, the sliding window-sum
tree, the <= d test, and hit consumption with weights."""
import ctypes
import os

import numpy as np
import pytest

from tests import helpers

CASES = [(10, 11, 3), (10, 10, 3), (10, 12, 4), (10, 8, 4), (10, 9, 4), (10, 12, 6), (10, 4, 2), (10, 2, 1),
         (10, 3, 0), (10, 5, 2), (10, 6, 3), (10, 7, 3), (5, 11, 3), (16, 11, 3), (3, 12, 4), (10, 12, 8),
         (10, 12, 12), (10, 11, 1), (10, 11, 5), (10, 12, 5), (10, 9, 5), (10, 10, 5), (10, 10, 6), (10, 11, 6), (10, 11, 7),
         (10, 12, 7)]


@pytest.fixture(scope="module")
def libs(built):
    from oracle import oracle as O
    probe = ctypes.CDLL(os.path.join(helpers.ROOT, "gkmqc_amd", "csrc", "bitslice_cpu_probe.so"))
    return O, probe


def _oracle_profile(O, t, L, d, A, B):
    opt = O.make_opt(t, L, L - d, d)
    P = np.zeros(d + 1, dtype=np.int32)
    O.lib().gkmo_profile(ctypes.byref(opt), A.ctypes.data_as(ctypes.c_void_p), len(A),
                         B.ctypes.data_as(ctypes.c_void_p), len(B), P.ctypes.data_as(ctypes.c_void_p))
    return P


def _dist_table(O, t, nmax):
    """wd[D] = positional weight at distance D from the centre l-mer (same for every length)."""
    dmax = nmax // 2 + 1
    wt = O.position_weights(t, 2 * dmax + 1)
    return np.ascontiguousarray(wt[dmax:])


def _probe_profile(O, probe, W, t, L, d, A, B):
    P = np.zeros(d + 1, dtype=np.int32)
    vp = ctypes.c_void_p
    wd = _dist_table(O, t, max(len(A), len(B))) if t in (4, 5) else None
    rc = probe.bsprobe_profile(W, L, d, A.ctypes.data_as(vp), len(A), B.ctypes.data_as(vp), len(B),
                               wd.ctypes.data_as(vp) if wd is not None else None, P.ctypes.data_as(vp))
    assert rc == 0
    return P


def test_distance_table_reproduces_every_length(libs):
    O, _ = libs
    for n in (1, 2, 3, 17, 139, 290, 291, 589, 2036):
        wd = _dist_table(O, 4, n)
        want = O.position_weights(4, n)
        got = np.array([wd[abs(n // 2 - p)] for p in range(n)], dtype=np.uint8)
        assert (got == want).all(), n


@pytest.mark.parametrize("W,L,d", CASES)
def test_lane_program_matches_oracle(libs, W, L, d):
    O, probe = libs
    rng = np.random.default_rng(1000 * W + 10 * L + d)
    shapes = [(300, 300), (L, L + 1), (700, 13 if L <= 13 else L), None, None]
    for trial, shp in enumerate(shapes):
        la, lb = shp if shp else (int(rng.integers(L, 500)), int(rng.integers(L, 500)))
        A = rng.integers(0, 4, la).astype(np.uint8)
        B = rng.integers(0, 4, lb).astype(np.uint8)
        if trial == 3:
            B = A.copy()                      # self profile (sqnorm path)
        if trial == 4:
            A[:] = 0
            B[:] = 0                          # poly-A: every window is a hit
        for t in (2, 4):
            assert (_oracle_profile(O, t, L, d, A, B) == _probe_profile(O, probe, W, t, L, d, A, B)).all(), \
                (trial, t, la, lb)


@pytest.mark.parametrize("W,L,d", [(10, 11, 3), (20, 11, 3), (10, 12, 4), (20, 12, 4), (20, 10, 3), (10, 6, 2), (20, 6, 2),
                                   (5, 11, 3)])
def test_packed_lanes_match_oracle(libs, W, L, d):
    """Several row sequences packed into lanes at bit-row granularity (gkm_pack.h): packing
    invariants (checked inside the probe) and per-row profiles against one column sequence."""
    O, probe = libs
    rng = np.random.default_rng(7 * W + 100 * L + d)
    lens = list(rng.integers(L, 700, 60)) + [L, L + 1, 2047, 300, 300, 300, 320, 321, 150, 150, 150, 149, 640, 641]
    seqs = [rng.integers(0, 4, int(n)).astype(np.uint8) for n in lens]
    seqs[5] = seqs[9].copy()
    seqs[6][:] = 0                                    # poly-A row
    codes = np.concatenate(seqs)
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    rows = np.arange(len(seqs), dtype=np.int32)
    vp = ctypes.c_void_p
    for t in (2, 4):
        wd = _dist_table(O, t, 2047) if t == 4 else None
        for col in (3, 6, 62, len(seqs) - 1):
            P = np.zeros((len(seqs), d + 1), dtype=np.int32)
            used = ctypes.c_int(0)
            rc = probe.bsprobe_profile_packed(W, L, d, codes.ctypes.data_as(vp), off.ctypes.data_as(vp),
                                              rows.ctypes.data_as(vp), len(rows), col,
                                              wd.ctypes.data_as(vp) if wd is not None else None,
                                              P.ctypes.data_as(vp), ctypes.byref(used))
            assert rc == 0, "packing invariant %d violated" % rc
            for i in (0, 5, 6, 9, 20, 41, 60, 61, 62, 63, 66, 67, 70, 73):
                assert (P[i] == _oracle_profile(O, t, L, d, seqs[i], seqs[col])).all(), (t, col, i)
    # the packing is dense: lanes used stay close to the information-theoretic minimum
    need = sum(-(-len(s) // W) for s in seqs) / 32.0
    assert used.value <= 1.25 * need + 2


def test_tiles_are_closed_at_jumps_in_the_row_list_where_that_saves_work_items(libs):
    """A multi-GPU rank's rows are two folded blocks (gkm_shard.h): rows 0..624 and 9375..9999 for rank 0 of an 8-way split
    of 10 000.  Packed in one go, the 49 low rows left over after nine full tiles share a tile with 15 high rows and ride
    along through 8 765 columns they do not need (measured: 109 660 work items instead of 100 760,
    profiles/r5_small_launch_blocks.txt); pack_rows(split_jump=64) closes the tile at the jump.  Adjacent blocks (rank 7)
    have no jump and must pack as before; a list with small regular gaps (every third row) must not be split."""
    import ctypes
    probe = libs[1] if isinstance(libs, tuple) else libs
    fn = probe.packprobe_triangle_items
    fn.restype = ctypes.c_longlong

    def items(rows, split):
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        nwin = np.full(len(rows), 290, dtype=np.int32)
        nt = ctypes.c_int(0)
        got = fn(rows.ctypes.data_as(ctypes.c_void_p), nwin.ctypes.data_as(ctypes.c_void_p), len(rows), 10, 11, 64, split,
                 ctypes.byref(nt))
        assert got > 0
        return got, nt.value

    rank0 = np.concatenate([np.arange(0, 625), np.arange(9375, 10000)])
    mixed, t0 = items(rank0, 0)
    split, t1 = items(rank0, 64)
    assert mixed == 109660 and split == 100760 and t1 == t0     # the same 20 tiles, 8.1 % fewer work items
    rank7 = np.arange(4375, 5625)
    assert items(rank7, 0) == items(rank7, 64)
    thirds = np.arange(0, 3000, 3)
    assert items(thirds, 0) == items(thirds, 64)
