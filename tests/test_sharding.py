"""Row sharding + assembly of the Gram matrix across ranks, on CPU with the
gloo backend (world_size 2 and 3).  The per-rank kernel is replaced by the oracle here so the
test exercises exactly the plumbing bench.py uses: folded row blocks -> PACKED fixed-size slabs (row a = a + 1
doubles) -> all_gather_into_tensor -> row offsets -> normalisation."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import helpers


def test_folded_rows_partition_properties():
    from gkmqc_amd import sharding
    for n in (1, 7, 64, 400, 10000, 20001):
        for world in (1, 2, 3, 4, 8):
            seen = np.zeros(n, dtype=int)
            areas = []
            for r in range(world):
                rows, pad = sharding.folded_rows(n, world, r)
                assert (np.diff(rows) > 0).all(), "rows must be strictly ascending"
                assert len(rows) + pad == sharding.slab_rows(n, world)
                seen[rows] += 1
                areas.append(int((rows.astype(np.int64) + 1).sum()))
            assert (seen == 1).all(), "every row owned exactly once"
            if n >= 64 * world:
                assert max(areas) / (sum(areas) / world) < 1.05, "lower-triangle area balanced within 5 %"
            slot = sharding.gather_index(n, world)
            assert len(np.unique(slot)) == n


def test_chunked_layout_properties():
    from gkmqc_amd import sharding
    for n in (5, 63, 400, 10001):
        for world in (1, 2, 3, 8):
            for chunks in (1, 2, 4, 5):
                slot = sharding.chunked_gather_index(n, world, chunks)
                assert len(np.unique(slot)) == n
                seen = np.zeros(n, dtype=int)
                for r in range(world):
                    parts, pc = sharding.chunked_layout(n, world, r, chunks)
                    assert len(parts) == chunks and all(len(p) <= pc for p in parts)
                    for p in parts:
                        assert (np.diff(p) > 0).all()
                        seen[p] += 1
                assert (seen == 1).all()
                assert slot.max() < chunks * world * pc
                assert pc <= -(-sharding.slab_rows(n, world) // chunks) + sharding.CHUNK_GROUP, "little padding"


def test_cpp_layout_matches_python(built):
    """gkm_shard.h (used by the one-process multi-GPU entry gkmhip_gram_allgather) against
    gkmqc_amd/sharding.py (used by bench.py's one-process-per-GPU path)."""
    import ctypes
    from gkmqc_amd import sharding
    lib = ctypes.CDLL(os.path.join(helpers.ROOT, "gkmqc_amd", "csrc", "bitslice_cpu_probe.so"))
    for n in (1, 5, 63, 64, 65, 400, 1000, 10001):
        for world in (1, 2, 3, 8):
            for chunks in (1, 2, 4, 5):
                slot = np.zeros(n, dtype=np.int64)
                lib.shardprobe_gather_index(n, world, chunks, slot.ctypes.data_as(ctypes.c_void_p))
                assert (slot == sharding.chunked_gather_index(n, world, chunks)).all()
                for rank in range(world):
                    parts, pc = sharding.chunked_layout(n, world, rank, chunks)
                    assert lib.shardprobe_chunk_rows(n, world, chunks) == pc
                    for c in range(chunks):
                        buf = np.zeros(pc + 1, dtype=np.int32)
                        cnt = lib.shardprobe_part(n, world, rank, chunks, c, buf.ctypes.data_as(ctypes.c_void_p))
                        assert cnt == len(parts[c]) and (buf[:cnt] == parts[c]).all()
                # the packed slabs (row a = a + 1 doubles): slab size and where every row starts
                lib.shardprobe_packed_chunk_elems.restype = ctypes.c_longlong
                assert lib.shardprobe_packed_chunk_elems(n, world, chunks) == sharding.packed_chunk_elems(n, world, chunks)
                off = np.zeros(n, dtype=np.int64)
                lib.shardprobe_packed_gather_offsets(n, world, chunks, off.ctypes.data_as(ctypes.c_void_p))
                assert (off == sharding.packed_gather_offsets(n, world, chunks)).all()
            assert lib.shardprobe_auto_chunks(n, world) == sharding.auto_chunks(n, world)


def test_packed_layout_properties():
    """Packed slabs: every row's a + 1 cells lie inside its chunk's slab, no two rows overlap, every rank sends the
    same count, and the bytes are about half of what full-width rows cost."""
    from gkmqc_amd import sharding
    for n in (5, 63, 400, 2000, 10000):
        for world in (1, 2, 3, 8):
            for chunks in (1, 2, 4, 5):
                pe = sharding.packed_chunk_elems(n, world, chunks)
                off = sharding.packed_gather_offsets(n, world, chunks)
                order = np.argsort(off)
                ends = off[order] + order + 1            # row a occupies [off[a], off[a] + a + 1)
                assert (ends[:-1] <= off[order][1:]).all() and ends[-1] <= chunks * world * pe
                for r in range(world):
                    parts, _ = sharding.chunked_layout(n, world, r, chunks)
                    for c, p in enumerate(parts):
                        ro = sharding.packed_row_offsets(p)
                        assert ro[-1] <= pe, "a chunk fits its slab"
                        assert (off[p] == (c * world + r) * pe + ro[:-1]).all()
    # what crosses the links: half of round 3's full-width rows (700 -> ~350 MB per GPU for config 2 on 8 ranks)
    for n, world in ((10000, 8), (10000, 4), (10000, 2), (20000, 8)):
        chunks = sharding.auto_chunks(n, world)
        packed = sharding.allgather_bytes_per_rank(n, world, chunks)
        full = sharding.allgather_bytes_per_rank(n, world, chunks, packed=False)
        ideal = (world - 1) / world * n * n / 2 * 8
        assert packed < 0.53 * full and packed < 1.05 * ideal, (n, world, chunks, packed, full, ideal)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, raw_path, n, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, helpers.ROOT)
    from gkmqc_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    raw = np.load(raw_path)                     # raw G(a, j), lower triangle + diagonal
    chunks = 3                                  # same layout functions and call order as bench.py
    parts, _ = sharding.chunked_layout(n, world, rank, chunks)
    pe = sharding.packed_chunk_elems(n, world, chunks)
    slab = torch.full((chunks, pe), float("nan"), dtype=torch.float64)        # padding is never read
    gathered = torch.zeros((chunks, world * pe), dtype=torch.float64)
    pending = []
    for c in range(chunks):
        ro = sharding.packed_row_offsets(parts[c])
        for i, a in enumerate(parts[c]):                                     # this rank's rows only, j <= a
            slab[c, ro[i]:ro[i] + a + 1] = torch.from_numpy(raw[a, :a + 1])
        pending.append(dist.all_gather_into_tensor(gathered[c], slab[c], async_op=True))
    for w in pending:
        w.wait()
    off = sharding.packed_gather_offsets(n, world, chunks)
    flat = gathered.view(-1)
    full = torch.zeros((n, n), dtype=torch.float64)
    for a in range(n):
        full[a, :a + 1] = flat[off[a]:off[a] + a + 1]
    if rank == 0:
        np.save(out_path, full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_all_gather_assembly_gloo(built, tmp_path, world):
    from oracle import oracle as O
    cases, lens, npos = helpers.quirks_expected()
    c = cases[0]
    opt = O.make_opt(c["kernel_type"], c["L"], c["k"], c["d"], c["M"], c["H"], c["gamma"],
                     helpers.QUIRK_POS, helpers.QUIRK_NEG)
    r = O.gram(opt, want_profiles=True, nthreads=8)
    n = r["n"]
    cm = O.mismatch_weights(c["kernel_type"], c["L"], c["k"])[: c["d"] + 1]
    raw = np.tril((r["P"].astype(np.float64) * cm).sum(axis=2))
    raw_path, out_path = str(tmp_path / "raw.npy"), str(tmp_path / "full.npy")
    np.save(raw_path, raw)
    mp.spawn(_worker, args=(world, _free_port(), raw_path, n, out_path), nprocs=world, join=True)
    full = np.load(out_path)
    assert (full == raw).all(), "assembled matrix must equal the unsharded one bit for bit"
    sq = np.sqrt(np.diag(full))
    K = np.tril(full / np.outer(sq, sq), -1)
    assert helpers.max_rel_err(helpers.tril_pack(K), c["K"]) < 1e-12
