"""ctypes binding of the C ABI exported by ``bin/gkmkern_pylib.so``.

Two layers, both declared under ``include/``:
  * ``gkm_main_pywrapper`` -- the drop-in boundary (include/gkmkern_pylib.h), bound exactly
    as the reference binds it (scripts/gkmsvm.py:48-61,85-88);
  * ``gkmhip_*`` -- the device layer (include/gkm_hip.h) used by bench.py and the GPU tests
    with device memory and streams supplied by PyTorch-ROCm.

There is no CPU compute path here: if the shared object is missing the import of the
library fails loudly, and on a box without a GPU the device calls return errors.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(HERE, "bin")
LIB_NAME = "gkmkern_pylib.so"

KERNEL_AUTO, KERNEL_DIRECT, KERNEL_BITSLICE = 0, 1, 2


class gkmOpt(ctypes.Structure):
    """Same layout as the reference's gkmOpt (src/libgkm.h:149-161)."""
    _fields_ = (
        ("kernel_type", ctypes.c_int), ("L", ctypes.c_int), ("k", ctypes.c_int), ("d", ctypes.c_int),
        ("M", ctypes.c_uint8), ("H", ctypes.c_double), ("gamma", ctypes.c_double),
        ("posfile", ctypes.c_char_p), ("negfile", ctypes.c_char_p),
        ("nthreads", ctypes.c_int), ("verbosity", ctypes.c_int),
    )


class GkmError(RuntimeError):
    pass


_lib = None


def lib_path():
    # GKM_LIB_PATH: load another build of the same library (A/B timing of kernel variants)
    return os.environ.get("GKM_LIB_PATH") or os.path.join(LIB_DIR, LIB_NAME)


def load():
    """Load (once) and prototype the shared object.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own HIP runtime: if this library (linked against /opt/rocm) were
    # loaded first, torch would afterwards find "no HIP GPUs".  Import torch first when it exists
    # so that one runtime serves both.  (The reference's own caller does not use torch at all.)
    import importlib.util
    if importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401
    path = lib_path()
    if not os.path.exists(path):
        raise GkmError("%s is missing: build it with `make -C gkmqc_amd/csrc` "
                       "(or `python -c 'import __graft_entry__ as g; g.build()'`)" % path)
    L = ctypes.CDLL(path)
    vp, i32, i64, dbl = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double
    L.gkm_main_pywrapper.restype = i32
    L.gkm_main_pywrapper.argtypes = (ctypes.POINTER(gkmOpt), vp, vp)
    L.gkm_check_parameter_values.restype = ctypes.c_char_p
    L.gkm_check_parameter_values.argtypes = (i32, i32, i32, i32)
    L.gkm_mismatch_weights.restype = i32
    L.gkm_mismatch_weights.argtypes = (i32, i32, i32, vp)
    L.gkm_position_weights.restype = None
    L.gkm_position_weights.argtypes = (i32, i32, ctypes.c_uint8, dbl, vp)
    L.gkm_problem_read.restype = vp
    L.gkm_problem_read.argtypes = (ctypes.c_char_p, ctypes.c_char_p)
    L.gkm_problem_free.restype = None
    L.gkm_problem_free.argtypes = (vp,)
    for name in ("gkm_problem_size", "gkm_problem_npos"):
        getattr(L, name).restype = i32
        getattr(L, name).argtypes = (vp,)
    L.gkm_problem_seqlen.restype = i32
    L.gkm_problem_seqlen.argtypes = (vp, i32)
    L.gkm_problem_codes.restype = ctypes.POINTER(ctypes.c_uint8)
    L.gkm_problem_codes.argtypes = (vp, i32)
    L.gkm_problem_offsets.restype = ctypes.POINTER(ctypes.c_int64)
    L.gkm_problem_offsets.argtypes = (vp,)
    L.gkm_problem_all_codes.restype = ctypes.POINTER(ctypes.c_uint8)
    L.gkm_problem_all_codes.argtypes = (vp,)
    for name in ("gkm_problem_invalid_chars", "gkm_problem_truncated"):
        getattr(L, name).restype = ctypes.c_long
        getattr(L, name).argtypes = (vp,)

    L.gkmhip_last_error.restype = ctypes.c_char_p
    L.gkmhip_device_count.restype = i32
    L.gkmhip_create.restype = vp
    L.gkmhip_create.argtypes = (i32, i32, i32, vp, i32, dbl)
    L.gkmhip_destroy.restype = None
    L.gkmhip_destroy.argtypes = (vp,)
    L.gkmhip_set_kernel.restype = i32
    L.gkmhip_set_kernel.argtypes = (vp, i32)
    L.gkmhip_set_scratch_slot.restype = i32
    L.gkmhip_set_scratch_slot.argtypes = (vp, i32)
    L.gkmhip_set_sequences.restype = i32
    L.gkmhip_set_sequences.argtypes = (vp, i32, vp, vp, vp, i32, vp)
    L.gkmhip_gram_rows.restype = i32
    L.gkmhip_gram_rows.argtypes = (vp, vp, i32, i32, vp, i64, vp, i64, vp)
    if hasattr(L, "gkmhip_gram_rows_packed"):   # (older builds loaded through GKM_LIB_PATH for A/B timing lack it)
        L.gkmhip_gram_rows_packed.restype = i32
        L.gkmhip_gram_rows_packed.argtypes = (vp, vp, i32, vp, vp, vp)
        L.gkmhip_allgather_bytes_per_rank.restype = ctypes.c_longlong
    L.gkmhip_gram_rows_full.restype = i32
    L.gkmhip_gram_rows_full.argtypes = (vp, vp, i32, i32, vp, i64, vp)
    L.gkmhip_self_norms.restype = i32
    L.gkmhip_self_norms.argtypes = (vp, vp, vp)
    L.gkmhip_normalize_rows_full.restype = i32
    L.gkmhip_normalize_rows_full.argtypes = (vp, vp, i32, i32, vp, i64, vp, vp)
    L.gkmhip_normalize.restype = i32
    L.gkmhip_normalize.argtypes = (vp, vp, i64, vp, i32, vp)
    L.gkmhip_malloc.restype = vp
    L.gkmhip_malloc.argtypes = (i32, ctypes.c_size_t)
    L.gkmhip_free.restype = None
    L.gkmhip_free.argtypes = (vp,)
    L.gkmhip_memcpy_d2h.restype = i32
    L.gkmhip_memcpy_d2h.argtypes = (vp, vp, ctypes.c_size_t)
    L.gkmhip_memcpy_h2d.restype = i32
    L.gkmhip_memcpy_h2d.argtypes = (vp, vp, ctypes.c_size_t)
    L.gkmhip_sync.restype = i32
    L.gkmhip_sync.argtypes = (vp,)
    L.gkmhip_copy_lower_to_rows.restype = i32
    L.gkmhip_copy_lower_to_rows.argtypes = (vp, vp, i64, i32, vp, i32)
    L.gkmhip_gram_to_host_rows.restype = i32
    L.gkmhip_gram_to_host_rows.argtypes = (vp, vp, i64, vp, i32)
    L.gkmhip_gram_part_to_host_rows.restype = i32
    L.gkmhip_gram_part_to_host_rows.argtypes = (vp, vp, i64, vp, i32, i32, i32)
    L.gkmhip_release_host_cache.restype = None
    L.gkmhip_current_device.restype = i32
    L.gkmhip_set_current_device.restype = i32
    L.gkmhip_set_current_device.argtypes = (i32,)
    L.gkmhip_n_sequences.restype = i32
    L.gkmhip_n_sequences.argtypes = (vp,)
    L.gkmhip_device_of.restype = i32
    L.gkmhip_device_of.argtypes = (vp,)
    L.gkmhip_gram_allgather.restype = i32
    L.gkmhip_gram_allgather.argtypes = (vp, i32, vp, i64, i32, i32)
    if hasattr(L, "gkmhip_gram_rank_alone"):   # (a GKM_LIB_PATH library built from an older revision lacks it: A/B runs)
        L.gkmhip_gram_rank_alone.restype = i32
        L.gkmhip_gram_rank_alone.argtypes = (vp, i32, i32, i32, vp, i64, i32, vp)
    L.gkmhip_last_transport.restype = ctypes.c_char_p
    L.gkmhip_release_comms.restype = None
    if hasattr(L, "gkmhip_probe_copy"):
        L.gkmhip_probe_copy.restype = i32
        L.gkmhip_probe_copy.argtypes = (vp, vp, ctypes.c_size_t, i32, i32, vp)
    if hasattr(L, "gkmhip_kernel_timeline_spans"):
        L.gkmhip_kernel_timeline_spans.restype = i32
        L.gkmhip_kernel_timeline_spans.argtypes = (vp, vp, i32)
    if hasattr(L, "gkmhip_allgather_chunk_times"):
        L.gkmhip_allgather_chunk_times.restype = i32
        L.gkmhip_allgather_chunk_times.argtypes = (i32, vp, i32)
    if hasattr(L, "gkmhip_allgather_stats"):   # (older builds loaded through GKM_LIB_PATH for A/B timing lack them)
        L.gkmhip_allgather_alloc_count.restype = ctypes.c_long
        L.gkmhip_allgather_stats.restype = i32
        L.gkmhip_allgather_stats.argtypes = (vp, i32)
    L.gkmhip_assemble_normalize.restype = i32
    L.gkmhip_assemble_normalize.argtypes = (vp, vp, i64, vp, vp, i64, vp, i32, vp)
    L.gkmhip_last_kernel_ms.restype = dbl
    L.gkmhip_last_kernel_ms.argtypes = (vp,)
    if hasattr(L, "gkmhip_kernel_timeline"):
        L.gkmhip_kernel_timeline.argtypes = (vp, ctypes.c_int)
        L.gkmhip_kernel_timeline_ms.restype = dbl
        L.gkmhip_kernel_timeline_ms.argtypes = (vp, ctypes.POINTER(ctypes.c_int))
    L.gkmhip_last_comparisons.restype = dbl
    L.gkmhip_last_comparisons.argtypes = (vp,)
    L.gkmhip_last_kernel_name.restype = ctypes.c_char_p
    L.gkmhip_last_kernel_name.argtypes = (vp,)
    _lib = L
    return L


# ------------------------------------------------------------------ host tables
def check_parameters(kernel_type, L, k, d):
    msg = load().gkm_check_parameter_values(kernel_type, L, k, d)
    return msg.decode() if msg else None


def mismatch_weights(kernel_type, L, k):
    out = np.zeros(L + 1, dtype=np.float64)
    if load().gkm_mismatch_weights(kernel_type, L, k, out.ctypes.data):
        raise GkmError("invalid (kernel_type, L, k)")
    return out


def position_weights(kernel_type, n, M=50, H=50.0):
    out = np.zeros(max(n, 0), dtype=np.uint8)
    load().gkm_position_weights(kernel_type, n, M, float(H), out.ctypes.data)
    return out


def distance_weights(kernel_type, max_n, M=50, H=50.0):
    """wd[D] = positional weight of an l-mer at distance D from the centre l-mer; the
    reference's w(n, p) (src/libgkm.c:912-925) equals wd[|n//2 - p|] for every n."""
    dmax = max_n // 2 + 1
    return np.ascontiguousarray(position_weights(kernel_type, 2 * dmax + 1, M, H)[dmax:])


class FlatSequences:
    """All sequences of a problem back to back: `codes` (uint8, 0..3) and `off` (int64, n+1).
    Behaves like a list of per-sequence arrays (views) where one is needed."""

    def __init__(self, codes, off):
        self.codes, self.off = codes, off

    def __len__(self):
        return len(self.off) - 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        return self.codes[self.off[i]:self.off[i + 1]]

    def __iter__(self):
        return (self[i] for i in range(len(self)))


def read_problem(posfile, negfile):
    """FASTA pair -> (FlatSequences of base codes, n_pos, n_invalid_chars, n_truncated).
    Two bulk copies out of the C reader, no per-sequence work in Python."""
    L = load()
    h = L.gkm_problem_read(os.fsencode(posfile), os.fsencode(negfile))
    if not h:
        raise GkmError("cannot read %s / %s" % (posfile, negfile))
    try:
        n = L.gkm_problem_size(h)
        off = np.ctypeslib.as_array(L.gkm_problem_offsets(h), shape=(n + 1,)).copy()
        total = int(off[-1])
        codes = (np.ctypeslib.as_array(L.gkm_problem_all_codes(h), shape=(total,)).copy() if total
                 else np.zeros(0, np.uint8))
        return (FlatSequences(codes, off), L.gkm_problem_npos(h), L.gkm_problem_invalid_chars(h),
                L.gkm_problem_truncated(h))
    finally:
        L.gkm_problem_free(h)


def encode(seq):
    """bytes/str of ACGT (any case; other characters count as A) -> uint8 codes 0..3."""
    if isinstance(seq, str):
        seq = seq.encode()
    lut = np.zeros(256, dtype=np.uint8)
    for ch, v in ((b"C", 1), (b"G", 2), (b"T", 3), (b"c", 1), (b"g", 2), (b"t", 3)):
        lut[ch[0]] = v
    return lut[np.frombuffer(seq, dtype=np.uint8)]


# ------------------------------------------------------------------ device layer
class GramContext:
    """Owns one gkmhip_ctx: parameters + uploaded sequences on one GPU."""

    def __init__(self, kernel_type, L, k, d, M=50, H=50.0, gamma=1.0, device=0):
        bad = check_parameters(kernel_type, L, k, d)
        if bad:
            raise GkmError(bad)
        self.lib = load()
        self.kernel_type, self.L, self.k, self.d, self.M, self.H, self.gamma = kernel_type, L, k, d, M, H, gamma
        self.weighted = kernel_type in (4, 5)
        self.rbf = kernel_type in (3, 5)
        self.c = mismatch_weights(kernel_type, L, k)
        self.device = device
        self.n = 0
        self.handle = self.lib.gkmhip_create(device, L, d, self.c.ctypes.data, int(self.rbf), float(gamma))
        if not self.handle:
            raise GkmError("gkmhip_create: " + self.lib.gkmhip_last_error().decode())

    def _chk(self, rc, what):
        if rc:
            raise GkmError("%s failed (%d): %s" % (what, rc, self.lib.gkmhip_last_error().decode()))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.gkmhip_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_kernel(self, which):
        self._chk(self.lib.gkmhip_set_kernel(self.handle, which), "gkmhip_set_kernel")

    def set_scratch_slot(self, slot):
        """Launches alternating between two streams alternate the scratch slot (include/gkm_hip.h)."""
        self._chk(self.lib.gkmhip_set_scratch_slot(self.handle, slot), "gkmhip_set_scratch_slot")

    def set_sequences(self, seqs, stream=0):
        """seqs: list of uint8 arrays of base codes 0..3, or a FlatSequences."""
        n = len(seqs)
        if isinstance(seqs, FlatSequences):
            off = np.ascontiguousarray(seqs.off, dtype=np.int64)
            codes = np.ascontiguousarray(seqs.codes, dtype=np.uint8)
            lens = np.diff(off)
        else:
            lens = np.array([len(s) for s in seqs], dtype=np.int64)
            off = np.zeros(n + 1, dtype=np.int64)
            np.cumsum(lens, out=off[1:])
            codes = np.concatenate(seqs).astype(np.uint8) if n else np.zeros(0, np.uint8)
        if (lens < self.L).any():
            raise GkmError("a sequence is shorter than L")
        wd = None
        if self.weighted:
            wd = distance_weights(self.kernel_type, int(lens.max()) - self.L + 1, self.M, self.H)
        self._keep = (codes, off, wd)
        self._chk(self.lib.gkmhip_set_sequences(
            self.handle, n, codes.ctypes.data, off.ctypes.data,
            wd.ctypes.data if wd is not None else None, len(wd) if wd is not None else 0, stream),
            "gkmhip_set_sequences")
        self.n = n
        self.lens = lens

    def gram_rows(self, rows, G_ptr, ld, P_ptr=None, ldp=0, local_rows=True, stream=0):
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        self._chk(self.lib.gkmhip_gram_rows(self.handle, rows.ctypes.data, len(rows), int(local_rows), G_ptr, ld,
                                            P_ptr, ldp, stream), "gkmhip_gram_rows")

    def gram_rows_packed(self, rows, G_ptr, row_off, stream=0):
        """Row rows[i] -> G_ptr + row_off[i] doubles, columns 0..rows[i] only (packed slabs, sharding.py)."""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        row_off = np.ascontiguousarray(row_off[:len(rows)], dtype=np.int64)
        self._chk(self.lib.gkmhip_gram_rows_packed(self.handle, rows.ctypes.data, len(rows), G_ptr, row_off.ctypes.data,
                                                   stream), "gkmhip_gram_rows_packed")

    def gram_rows_full(self, rows, G_ptr, ld, local_rows=True, stream=0):
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        self._chk(self.lib.gkmhip_gram_rows_full(self.handle, rows.ctypes.data, len(rows), int(local_rows), G_ptr, ld,
                                                 stream), "gkmhip_gram_rows_full")

    def self_norms(self, sq_ptr, stream=0):
        self._chk(self.lib.gkmhip_self_norms(self.handle, sq_ptr, stream), "gkmhip_self_norms")

    def normalize_rows_full(self, rows, G_ptr, ld, sq_ptr, local_rows=True, stream=0):
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        self._chk(self.lib.gkmhip_normalize_rows_full(self.handle, rows.ctypes.data, len(rows), int(local_rows), G_ptr,
                                                      ld, sq_ptr, stream), "gkmhip_normalize_rows_full")

    def normalize(self, G_ptr, ld, sq_ptr=None, symmetric=False, stream=0):
        self._chk(self.lib.gkmhip_normalize(self.handle, G_ptr, ld, sq_ptr, int(symmetric), stream),
                  "gkmhip_normalize")

    def assemble_normalize(self, slabs_ptr, lds, slot_ptr, K_ptr, ld, sq_ptr, symmetric=False, stream=0):
        """Un-permute (matrix row a = row slot[a] of the gathered slabs; with lds == 1 slot[a] is the element offset
        at which row a starts: packed slabs) + normalise in one pass."""
        self._chk(self.lib.gkmhip_assemble_normalize(self.handle, slabs_ptr, lds, slot_ptr, K_ptr, ld, sq_ptr,
                                                     int(symmetric), stream), "gkmhip_assemble_normalize")

    def last_kernel_ms(self):
        return self.lib.gkmhip_last_kernel_ms(self.handle)

    def kernel_timeline(self, on):
        """While on, every launch keeps its own event pair: kernel_timeline_ms() sums the Gram kernels of a loop."""
        self._chk(self.lib.gkmhip_kernel_timeline(self.handle, 1 if on else 0), "gkmhip_kernel_timeline")

    def kernel_timeline_ms(self):
        k = ctypes.c_int(0)
        ms = self.lib.gkmhip_kernel_timeline_ms(self.handle, ctypes.byref(k))
        return ms, k.value

    def last_comparisons(self):
        return self.lib.gkmhip_last_comparisons(self.handle)

    def last_kernel_name(self):
        return self.lib.gkmhip_last_kernel_name(self.handle).decode()


def cross_kernel(seqs, rows, kernel_type, L, k, d, M=50, H=50.0, gamma=1.0, device=0, kernel=KERNEL_AUTO):
    """K(rows[i], j) for every sequence j (prediction-style rectangular kernel): torch fp64
    [len(rows), n] with 1.0 where j == rows[i], plus the self norms."""
    import torch
    ctx = GramContext(kernel_type, L, k, d, M, H, gamma, device)
    try:
        ctx.set_kernel(kernel)
        dev = torch.device("cuda", device)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream().cuda_stream
            ctx.set_sequences(seqs, stream)
            n = len(seqs)
            rows = np.ascontiguousarray(sorted(rows), dtype=np.int32)
            G = torch.zeros((len(rows), n), dtype=torch.float64, device=dev)
            sq = torch.zeros(n, dtype=torch.float64, device=dev)
            ctx.self_norms(sq.data_ptr(), stream)
            ctx.gram_rows_full(rows, G.data_ptr(), n, True, stream)
            ctx.normalize_rows_full(rows, G.data_ptr(), n, sq.data_ptr(), True, stream)
            torch.cuda.synchronize(dev)
            return dict(K=G, sqnorm=sq, rows=rows, kernel=ctx.last_kernel_name())
    finally:
        ctx.close()


_CTX_CACHE = {}
_CTX_CACHE_LOCK = __import__("threading").Lock()   # init_many's workers insert and release concurrently


def cached_context(kernel_type, L, k, d, M=50, H=50.0, gamma=1.0, device=0, slot=0):
    """One long-lived GramContext per (device, parameters).  Destroying a context frees device memory,
    which waits for EVERYTHING on the device (hipFree): a caller that evaluates subset after subset
    with the cross-validation of the previous one still running on another stream (gkmsvm.init_many)
    keeps its context instead, re-uploads the next subset's sequences into the same buffers and so
    never blocks on the other stream.  `slot` tells apart callers that work on the same device at the same time
    (gkmsvm.init_many with one worker per entry of `gpus`)."""
    key = (device, slot, kernel_type, L, k, d, int(M), float(H), float(gamma))
    with _CTX_CACHE_LOCK:
        ctx = _CTX_CACHE.get(key)
        if ctx is None or not ctx.handle:
            ctx = _CTX_CACHE[key] = GramContext(kernel_type, L, k, d, M, H, gamma, device)
    return ctx


def release_cached_contexts(device=None, slot=None):
    """Close the cached contexts (all of them, or those of one device / slot): each keeps its device scratch --
    the tile-transposed output alone is ~0.6 GB at n = 10 000 -- for as long as it lives."""
    with _CTX_CACHE_LOCK:
        gone = [_CTX_CACHE.pop(key) for key in list(_CTX_CACHE)
                if (device is None or key[0] == device) and (slot is None or key[1] == slot)]
    for ctx in gone:     # (hipFree waits for the device: outside the lock)
        ctx.close()


def gram_matrix(seqs, kernel_type, L, k, d, M=50, H=50.0, gamma=1.0, device=0, want_profiles=False,
                kernel=KERNEL_AUTO, symmetric=False, keep_context=False, context_slot=0, wait=True):
    """Whole Gram matrix of `seqs` on one GPU (device memory through torch).

    Returns dict(K=torch fp64 [n,n] (lower triangle + unit diagonal; upper too if symmetric),
    P=int32 [n,n,d+1] or None, sqnorm=[n], kernel=name, ms=device ms of the gram kernel).
    keep_context: use (and keep) the cached context of these parameters, see cached_context().
    wait=False (with keep_context): return as soon as the work is enqueued on torch's current stream -- whatever the
    caller enqueues on that stream next is ordered behind it, and the host is free meanwhile (gkmsvm.init draws the
    cross-validation folds while the Gram kernel runs); `ms` is then None."""
    import torch
    ctx = (cached_context(kernel_type, L, k, d, M, H, gamma, device, context_slot) if keep_context
           else GramContext(kernel_type, L, k, d, M, H, gamma, device))
    try:
        ctx.set_kernel(kernel)
        dev = torch.device("cuda", device)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream().cuda_stream
            ctx.set_sequences(seqs, stream)
            n = len(seqs)
            G = torch.zeros((n, n), dtype=torch.float64, device=dev)
            P = torch.zeros((n, n, d + 1), dtype=torch.int32, device=dev) if want_profiles else None
            sq = torch.zeros(n, dtype=torch.float64, device=dev)
            ctx.gram_rows(np.arange(n), G.data_ptr(), n, P.data_ptr() if want_profiles else None, n, False, stream)
            ctx.normalize(G.data_ptr(), n, sq.data_ptr(), symmetric, stream)
            if not (keep_context and not wait):
                torch.cuda.current_stream().synchronize()   # (this stream only: others may carry unrelated work)
            return dict(K=G, P=P, sqnorm=sq, kernel=ctx.last_kernel_name(),
                        ms=ctx.last_kernel_ms() if (wait or not keep_context) else None,
                        comparisons=ctx.last_comparisons())
    finally:
        if not keep_context:
            ctx.close()


def allgather_stats():
    """Per-rank HIP-event timings of the most recent gkmhip_gram_allgather (include/gkm_hip.h)."""
    buf = np.zeros(3 + 4 * 64)
    got = load().gkmhip_allgather_stats(buf.ctypes.data, len(buf))
    if not got:
        return None
    G = int(buf[0])
    per = buf[3:3 + 4 * G].reshape(G, 4)
    return dict(ranks=G, chunks=int(buf[1]), transport=("none", "p2p", "rccl")[int(buf[2])],
                kernel_ms=per[:, 0].tolist(), transfer_ms=per[:, 1].tolist(), assemble_ms=per[:, 2].tolist(),
                comparisons=per[:, 3].tolist())


def gram_matrix_multi(seqs, kernel_type, L, k, d, M=50, H=50.0, gamma=1.0, devices=(0,), symmetric=False, chunks=0,
                      kernel=KERNEL_AUTO):
    """Whole Gram matrix computed on several GPUs by ONE process (include/gkm_hip.h,
    gkmhip_gram_allgather): rows sharded by folded row blocks, slabs all-gathered over xGMI (RCCL),
    every device ends up with the whole normalised matrix -- bit-identical to gram_matrix().
    `devices` may name a device more than once (rehearsal on a one-GPU box: peer copies instead of RCCL).

    Returns dict(K=[torch fp64 [n,n] per entry of devices], transport="rccl"|"p2p"|"none", ms=wall)."""
    import time
    import torch
    lib = load()
    ctxs = []
    try:
        for dv in devices:
            c = GramContext(kernel_type, L, k, d, M, H, gamma, dv)
            c.set_kernel(kernel)
            ctxs.append(c)
        n = len(seqs)
        Ks = []
        for c, dv in zip(ctxs, devices):
            with torch.cuda.device(dv):
                c.set_sequences(seqs, torch.cuda.current_stream().cuda_stream)
                Ks.append(torch.zeros((n, n), dtype=torch.float64, device=torch.device("cuda", dv)))
        for dv in set(devices):
            torch.cuda.synchronize(dv)
        handles = (ctypes.c_void_p * len(ctxs))(*[c.handle for c in ctxs])
        outs = (ctypes.c_void_p * len(ctxs))(*[K.data_ptr() for K in Ks])
        t0 = time.perf_counter()
        rc = lib.gkmhip_gram_allgather(handles, len(ctxs), outs, n, int(symmetric), int(chunks))
        wall = time.perf_counter() - t0
        if rc:
            raise GkmError("gkmhip_gram_allgather failed (%d): %s" % (rc, lib.gkmhip_last_error().decode()))
        return dict(K=Ks, transport=lib.gkmhip_last_transport().decode(), ms=wall * 1e3)
    finally:
        for c in ctxs:
            c.close()
