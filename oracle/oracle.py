"""ctypes bindings of the CPU checkers.  TEST INFRASTRUCTURE ONLY.

  * liboracle.so              -- our plain-C restatement (oracle/gkm_oracle.c)
  * _ref/gkmkern_pylib_ref.so -- the unmodified reference, built by `make -C oracle ref`
  * _ref/ref_probe.so         -- harness exposing the reference's integer profiles

Nothing in gkmqc_amd/ imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class GkmOpt(ctypes.Structure):
    """gkmOpt, reference src/libgkm.h:149-161 == scripts/gkmsvm.py:48-61."""
    _fields_ = (
        ("kernel_type", ctypes.c_int), ("L", ctypes.c_int), ("k", ctypes.c_int), ("d", ctypes.c_int),
        ("M", ctypes.c_uint8), ("H", ctypes.c_double), ("gamma", ctypes.c_double),
        ("posfile", ctypes.c_char_p), ("negfile", ctypes.c_char_p),
        ("nthreads", ctypes.c_int), ("verbosity", ctypes.c_int),
    )


def make_opt(kernel_type, L, k, d, M=50, H=50.0, gamma=1.0, posfile="", negfile="", nthreads=1,
             verbosity=0):
    return GkmOpt(kernel_type, L, k, d, M, float(H), float(gamma), os.fsencode(posfile),
                  os.fsencode(negfile), nthreads, verbosity)


def build(ref=False):
    targets = ["all"] + (["ref"] if ref else [])
    subprocess.check_call(["make", "-s", "-C", HERE] + targets)


def _load(path):
    if not os.path.exists(path):
        raise OSError("%s not built (run `make -C oracle%s`)" % (path, " ref" if "_ref" in path else ""))
    return ctypes.CDLL(path)


_cache = {}


def lib():
    if "o" not in _cache:
        L = _load(os.path.join(HERE, "liboracle.so"))
        L.gkmo_check_params.restype = ctypes.c_char_p
        L.gkmo_position_weights.restype = None
        L.gkmo_profile.restype = None
        _cache["o"] = L
    return _cache["o"]


def have_ref():
    return os.path.exists(os.path.join(HERE, "_ref", "gkmkern_pylib_ref.so"))


def ref_lib():
    if "r" not in _cache:
        _cache["r"] = _load(os.path.join(HERE, "_ref", "gkmkern_pylib_ref.so"))
    return _cache["r"]


def ref_probe():
    if "p" not in _cache:
        _cache["p"] = _load(os.path.join(HERE, "_ref", "ref_probe.so"))
    return _cache["p"]


# ----------------------------------------------------------------------------- restatement
def mismatch_weights(kernel_type, L, k):
    out = np.zeros(L + 1, dtype=np.float64)
    rc = lib().gkmo_mismatch_weights(kernel_type, L, k, out.ctypes.data_as(ctypes.c_void_p))
    if rc:
        raise ValueError("bad parameters")
    return out


def position_weights(kernel_type, n, M=50, H=50.0):
    out = np.zeros(n, dtype=np.uint8)
    lib().gkmo_position_weights(kernel_type, n, ctypes.c_uint8(M), ctypes.c_double(H),
                                out.ctypes.data_as(ctypes.c_void_p))
    return out


class _Problem(ctypes.Structure):
    _fields_ = (("n", ctypes.c_int), ("n_pos", ctypes.c_int), ("len", ctypes.POINTER(ctypes.c_int)),
                ("seq", ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8))),
                ("n_invalid", ctypes.c_long), ("n_truncated", ctypes.c_long))


def read_problem(posfile, negfile):
    """-> (list of uint8 code arrays, n_pos, n_invalid, n_truncated)"""
    p = _Problem()
    rc = lib().gkmo_read_problem(os.fsencode(posfile), os.fsencode(negfile), ctypes.byref(p))
    if rc:
        raise OSError("gkmo_read_problem failed (%d)" % rc)
    seqs = [np.ctypeslib.as_array(p.seq[i], shape=(p.len[i],)).copy() for i in range(p.n)]
    res = (seqs, p.n_pos, p.n_invalid, p.n_truncated)
    lib().gkmo_free_problem(ctypes.byref(p))
    return res


def gram(opt, want_profiles=True, nthreads=8):
    """Full problem through the restatement -> dict(K, P, sqnorm, n_pos, n)."""
    p = _Problem()
    rc = lib().gkmo_read_problem(opt.posfile, opt.negfile, ctypes.byref(p))
    if rc:
        raise OSError("gkmo_read_problem failed (%d)" % rc)
    n, d = p.n, opt.d
    K = np.zeros((n, n), dtype=np.float64)
    sq = np.zeros(n, dtype=np.float64)
    P = np.zeros((n, n, d + 1), dtype=np.int32) if want_profiles else None
    rc = lib().gkmo_gram(ctypes.byref(opt), ctypes.byref(p), sq.ctypes.data_as(ctypes.c_void_p),
                         P.ctypes.data_as(ctypes.c_void_p) if want_profiles else None,
                         K.ctypes.data_as(ctypes.c_void_p), nthreads)
    n_pos = p.n_pos
    lib().gkmo_free_problem(ctypes.byref(p))
    if rc:
        raise ValueError("gkmo_gram failed (%d)" % rc)
    return dict(K=K, P=P, sqnorm=sq, n_pos=n_pos, n=n)


def _call_pywrapper(fn, opt, nrows):
    kmat = np.zeros((nrows, nrows), dtype=np.float64)
    rows = (kmat.ctypes.data + np.arange(nrows) * kmat.strides[0]).astype(np.uintp)
    sizes = np.ones(2, dtype=np.int32)
    fn.restype = ctypes.c_int
    rc = fn(ctypes.byref(opt), rows.ctypes.data_as(ctypes.c_void_p), sizes.ctypes.data_as(ctypes.c_void_p))
    return rc, kmat, int(sizes[0]), int(sizes[1])


def oracle_pywrapper(opt, nrows):
    return _call_pywrapper(lib().gkmo_main_pywrapper, opt, nrows)


# ----------------------------------------------------------------------------- reference
def ref_pywrapper(opt, nrows):
    """The unmodified reference gkm_main_pywrapper (src/gkmkern_pylib.c:92)."""
    return _call_pywrapper(ref_lib().gkm_main_pywrapper, opt, nrows)


def ref_weights(kernel_type, L, k, d):
    out = np.zeros(d + 1, dtype=np.float64)
    ref_probe().refp_weights(kernel_type, L, k, d, out.ctypes.data_as(ctypes.c_void_p))
    return out


def ref_profiles(opt, maxn=4096, wt_stride=2048):
    P = np.zeros((maxn, maxn, opt.d + 1), dtype=np.int32)
    sq = np.zeros(maxn, dtype=np.float64)
    lens = np.zeros(maxn, dtype=np.int32)
    wt = np.zeros((maxn, wt_stride), dtype=np.uint8)
    npos = ctypes.c_int(0)
    # P is addressed with the true N as row stride inside the probe: read it back after
    flat = np.zeros(maxn * maxn * (opt.d + 1), dtype=np.int32)
    n = ref_probe().refp_profiles(opt.kernel_type, opt.L, opt.k, opt.d, int(opt.M), ctypes.c_double(opt.H),
                                  ctypes.c_double(opt.gamma), opt.posfile, opt.negfile, maxn,
                                  flat.ctypes.data_as(ctypes.c_void_p), sq.ctypes.data_as(ctypes.c_void_p),
                                  lens.ctypes.data_as(ctypes.c_void_p), wt.ctypes.data_as(ctypes.c_void_p),
                                  wt_stride, ctypes.byref(npos))
    if n < 0:
        raise ValueError("refp_profiles failed")
    del P
    P = flat[: n * n * (opt.d + 1)].reshape(n, n, opt.d + 1).copy()
    return dict(P=P, sqnorm=sq[:n].copy(), lens=lens[:n].copy(), wt=wt[:n].copy(), n_pos=npos.value, n=n)


def ref_batch_rows(opt, rows, maxn=4096):
    """Rows `rows` of the problem scored against ALL its sequences by the reference's batch-vs-set entry
    gkmkernel_kernelfunc_batch (src/libgkm.c:1115-1153): [len(rows), n] normalised kernel values (RBF applied for
    types 3 / 5; the diagonal entry is G / sqnorm^2 as the reference computes it, not forced to 1.0)."""
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    out = np.zeros((len(rows), maxn), dtype=np.float64)
    n = ref_probe().refp_batch_rows(opt.kernel_type, opt.L, opt.k, opt.d, int(opt.M), ctypes.c_double(opt.H),
                                    ctypes.c_double(opt.gamma), opt.posfile, opt.negfile,
                                    rows.ctypes.data_as(ctypes.c_void_p), len(rows), maxn,
                                    out.ctypes.data_as(ctypes.c_void_p))
    if n < 0:
        raise ValueError("refp_batch_rows failed (%d)" % n)
    # the probe addresses `out` with the true n as row stride
    return out.reshape(-1)[: len(rows) * n].reshape(len(rows), n).copy()
