"""Parity of the HIP path (through the C ABI) against the oracle and the reference fixtures.
Integer mismatch profiles: bit-exact.  K: 1e-6 relative is the bar (north_star); the
observed error is at the 1e-15 level and the tests hold it to 1e-12."""
import ctypes
import hashlib
import os

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

K_TOL = 1e-12   # the contract is 1e-6 relative; fp64 epilogue gives ~1e-16


@pytest.fixture(scope="module")
def dev(built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from gkmqc_amd import device
    device.load()
    return device


@pytest.fixture(scope="module")
def quirk_seqs(dev):
    seqs, n_pos, _, _ = dev.read_problem(helpers.QUIRK_POS, helpers.QUIRK_NEG)
    return seqs, n_pos


def _check_against_case(res, c, n):
    il = np.tril_indices(n)
    P = res["P"].cpu().numpy()
    assert (P[il] == c["P"][il]).all(), "integer mismatch profiles differ (%s)" % res["kernel"]
    assert helpers.max_rel_err(res["sqnorm"].cpu().numpy(), c["sqnorm"]) < K_TOL
    K = res["K"].cpu().numpy()
    assert (np.diag(K) == 1.0).all()
    assert (np.triu(K, 1) == 0).all(), "cells above the diagonal must not be written"
    assert helpers.max_rel_err(helpers.tril_pack(K), c["K"]) < K_TOL


@pytest.mark.parametrize("idx", range(11))
def test_quirks_direct_kernel(dev, quirk_seqs, idx):
    """General kernel, every parameter set of the fixture (all kernel types, L=4..12, d<=6,
    M=255 wrap, lengths 12..2047, lowercase/N/CRLF/duplicate/reverse-complement/poly-A)."""
    seqs, _ = quirk_seqs
    c = helpers.quirks_expected()[0][idx]
    res = dev.gram_matrix(seqs, c["kernel_type"], c["L"], c["k"], c["d"], c["M"], c["H"], c["gamma"],
                          want_profiles=True, kernel=dev.KERNEL_DIRECT)
    assert res["kernel"] == "k_gram_direct"
    _check_against_case(res, c, len(seqs))


@pytest.mark.parametrize("idx", range(11))
def test_quirks_bitslice_kernel(dev, quirk_seqs, idx):
    """Bit-sliced kernel where instantiated; multi-segment rows (up to 2047 nt) included."""
    seqs, _ = quirk_seqs
    c = helpers.quirks_expected()[0][idx]
    try:
        res = dev.gram_matrix(seqs, c["kernel_type"], c["L"], c["k"], c["d"], c["M"], c["H"], c["gamma"],
                              want_profiles=True, kernel=dev.KERNEL_BITSLICE)
    except dev.GkmError as e:
        if "not instantiated" in str(e):
            pytest.skip("no bit-sliced instantiation for L=%d d=%d" % (c["L"], c["d"]))
        raise
    assert res["kernel"].startswith("k_gram_bitslice")
    _check_against_case(res, c, len(seqs))


@pytest.mark.parametrize("kernel", ["direct", "bitslice"])
def test_c1_full_matrix(dev, kernel):
    """BASELINE config 1: 200+200 x 300 bp, L=10 k=6 d=3, whole matrix vs the reference."""
    z = helpers.synthetic_expected()
    seqs = helpers.synth_codes(200, 200, 300)
    res = dev.gram_matrix(seqs, 2, 10, 6, 3, kernel=dev.KERNEL_DIRECT if kernel == "direct" else dev.KERNEL_BITSLICE)
    K = res["K"].cpu().numpy()
    assert helpers.max_rel_err(helpers.tril_pack(K), z["c1_full_K"]) < K_TOL


@pytest.mark.parametrize("name,npos,nneg,length,lr,t,L,k,d", [
    ("c2_cut192", 192, 192, 300, None, 4, 11, 7, 3),
    ("c5_cut64", 64, 64, 300, (150, 600), 4, 12, 8, 4),
])
def test_config_cuts(dev, name, npos, nneg, length, lr, t, L, k, d):
    """Cuts of BASELINE configs 2 and 5: int profiles bit-exact, K vs the reference."""
    z = helpers.synthetic_expected()
    seqs = helpers.synth_codes(npos, nneg, length, lr)
    n = npos + nneg
    for kern in (dev.KERNEL_BITSLICE, dev.KERNEL_DIRECT):
        res = dev.gram_matrix(seqs, t, L, k, d, want_profiles=True, kernel=kern)
        il = np.tril_indices(n)
        assert (res["P"].cpu().numpy()[il] == z[name + "_P"][il]).all()
        assert helpers.max_rel_err(res["sqnorm"].cpu().numpy(), z[name + "_sqnorm"]) < K_TOL
        assert helpers.max_rel_err(helpers.tril_pack(res["K"].cpu().numpy()), z[name + "_K"]) < K_TOL


def test_row_subsets_and_local_rows(dev):
    """gkmhip_gram_rows on arbitrary ascending row subsets (the multi-GPU shards) gives the
    same raw values as the full call, in both output placements."""
    import torch
    seqs = helpers.synth_codes(100, 90, 300, (150, 600))
    n = len(seqs)
    ctx = dev.GramContext(4, 11, 7, 3)
    stream = torch.cuda.current_stream().cuda_stream
    ctx.set_sequences(seqs, stream)
    full = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    ctx.gram_rows(np.arange(n), full.data_ptr(), n, None, 0, False, stream)
    rows = np.array(sorted(set(range(0, n, 3)) | {1, n - 1}), dtype=np.int32)
    loc = torch.zeros((len(rows), n), dtype=torch.float64, device="cuda")
    ctx.gram_rows(rows, loc.data_ptr(), n, None, 0, True, stream)
    glob = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    ctx.gram_rows(rows, glob.data_ptr(), n, None, 0, False, stream)
    torch.cuda.synchronize()
    full, loc, glob = full.cpu().numpy(), loc.cpu().numpy(), glob.cpu().numpy()
    for i, a in enumerate(rows):
        assert (loc[i, : a + 1] == full[a, : a + 1]).all()
        assert (glob[a, : a + 1] == full[a, : a + 1]).all()
    untouched = np.setdiff1d(np.arange(n), rows)
    assert (glob[untouched] == 0).all()
    ctx.close()


def test_kernel_timeline_sums_the_launches_of_a_loop(dev):
    """gkmhip_kernel_timeline (what bench.py's kernel_ms comes from): while on, every launch keeps its own pair of HIP
    events, so a loop enqueued without host waits can be read afterwards -- launches counted, the sum close to launches x
    one launch; off again, the most recent launch's time is what gkmhip_last_kernel_ms returns, as before."""
    import torch
    seqs = helpers.synth_codes(300, 300, 300)
    n = len(seqs)
    ctx = dev.GramContext(4, 11, 7, 3)
    stream = torch.cuda.current_stream().cuda_stream
    ctx.set_sequences(seqs, stream)
    G = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    rows = np.arange(n)
    ctx.gram_rows(rows, G.data_ptr(), n, None, 0, False, stream)
    torch.cuda.synchronize()
    one = ctx.last_kernel_ms()
    assert one > 0
    ctx.kernel_timeline(True)
    for _ in range(5):
        ctx.gram_rows(rows, G.data_ptr(), n, None, 0, False, stream)
    total, launches = ctx.kernel_timeline_ms()
    assert launches == 5 and 0.5 * 5 * one < total < 2.0 * 5 * one
    ctx.kernel_timeline(False)
    assert ctx.kernel_timeline_ms() == (0.0, 0)
    ctx.gram_rows(rows, G.data_ptr(), n, None, 0, False, stream)
    torch.cuda.synchronize()
    assert 0.5 * one < ctx.last_kernel_ms() < 2.0 * one
    ctx.close()


def test_profile_symmetry_property(dev):
    """Size-independent property: P_m(a,j) computed with a as the row equals P_m(j,a) with j as
    the row.  Checked by reversing the sequence order."""
    seqs = helpers.synth_codes(70, 70, 300, (100, 700))
    n = len(seqs)
    fwd = dev.gram_matrix(seqs, 4, 11, 7, 3, want_profiles=True)
    rev = dev.gram_matrix(seqs[::-1], 4, 11, 7, 3, want_profiles=True)
    Pf, Pr = fwd["P"].cpu().numpy(), rev["P"].cpu().numpy()
    for a in range(n):
        for j in range(a + 1):
            assert (Pf[a, j] == Pr[n - 1 - j, n - 1 - a]).all()


def test_boundary_end_to_end(dev):
    """gkm_main_pywrapper exactly as the reference's Python caller drives it
    (scripts/gkmsvm.py:67-99): row pointers into a larger zeroed matrix."""
    cases, lens, npos = helpers.quirks_expected()
    n = len(lens)
    for c in (cases[0], cases[4], cases[5]):
        opt = dev.gkmOpt(c["kernel_type"], c["L"], c["k"], c["d"], c["M"], c["H"], c["gamma"],
                         helpers.QUIRK_POS.encode(), helpers.QUIRK_NEG.encode(), 3, 0)
        kmat = np.zeros((n + 9, n + 9))
        rows = (kmat.ctypes.data + np.arange(kmat.shape[0]) * kmat.strides[0]).astype(np.uintp)
        sizes = np.ones(2, dtype=np.int32)
        rc = dev.load().gkm_main_pywrapper(ctypes.byref(opt), rows.ctypes.data, sizes.ctypes.data)
        assert rc == 0 and tuple(sizes) == (npos, n - npos)
        assert (np.triu(kmat, 1) == 0).all() and (kmat[n:] == 0).all() and (kmat[:, n:] == 0).all()
        assert (np.diag(kmat)[:n] == 1.0).all()
        # RBF types go through the device exp(): 1e-6 is the contract, allow a few ulp
        tol = 1e-10 if c["kernel_type"] in (3, 5) else K_TOL
        assert helpers.max_rel_err(helpers.tril_pack(kmat[:n, :n]), c["K"]) < tol


FULL_CONFIGS = {"c2": (5000, 5000, 300, None, 4, 11, 7, 3),       # BASELINE.json configs[1] (the headline)
                "c3": (10000, 10000, 300, None, 4, 11, 7, 3),     # configs[2]
                "c5": (5000, 5000, None, (150, 600), 4, 12, 8, 4),  # configs[4]
                # configs[3] stand-in: one `gkmqc.py evaluate` subset on peak-like sequences (reference
                # bin/gkmqc.py:150-154,181-185; generator gkmqc_amd.synth.make_peak_sequences)
                "c4": (5000, 5000, 600, None, 4, 10, 6, 3)}


@pytest.mark.parametrize("name", sorted(FULL_CONFIGS))
def test_full_size_against_reference_digest(dev, name):
    """BASELINE configurations at FULL size: SHA-256 of the lower triangle, sampled entries, row
    sums and total of the reference's own matrix (computed once in the build container by
    tests/golden/make_golden.py --full; the reference needs minutes to an hour per matrix)."""
    path = os.path.join(helpers.GOLDEN, name + "_full_digest.npz")
    if not os.path.exists(path):
        pytest.skip("digest fixture missing")
    z = np.load(path)
    npos, nneg, ln, lr, t, L, k, d = FULL_CONFIGS[name]
    if name == "c4":
        from gkmqc_amd import synth
        seqs = [dev.encode(x) for x in synth.make_peak_sequences(11, npos, ln, True) +
                synth.make_peak_sequences(12, nneg, ln, False)]
    else:
        seqs = helpers.synth_codes(npos, nneg, ln or 300, lr)
    res = dev.gram_matrix(seqs, t, L, k, d)
    K = res["K"].cpu().numpy()
    del res["K"]
    tri = helpers.tril_pack(K)
    assert helpers.max_rel_err(tri[z["sample_idx"]], z["sample_val"]) < K_TOL
    assert helpers.max_rel_err(np.tril(K, -1).sum(axis=1)[1:], z["row_sums"][1:]) < 1e-10
    assert abs(tri.sum() - float(z["total"])) < 1e-9 * abs(float(z["total"]))
    same = hashlib.sha256(tri.tobytes()).digest() == z["sha256"].tobytes()
    print("%s full: %s, %.1f ms device, bit-identical to the reference: %s" % (name, res["kernel"], res["ms"], same))
    assert same, "K differs from the reference in the last bits (still within %g)" % K_TOL


@pytest.mark.parametrize("name,nthreads", [("c2", 1), ("c2", 16), ("c4", 1), ("c4", 16)])
def test_drop_in_call_at_full_size_with_the_reference_callers_geometry(dev, tmp_path, name, nthreads):
    """gkm_main_pywrapper at N = 10 000 exactly as the reference's Python caller supplies its arguments on EVERY call
    (scripts/gkmsvm.py:75-77): a FRESH np.zeros((15000, 15000)) -- untouched pages --, row r at byte 120 000 r, -@ 1
    (gkmQC's default) and 16.  The strict lower triangle must have the SHA-256 of the reference's own matrix
    (tests/golden/c2_full_digest.npz / c4_full_digest.npz: configs[1], the configs[3] stand-in), the diagonal 1.0,
    everything else -- the upper triangle, rows and columns >= N -- must be as the caller left it (zero)."""
    from gkmqc_amd import synth
    z = np.load(os.path.join(helpers.GOLDEN, name + "_full_digest.npz"))
    npos, nneg, ln, lr, t, L, k, d = FULL_CONFIGS[name]
    n = npos + nneg
    pf, nf = str(tmp_path / "p.fa"), str(tmp_path / "n.fa")
    if name == "c4":
        synth.write_peak_problem(pf, nf, npos, nneg, ln)
    else:
        synth.write_problem(pf, nf, npos, nneg, ln or 300, lr)
    cap = 15000
    kmat = np.zeros((cap, cap))
    assert kmat.strides == (120000, 8)
    rows = (kmat.ctypes.data + np.arange(cap) * kmat.strides[0]).astype(np.uintp)
    sizes = np.full(2, -1, dtype=np.int32)
    opt = dev.gkmOpt(t, L, k, d, 50, 50.0, 1.0, pf.encode(), nf.encode(), nthreads, 0)
    rc = dev.load().gkm_main_pywrapper(ctypes.byref(opt), rows.ctypes.data, sizes.ctypes.data)
    assert rc == 0 and tuple(sizes) == (npos, nneg)
    h = hashlib.sha256()
    for a in range(1, n):
        h.update(memoryview(kmat[a, :a]))
    assert h.digest() == z["sha256"].tobytes(), "the drop-in call's lower triangle is not the reference's"
    assert (np.diagonal(kmat)[:n] == 1.0).all() and not np.diagonal(kmat)[n:].any()
    assert not kmat[n:].any() and not kmat[:n, n:].any(), "rows / columns >= N were written"
    for a in range(n):
        assert not kmat[a, a + 1:n].any(), "upper triangle written in row %d" % a


def _oracle_profiles(seqs, t, L, k, d, M=50, H=50.0):
    from oracle import oracle as O
    opt = O.make_opt(t, L, k, d, M, H)
    n = len(seqs)
    P = np.zeros((n, n, d + 1), dtype=np.int32)
    prof = np.zeros(d + 1, dtype=np.int32)
    vp = ctypes.c_void_p
    for a in range(n):
        for j in range(a + 1):
            O.lib().gkmo_profile(ctypes.byref(opt), seqs[a].ctypes.data_as(vp), len(seqs[a]),
                                 seqs[j].ctypes.data_as(vp), len(seqs[j]), prof.ctypes.data_as(vp))
            P[a, j] = prof
    return P


def test_dense_hits_and_repeats(dev):
    """Repeat-like input where almost every window is a hit (the rare path becomes the hot
    path: multi-hit words, list overflow handling): identical sequences, poly-A, dinucleotide
    and short-period repeats, next to random ones."""
    rng = np.random.default_rng(99)
    base = rng.integers(0, 4, 300).astype(np.uint8)
    seqs = [base.copy() for _ in range(12)]
    seqs += [np.zeros(300, np.uint8) for _ in range(8)]                                   # poly-A
    seqs += [np.tile(np.array([0, 1], np.uint8), 150) for _ in range(6)]                  # (AC)n
    seqs += [np.tile(np.array([0, 3, 3, 0, 2], np.uint8), 60) for _ in range(6)]          # period 5
    seqs += [np.full(300, 3, np.uint8) for _ in range(4)]                                 # poly-T (rc of poly-A)
    seqs += [rng.integers(0, 4, int(n)).astype(np.uint8) for n in rng.integers(40, 700, 40)]
    order = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in order]
    for (t, L, k, d) in [(4, 11, 7, 3), (2, 10, 6, 3), (4, 12, 8, 4)]:
        want = _oracle_profiles(seqs, t, L, k, d)
        il = np.tril_indices(len(seqs))
        for kern in (dev.KERNEL_BITSLICE, dev.KERNEL_DIRECT):
            res = dev.gram_matrix(seqs, t, L, k, d, want_profiles=True, kernel=kern)
            assert (res["P"].cpu().numpy()[il] == want[il]).all(), (t, L, d, res["kernel"])


def test_int32_wraparound_matches_wrapping_arithmetic(dev):
    """Long sequences with M=255 overflow the reference's int accumulators;
    the kernels wrap modulo 2^32 exactly like the oracle's unsigned arithmetic."""
    seqs = [np.zeros(2047, np.uint8), np.zeros(2000, np.uint8), np.tile(np.array([0, 0, 1], np.uint8), 600)]
    want = _oracle_profiles(seqs, 4, 10, 6, 3, M=255, H=2000.0)
    assert (want < 0).any() or (np.abs(want.astype(np.int64)) > 2 ** 30).any()
    il = np.tril_indices(len(seqs))
    for kern in (dev.KERNEL_BITSLICE, dev.KERNEL_DIRECT):
        res = dev.gram_matrix(seqs, 4, 10, 6, 3, 255, 2000.0, want_profiles=True, kernel=kern)
        assert (res["P"].cpu().numpy()[il] == want[il]).all()


ALL_LD = [(L, d) for L in range(5, 13) for d in range(0, 5)] + [(11, 5), (12, 5), (12, 6)]


@pytest.mark.parametrize("L,d", ALL_LD)
def test_every_bitslice_instantiation_against_the_general_kernel(dev, L, d):
    """All (L, d) pairs the bit-sliced kernel is instantiated for, weighted and unweighted, on
    mixed-length input with multi-segment rows: integer profiles equal to the general kernel's
    (itself pinned to the reference fixtures) and, on a subset, to the oracle's."""
    rng = np.random.default_rng(100 * L + d)
    seqs = [rng.integers(0, 4, int(n)).astype(np.uint8) for n in rng.integers(L, 900, 70)]
    seqs[3] = seqs[7].copy()
    seqs[11] = (3 - seqs[20][::-1]).astype(np.uint8)      # reverse complement of another sequence
    k = L - d
    for t in (2, 4):
        a = dev.gram_matrix(seqs, t, L, k, d, want_profiles=True, kernel=dev.KERNEL_BITSLICE)
        b = dev.gram_matrix(seqs, t, L, k, d, want_profiles=True, kernel=dev.KERNEL_DIRECT)
        assert a["kernel"].startswith("k_gram_bitslice") and b["kernel"] == "k_gram_direct"
        il = np.tril_indices(len(seqs))
        Pa, Pb = a["P"].cpu().numpy(), b["P"].cpu().numpy()
        assert (Pa[il] == Pb[il]).all()
        assert (a["K"].cpu().numpy() == b["K"].cpu().numpy()).all()
        sub = seqs[:14]
        want = _oracle_profiles(sub, t, L, k, d)
        is_ = np.tril_indices(len(sub))
        assert (Pa[:14, :14][is_] == want[is_]).all()


def test_auto_takes_the_faster_kernel(dev):
    """`auto` sends an (L, d) whose iid share of window pairs within d mismatches exceeds ~7.5 % to the general kernel,
    although the bit-sliced one is instantiated for it (every hit takes a lane of a trip there; measured break-even,
    profiles/r4_high_d_bitslice_vs_direct.txt), and takes the bit-sliced kernel below that.  Same numbers either way."""
    rng = np.random.default_rng(17)
    seqs = [rng.integers(0, 4, int(n)).astype(np.uint8) for n in rng.integers(40, 400, 90)]
    il = np.tril_indices(len(seqs))
    for L, d, want in ((11, 3, "bitslice"), (7, 3, "bitslice"), (12, 6, "bitslice"), (9, 4, "bitslice"),
                       (10, 5, "direct"), (8, 4, "direct"), (6, 3, "direct"), (5, 4, "direct"), (11, 6, "direct"), (10, 7, "direct")):
        auto = dev.gram_matrix(seqs, 4, L, L - d, d, want_profiles=True)
        assert ("bitslice" in auto["kernel"]) == (want == "bitslice"), (L, d, auto["kernel"])
        other = dev.gram_matrix(seqs, 4, L, L - d, d, want_profiles=True,
                                kernel=dev.KERNEL_DIRECT if want == "bitslice" else dev.KERNEL_BITSLICE) if (L, d) in ALL_LD else None
        if other is not None:
            assert other["kernel"] != auto["kernel"]
            assert (auto["P"].cpu().numpy()[il] == other["P"].cpu().numpy()[il]).all()
            assert (auto["K"].cpu().numpy() == other["K"].cpu().numpy()).all()
    # ... and by what the uploaded sequences themselves say (round 5: 8 192 sampled l-mer pairs at upload) where that is
    # far above the iid rate: low-complexity input -- poly-A with a few substitutions -- has nearly every window pair
    # within d = 3 of 11, which would put every lane of every trip to work: the general kernel, same numbers
    low = []
    for n in rng.integers(100, 320, 60):
        x = np.zeros(int(n), dtype=np.uint8)
        at = rng.integers(0, int(n), 4)
        x[at] = rng.integers(0, 4, 4)
        low.append(x)
    il2 = np.tril_indices(len(low))
    auto = dev.gram_matrix(low, 4, 11, 7, 3, want_profiles=True)
    forced = dev.gram_matrix(low, 4, 11, 7, 3, want_profiles=True, kernel=dev.KERNEL_BITSLICE)
    assert auto["kernel"] == "k_gram_direct" and "bitslice" in forced["kernel"]
    assert (auto["P"].cpu().numpy()[il2] == forced["P"].cpu().numpy()[il2]).all()
    assert (auto["K"].cpu().numpy() == forced["K"].cpu().numpy()).all()


def test_general_kernel_with_several_columns_per_workgroup(dev):
    """k_gram_direct takes 1..16 columns per workgroup, sized by the problem (one column for the small problems of this
    suite): 2 600 short ragged sequences make it three columns per workgroup (with a ragged last chunk per tile), against
    the bit-sliced kernel -- profiles and values bit for bit.  (16 columns: tools/high_d_ab.py --n 8000, same check.)"""
    rng = np.random.default_rng(29)
    seqs = [rng.integers(0, 4, int(n)).astype(np.uint8) for n in rng.integers(24, 72, 2600)]
    a = dev.gram_matrix(seqs, 4, 8, 6, 2, want_profiles=True, kernel=dev.KERNEL_DIRECT)
    b = dev.gram_matrix(seqs, 4, 8, 6, 2, want_profiles=True, kernel=dev.KERNEL_BITSLICE)
    assert a["kernel"] == "k_gram_direct" and b["kernel"].startswith("k_gram_bitslice")
    il = np.tril_indices(len(seqs))
    assert (a["P"].cpu().numpy()[il] == b["P"].cpu().numpy()[il]).all()
    assert (a["K"].cpu().numpy() == b["K"].cpu().numpy()).all()


def test_packed_variants_on_fixed_length_data(dev, monkeypatch):
    """Fixed-length data normally takes the one-piece-per-lane variant; force the two several-pieces
    variants (64 and 128 row slots per tile) on it as well."""
    z = helpers.synthetic_expected()
    seqs = helpers.synth_codes(192, 192, 300)
    il = np.tril_indices(len(seqs))
    for forced, name in ((None, "k_gram_bitslice<same length>"), ("1", "k_gram_bitslice<packed>"), ("128", "k_gram_bitslice<packed,128>")):
        if forced:
            monkeypatch.setenv("GKM_FORCE_PACKED", forced)
        res = dev.gram_matrix(seqs, 4, 11, 7, 3, want_profiles=True, kernel=dev.KERNEL_BITSLICE)
        assert res["kernel"] == name
        assert (res["P"].cpu().numpy()[il] == z["c2_cut192_P"][il]).all()
        assert helpers.max_rel_err(helpers.tril_pack(res["K"].cpu().numpy()), z["c2_cut192_K"]) < K_TOL


def test_same_length_and_ragged_one_piece_rows(dev, monkeypatch):
    """Where every sequence has the same length a trip learns the source lane's row slot and piece index from the record's
    origin word itself (no table, no permute: `<same length>`), on 300-bp rows (one lane each) and 600-bp rows (two lanes
    each); ragged lengths that still take one lane per row go to the several-pieces variant, which serves every other
    shape.  Word lengths below 5 have no bit-sliced kernel (a group of five windows must fit the zero bytes either side
    of a weight table): `auto` takes the general kernel there and an explicit request is refused."""
    z = helpers.synthetic_expected()
    seqs = helpers.synth_codes(192, 192, 300)
    il = np.tril_indices(len(seqs))
    long_rows = helpers.synth_codes(70, 70, 600)
    want = dev.gram_matrix(long_rows, 4, 10, 6, 3, want_profiles=True, kernel=dev.KERNEL_DIRECT)
    il2 = np.tril_indices(len(long_rows))
    res = dev.gram_matrix(seqs, 4, 11, 7, 3, want_profiles=True, kernel=dev.KERNEL_BITSLICE)
    assert res["kernel"] == "k_gram_bitslice<same length>"
    assert (res["P"].cpu().numpy()[il] == z["c2_cut192_P"][il]).all()
    res = dev.gram_matrix(long_rows, 4, 10, 6, 3, want_profiles=True, kernel=dev.KERNEL_BITSLICE)
    assert res["kernel"] == "k_gram_bitslice<same length>"
    assert (res["P"].cpu().numpy()[il2] == want["P"].cpu().numpy()[il2]).all()
    assert (res["K"].cpu().numpy() == want["K"].cpu().numpy()).all()
    ragged = helpers.synth_codes(100, 100, 300, (305, 320))
    il3 = np.tril_indices(len(ragged))
    r = dev.gram_matrix(ragged, 4, 11, 7, 3, want_profiles=True, kernel=dev.KERNEL_BITSLICE)
    w = dev.gram_matrix(ragged, 4, 11, 7, 3, want_profiles=True, kernel=dev.KERNEL_DIRECT)
    assert r["kernel"] == "k_gram_bitslice<packed>"
    assert (r["P"].cpu().numpy()[il3] == w["P"].cpu().numpy()[il3]).all() and (r["K"].cpu().numpy() == w["K"].cpu().numpy()).all()
    # same-length rows short enough for two per lane: several pieces per lane, so not the same-length variant
    short = helpers.synth_codes(150, 150, 140)
    r = dev.gram_matrix(short, 4, 10, 6, 3, want_profiles=True, kernel=dev.KERNEL_BITSLICE)
    w = dev.gram_matrix(short, 4, 10, 6, 3, want_profiles=True, kernel=dev.KERNEL_DIRECT)
    il4 = np.tril_indices(len(short))
    assert r["kernel"].startswith("k_gram_bitslice<packed")
    assert (r["P"].cpu().numpy()[il4] == w["P"].cpu().numpy()[il4]).all() and (r["K"].cpu().numpy() == w["K"].cpu().numpy()).all()
    assert dev.gram_matrix(ragged, 4, 4, 2, 2)["kernel"] == "k_gram_direct"
    with pytest.raises(dev.GkmError, match="not instantiated"):
        dev.gram_matrix(ragged, 4, 4, 2, 2, kernel=dev.KERNEL_BITSLICE)


def test_reused_context_with_longer_second_subset(dev):
    """A cached context (gkmsvm.init_many keeps one per device) evaluated on a second subset whose sequences are
    longer: every per-sequence table (strand bit planes, packed strands, the dynamic LDS they size) must follow
    the NEW subset.  Short -> long -> short again, weighted and unweighted, bit for bit against fresh contexts."""
    first = helpers.synth_codes(70, 70, 300, (150, 280))
    second = helpers.synth_codes(60, 60, 600, (590, 700))     # maxlen larger by far more than 16 bp
    try:
        for t, L, k, d in ((4, 10, 6, 3), (2, 11, 7, 3)):
            for seqs in (first, second, first):
                kept = dev.gram_matrix(seqs, t, L, k, d, want_profiles=True, kernel=dev.KERNEL_BITSLICE, keep_context=True,
                                       context_slot=7)
                fresh = dev.gram_matrix(seqs, t, L, k, d, want_profiles=True, kernel=dev.KERNEL_DIRECT)
                il = np.tril_indices(len(seqs))
                assert (kept["P"].cpu().numpy()[il] == fresh["P"].cpu().numpy()[il]).all()
                assert (kept["K"].cpu().numpy() == fresh["K"].cpu().numpy()).all()
    finally:
        dev.release_cached_contexts()


def test_many_short_rows_take_the_128_slot_variant(dev):
    """Rows much shorter than a lane: more than 64 of them fit a tile, so the host keeps 128 row slots
    (capping the tile at 64 rows would leave lanes empty); ragged 150-600 bp rows take the 64-slot variant."""
    short = helpers.synth_codes(150, 150, 300, (60, 110))
    a = dev.gram_matrix(short, 4, 10, 6, 3, want_profiles=True, kernel=dev.KERNEL_BITSLICE)
    b = dev.gram_matrix(short, 4, 10, 6, 3, want_profiles=True, kernel=dev.KERNEL_DIRECT)
    assert a["kernel"] == "k_gram_bitslice<packed,128>"
    il = np.tril_indices(len(short))
    assert (a["P"].cpu().numpy()[il] == b["P"].cpu().numpy()[il]).all()
    ragged = helpers.synth_codes(100, 100, 300, (150, 600))
    assert dev.gram_matrix(ragged, 4, 12, 8, 4, kernel=dev.KERNEL_BITSLICE)["kernel"] == "k_gram_bitslice<packed>"


def test_tiny_and_degenerate_problems(dev, tmp_path):
    """1 + 1 sequences, sequences of exactly L bases, many very short sequences (more rows than
    a tile has row slots), through the boundary and through the device layer."""
    rng = np.random.default_rng(3)

    def fasta(path, seqs):
        with open(path, "wb") as f:
            for i, s in enumerate(seqs):
                f.write(b">s%d\n" % i + bytes(b"ACGT"[c] for c in s) + b"\n")

    # (a) the smallest legal problem through gkm_main_pywrapper
    a, b = rng.integers(0, 4, 11).astype(np.uint8), rng.integers(0, 4, 40).astype(np.uint8)
    pf, nf = str(tmp_path / "p.fa"), str(tmp_path / "n.fa")
    fasta(pf, [a])
    fasta(nf, [b])
    kmat = np.zeros((4, 4))
    rows = (kmat.ctypes.data + np.arange(4) * kmat.strides[0]).astype(np.uintp)
    sizes = np.zeros(2, dtype=np.int32)
    opt = dev.gkmOpt(4, 11, 7, 3, 50, 50.0, 1.0, pf.encode(), nf.encode(), 1, 0)
    assert dev.load().gkm_main_pywrapper(ctypes.byref(opt), rows.ctypes.data, sizes.ctypes.data) == 0
    want = _oracle_profiles([a, b], 4, 11, 7, 3)
    c = dev.mismatch_weights(4, 11, 7)[:4]
    g = (want.astype(np.float64) * c).sum(axis=2)
    assert tuple(sizes) == (1, 1) and kmat[0, 0] == 1.0 and kmat[1, 1] == 1.0 and kmat[0, 1] == 0.0
    assert abs(kmat[1, 0] - g[1, 0] / np.sqrt(g[0, 0] * g[1, 1])) < 1e-15
    # (b) 300 sequences of 11..40 bases: several rows per lane, more than 128 rows -> several tiles
    seqs = [rng.integers(0, 4, int(n)).astype(np.uint8) for n in rng.integers(11, 41, 300)]
    want = _oracle_profiles(seqs, 2, 11, 7, 3)
    il = np.tril_indices(len(seqs))
    for kern in (dev.KERNEL_BITSLICE, dev.KERNEL_DIRECT):
        res = dev.gram_matrix(seqs, 2, 11, 7, 3, want_profiles=True, kernel=kern)
        assert (res["P"].cpu().numpy()[il] == want[il]).all(), res["kernel"]


@pytest.mark.parametrize("kernel", ["bitslice", "direct"])
def test_rectangular_kernel_and_self_norms(dev, kernel):
    """gkmhip_gram_rows_full / gkmhip_self_norms / gkmhip_normalize_rows_full: K(a, j) for every j
    equals the symmetric completion of the triangular matrix (the profile is symmetric,
    SURVEY.md App. A.3), self norms from the diagonal band equal those of the full run."""
    seqs = helpers.synth_codes(60, 70, 300, (120, 650))
    kern = dev.KERNEL_BITSLICE if kernel == "bitslice" else dev.KERNEL_DIRECT
    full = dev.gram_matrix(seqs, 4, 11, 7, 3, kernel=kern)
    K = full["K"].cpu().numpy()
    Ksym = np.tril(K) + np.tril(K, -1).T
    rows = [0, 3, 4, 59, 60, 61, 100, 129]
    rect = dev.cross_kernel(seqs, rows, 4, 11, 7, 3, kernel=kern)
    assert (rect["sqnorm"].cpu().numpy() == full["sqnorm"].cpu().numpy()).all()
    got = rect["K"].cpu().numpy()
    for i, a in enumerate(rect["rows"]):
        assert (got[i] == Ksym[a]).all(), a


@pytest.mark.parametrize("kernel", ["bitslice", "direct"])
def test_scoring_against_a_set_matches_the_reference(dev, kernel):
    """SURVEY.md section 8 row f4, the batch-vs-set half: gkmhip_self_norms + gkmhip_gram_rows_full +
    gkmhip_normalize_rows_full (device.cross_kernel) against OUTPUT OF THE REFERENCE's gkmkernel_kernelfunc_batch
    (src/libgkm.c:1115-1153; tests/golden/batch_rows_expected.npz, made by make_golden.py --only-batch from the
    compiled reference): 40 query rows x 240 sequences of 60-700 bp (types 2, 4, 5, 0; L 8-12; d 3-4) and 24 x 144
    fixed-length ones (types 4 and 3 = RBF).  Off the diagonal the values must be IDENTICAL (same integer profiles,
    same fp64 operation order: sum over m ascending, product of the norms first, one division, libgkm.c:576-582,
    1140-1143; the RBF types pass through the device's exp(): 1e-12); on the diagonal the reference's entry leaves G / sqnorm^2 (1.0 to rounding) where this build writes
    1.0 as gkm_main_pywrapper does (gkmkern_pylib.c:218-221)."""
    kern = dev.KERNEL_BITSLICE if kernel == "bitslice" else dev.KERNEL_DIRECT
    for c in helpers.batch_rows_expected():
        seqs = helpers.synth_codes(c["n_support"], c["n_query"], c["length"], c["length_range"])
        rows = list(range(c["n_support"], c["n_support"] + c["n_query"]))
        got = dev.cross_kernel(seqs, rows, c["kernel_type"], c["L"], c["k"], c["d"], c["M"], c["H"], c["gamma"],
                               kernel=kern)
        K = got["K"].cpu().numpy()
        assert K.shape == c["K"].shape, c["name"]
        for i, a in enumerate(rows):
            off = np.arange(K.shape[1]) != a
            err = helpers.max_rel_err(K[i][off], c["K"][i][off])
            if c["kernel_type"] in (3, 5):    # RBF: exp() of the device against the host's libm, a few ulp
                assert err < 1e-12, (c["name"], a, err)
            else:
                assert np.array_equal(K[i][off], c["K"][i][off]), (c["name"], a, err)
            assert K[i][a] == 1.0 and abs(c["K"][i][a] - 1.0) < 1e-12


@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_boundary_over_several_device_contexts(dev, monkeypatch, devices):
    """GKM_DEVICES: the boundary call spreads row blocks over one context + host thread per listed
    device (the same GPU listed several times on a one-GPU box): identical matrix."""
    z = helpers.synthetic_expected()
    import tempfile
    from gkmqc_amd import synth
    tmp = tempfile.mkdtemp()
    pf, nf = os.path.join(tmp, "p.fa"), os.path.join(tmp, "n.fa")
    synth.write_problem(pf, nf, 192, 192, 300)
    n = 384

    def call():
        kmat = np.zeros((n, n))
        rows = (kmat.ctypes.data + np.arange(n) * kmat.strides[0]).astype(np.uintp)
        sizes = np.zeros(2, dtype=np.int32)
        opt = dev.gkmOpt(4, 11, 7, 3, 50, 50.0, 1.0, pf.encode(), nf.encode(), 2, 0)
        assert dev.load().gkm_main_pywrapper(ctypes.byref(opt), rows.ctypes.data, sizes.ctypes.data) == 0
        return kmat

    one = call()
    monkeypatch.setenv("GKM_DEVICES", devices)
    many = call()
    assert (one == many).all()
    assert (np.triu(many, 1) == 0).all() and (np.diag(many) == 1.0).all()
    assert helpers.max_rel_err(helpers.tril_pack(many), z["c2_cut192_K"]) < K_TOL


@pytest.mark.parametrize("part", [0, 1, 2])
def test_multi_device_boundary_failure_leaves_the_matrix_untouched(dev, monkeypatch, tmp_path, part):
    """GKM_DEVICES path: every device goes through context, upload and allocation BEFORE any device computes, so a
    device that fails there (here: an injected allocation failure on one of three) makes the call return non-zero
    without a single cell of the caller's matrix or of kmat_size written -- the reference returns from its own checks
    before writing anything (src/gkmkern_pylib.c:157-161).  The next call, without the fault, works."""
    from gkmqc_amd import synth
    pf, nf = str(tmp_path / "p.fa"), str(tmp_path / "n.fa")
    synth.write_problem(pf, nf, 150, 150, 300)
    n = 300
    kmat = np.full((n, n), -7.5)
    rows = (kmat.ctypes.data + np.arange(n) * kmat.strides[0]).astype(np.uintp)
    sizes = np.full(2, -3, dtype=np.int32)
    opt = dev.gkmOpt(4, 11, 7, 3, 50, 50.0, 1.0, pf.encode(), nf.encode(), 2, 0)
    monkeypatch.setenv("GKM_DEVICES", "0,0,0")
    monkeypatch.setenv("GKM_FAULT_INJECT", "alloc:%d" % part)
    assert dev.load().gkm_main_pywrapper(ctypes.byref(opt), rows.ctypes.data, sizes.ctypes.data) != 0
    assert (kmat == -7.5).all() and (sizes == -3).all()
    monkeypatch.delenv("GKM_FAULT_INJECT")
    assert dev.load().gkm_main_pywrapper(ctypes.byref(opt), rows.ctypes.data, sizes.ctypes.data) == 0
    assert (np.diag(kmat) == 1.0).all() and (kmat[np.triu_indices(n, 1)] == -7.5).all() and tuple(sizes) == (150, 150)
    monkeypatch.delenv("GKM_DEVICES")
    one = np.zeros((n, n))
    rows1 = (one.ctypes.data + np.arange(n) * one.strides[0]).astype(np.uintp)
    assert dev.load().gkm_main_pywrapper(ctypes.byref(opt), rows1.ctypes.data, sizes.ctypes.data) == 0
    il = np.tril_indices(n)
    assert (kmat[il] == one[il]).all()


@pytest.mark.parametrize("nctx,chunks", [(2, 0), (3, 2), (5, 4)])
def test_one_process_multi_gpu_assembly(dev, quirk_seqs, nctx, chunks):
    """gkmhip_gram_allgather (include/gkm_hip.h): several contexts -- here all on the one GPU of the
    box, so the slabs move by peer copies -- must assemble the same matrix, bit for bit, as one
    context alone; every context's copy is checked, ragged lengths and the tiny quirks problem
    (fewer rows than 2 x ranks x 64) included."""
    import torch
    seqs, _ = quirk_seqs
    big = helpers.synth_codes(300, 300, 300, (150, 400))
    for problem, (t, L, k, d) in ((seqs, (4, 11, 7, 3)), (big, (5, 10, 6, 3))):
        one = dev.gram_matrix(problem, t, L, k, d, gamma=2.0, symmetric=True)["K"]
        res = dev.gram_matrix_multi(problem, t, L, k, d, gamma=2.0, devices=[0] * nctx, symmetric=True, chunks=chunks)
        assert res["transport"] == "p2p"
        for K in res["K"]:
            assert torch.equal(K, one), "assembled matrix differs from the single-GPU matrix"
        low = dev.gram_matrix_multi(problem, t, L, k, d, gamma=2.0, devices=[0] * nctx, symmetric=False, chunks=chunks)
        assert torch.equal(low["K"][nctx - 1], torch.tril(one))


@pytest.mark.parametrize("nctx,chunks", [(2, 0), (4, 3)])
def test_one_rank_alone_reassembles_the_matrix(dev, nctx, chunks):
    """gkmhip_gram_rank_alone (the measurement entry behind tools/rank_alone.py and DESIGN.md section 7's table): after a
    gkmhip_gram_allgather over `nctx` contexts, every rank g re-runs ITS step alone -- its chunks into its packed slab,
    the slab into its gathered buffer, un-permute + normalise of the whole matrix -- and must leave the same matrix,
    bit for bit, in a zeroed K; without the earlier call (no gathered slabs of that shape) it must refuse."""
    import torch
    problem = helpers.synth_codes(200, 180, 300, (150, 420))
    t, L, k, d = 4, 11, 7, 3
    one = torch.tril(dev.gram_matrix(problem, t, L, k, d)["K"])
    lib = dev.load()
    n = len(problem)
    stream = torch.cuda.current_stream().cuda_stream
    ctxs = [dev.GramContext(t, L, k, d, 50, 50.0, 1.0, 0) for _ in range(nctx)]
    try:
        for c in ctxs:
            c.set_sequences(problem, stream)
        Ks = [torch.zeros((n, n), dtype=torch.float64, device="cuda") for _ in range(nctx)]
        out6 = np.zeros(6)
        lib.gkmhip_release_comms()
        assert lib.gkmhip_gram_rank_alone(ctxs[0].handle, 0, nctx, chunks, Ks[0].data_ptr(), n, 0, out6.ctypes.data) != 0
        handles = (ctypes.c_void_p * nctx)(*[c.handle for c in ctxs])
        outs = (ctypes.c_void_p * nctx)(*[K.data_ptr() for K in Ks])
        assert lib.gkmhip_gram_allgather(handles, nctx, outs, n, 0, chunks) == 0
        for g in range(nctx):
            assert torch.equal(Ks[g], one)
            Ks[g].zero_()
            torch.cuda.synchronize()
            rc = lib.gkmhip_gram_rank_alone(ctxs[g].handle, g, nctx, chunks, Ks[g].data_ptr(), n, 0, out6.ctypes.data)
            assert rc == 0, lib.gkmhip_last_error().decode()
            assert torch.equal(Ks[g], one), "rank %d alone" % g
            assert out6[0] > 0 and out6[1] > 0 and out6[3] > 0 and out6[4] > 0 and int(out6[5]) >= 2
            # the chunks of a rank follow each other on ONE compute stream (gkm_multi.hip compute_streams()): chunk c is
            # complete before chunk c + 1 starts, or its transfer could hide behind nothing
            ct = np.zeros(4 * 16)
            nct = lib.gkmhip_allgather_chunk_times(g, ct.ctypes.data, len(ct))
            assert nct == 4 * int(out6[5])
            ct = ct[:nct].reshape(-1, 4)
            assert (ct[:-1, 1] <= ct[1:, 0] + 1e-3).all() and (ct[:, 0] <= ct[:, 1]).all(), ct
    finally:
        for c in ctxs:
            c.close()
        lib.gkmhip_release_comms()


@pytest.mark.parametrize("chunk", ["8", "40", "64"])
def test_column_chunked_work_item_order(dev, quirk_seqs, monkeypatch, chunk):
    """The bit-sliced kernel's work items can be ordered by (column chunk, tile) entries (all tiles take a chunk of
    columns before any takes the next: the chunk's column tables then come from the XCD's L2; taken by default at
    n > 4 096 for ~300-bp data, so the full-size digest tests run it too).  Tiny chunks here: many entries, padding
    items (entries are rounded up to 8 items), ragged lengths, the triangle, the full rectangle and the diagonal band --
    the same integers and the same matrix as the plain order."""
    import torch
    seqs, _ = quirk_seqs
    ragged = helpers.synth_codes(150, 150, 300, (150, 700))
    for problem, (t, L, k, d) in ((seqs, (4, 11, 7, 3)), (ragged, (4, 12, 8, 4))):
        monkeypatch.setenv("GKM_COL_CHUNK", "0")
        want = dev.gram_matrix(problem, t, L, k, d, want_profiles=True, kernel=dev.KERNEL_BITSLICE)
        rows = np.arange(0, len(problem), 3)
        want_x = dev.cross_kernel(problem, rows, t, L, k, d, kernel=dev.KERNEL_BITSLICE)
        monkeypatch.setenv("GKM_COL_CHUNK", chunk)
        got = dev.gram_matrix(problem, t, L, k, d, want_profiles=True, kernel=dev.KERNEL_BITSLICE)
        got_x = dev.cross_kernel(problem, rows, t, L, k, d, kernel=dev.KERNEL_BITSLICE)
        assert torch.equal(got["P"], want["P"]) and torch.equal(got["K"], want["K"]) and torch.equal(got["sqnorm"], want["sqnorm"])
        assert torch.equal(got_x["K"], want_x["K"]) and torch.equal(got_x["sqnorm"], want_x["sqnorm"])   # full rows + diagonal band


@pytest.mark.parametrize("kernel", ["bitslice", "direct"])
def test_packed_row_slabs_hold_the_same_cells(dev, quirk_seqs, kernel):
    """gkmhip_gram_rows_packed (the send buffer of the multi-GPU all-gather): row rows[i] at G + row_off[i], columns
    0..rows[i] only, rows back to back -- the same raw values as gkmhip_gram_rows writes at full width, nothing
    outside the rows' own cells touched, for both kernels, an arbitrary ascending row subset and ragged lengths."""
    import torch
    from gkmqc_amd import sharding
    seqs, _ = quirk_seqs
    ragged = helpers.synth_codes(150, 150, 300, (150, 700))
    for problem, (t, L, k, d) in ((seqs, (4, 11, 7, 3)), (ragged, (2, 10, 6, 3))):
        n = len(problem)
        ctx = dev.GramContext(t, L, k, d)
        try:
            ctx.set_kernel(dev.KERNEL_BITSLICE if kernel == "bitslice" else dev.KERNEL_DIRECT)
            stream = torch.cuda.current_stream().cuda_stream
            ctx.set_sequences(problem, stream)
            full = torch.zeros((n, n), dtype=torch.float64, device="cuda")
            ctx.gram_rows(np.arange(n), full.data_ptr(), n, None, 0, False, stream)
            rows = np.array(sorted(set(np.random.default_rng(3).integers(0, n, n // 2).tolist()) | {0, n - 1}), dtype=np.int32)
            off = sharding.packed_row_offsets(rows) + 5           # (not at the start of the buffer)
            slab = torch.full((int(off[-1]) + 7,), -3.25, dtype=torch.float64, device="cuda")
            ctx.gram_rows_packed(rows, slab.data_ptr(), off, stream)
            torch.cuda.synchronize()
            got, want = slab.cpu().numpy(), full.cpu().numpy()
            assert (got[:5] == -3.25).all() and (got[int(off[-1]):] == -3.25).all()
            for i, a in enumerate(rows):
                assert (got[off[i]:off[i] + a + 1] == want[a, :a + 1]).all(), (kernel, a)
            bad = off.copy()
            bad[1] = bad[0] + rows[0]                              # row 1 would overlap row 0's last cell
            with pytest.raises(dev.GkmError):
                ctx.gram_rows_packed(rows, slab.data_ptr(), bad, stream)
        finally:
            ctx.close()


def test_one_process_assembly_keeps_its_buffers(dev):
    """gkmhip_gram_allgather keeps each rank's slab, gathered slabs, gather index, streams and events between calls of
    the same shape (bin/gkmqc.py asks for ~20 matrices of one size per run; hipMalloc / hipFree synchronise the
    device): the second call makes no hipMalloc, a call of another shape rebuilds them, and the per-rank HIP-event
    statistics of a call are there to read."""
    import torch
    lib = dev.load()
    a = helpers.synth_codes(200, 200, 300)
    b = helpers.synth_codes(120, 130, 300, (150, 400))
    one_a = dev.gram_matrix(a, 4, 11, 7, 3)["K"]
    one_b = dev.gram_matrix(b, 4, 11, 7, 3)["K"]
    lib.gkmhip_release_comms()
    n0 = lib.gkmhip_allgather_alloc_count()
    first = dev.gram_matrix_multi(a, 4, 11, 7, 3, devices=[0, 0], chunks=3)
    n1 = lib.gkmhip_allgather_alloc_count()
    assert n1 > n0 and torch.equal(first["K"][1], one_a)
    again = dev.gram_matrix_multi(a, 4, 11, 7, 3, devices=[0, 0], chunks=3)      # fresh contexts, same shape
    assert lib.gkmhip_allgather_alloc_count() == n1, "the second call of the same shape allocated device memory"
    assert torch.equal(again["K"][0], one_a) and torch.equal(again["K"][1], one_a)
    st = dev.allgather_stats()
    assert st["ranks"] == 2 and st["chunks"] == 3 and st["transport"] == "p2p"
    assert min(st["kernel_ms"]) > 0 and min(st["transfer_ms"]) > 0 and min(st["assemble_ms"]) > 0
    n = len(a)
    assert sum(st["comparisons"]) == 2.0 * 290 * 290 * (n * (n + 1) / 2)        # every (a, j <= a) pair exactly once
    other = dev.gram_matrix_multi(b, 4, 11, 7, 3, devices=[0, 0], chunks=3)      # another n: rebuilt, still right
    assert lib.gkmhip_allgather_alloc_count() > n1 and torch.equal(other["K"][1], one_b)
    back = dev.gram_matrix_multi(a, 4, 11, 7, 3, devices=[0, 0, 0], chunks=2)    # another rank count
    assert all(torch.equal(K, one_a) for K in back["K"])
    lib.gkmhip_release_comms()
    after = dev.gram_matrix_multi(a, 4, 11, 7, 3, devices=[0, 0], chunks=3)      # released: allocates again, still right
    assert torch.equal(after["K"][0], one_a)
    lib.gkmhip_release_comms()


def test_one_process_assembly_through_rccl(dev, monkeypatch):
    """The RCCL binding itself (dlopen, ncclCommInitAll, ncclAllGather on the transfer stream) on the
    one device a test box has: a one-rank communicator; more ranks need more GPUs (the driver's node)."""
    import torch
    monkeypatch.setenv("GKM_ALLGATHER", "rccl")
    problem = helpers.synth_codes(150, 150, 300)
    one = dev.gram_matrix(problem, 4, 11, 7, 3)["K"]
    for _ in range(2):   # second call: cached communicator
        res = dev.gram_matrix_multi(problem, 4, 11, 7, 3, devices=[0])
        assert res["transport"] == "rccl"
        assert torch.equal(res["K"][0], one)
    # the CHUNKED sequence (chunk c's all-gather on the transfer stream beside the kernel of chunk c + 1 on the compute
    # stream, the host-side agreement before and after every collective) through the same one-rank communicator
    for chunks in (2, 3):
        res = dev.gram_matrix_multi(problem, 4, 11, 7, 3, devices=[0], chunks=chunks)
        assert res["transport"] == "rccl" and torch.equal(res["K"][0], one)
        st = dev.allgather_stats()
        assert st["chunks"] == chunks and st["transport"] == "rccl" and st["transfer_ms"][0] > 0
    with pytest.raises(dev.GkmError):   # RCCL refuses one device twice; the call must fail cleanly, not hang
        dev.gram_matrix_multi(problem, 4, 11, 7, 3, devices=[0, 0])
    dev.load().gkmhip_release_comms()


def test_one_process_multi_gpu_assembly_at_full_size(dev):
    """The configs[3] stand-in (313 tiles, ~2 MB of launch tables per call) through two contexts on one
    GPU, each alternating its launches between two streams.  Two races showed only at this size: an
    asynchronous upload from a local host vector still in flight when the vector was freed (the launch
    tables now travel through pinned buffers that outlive the call), and the strand tables built by the
    first launch on one stream being read by the second launch on the other stream (they are now built,
    and complete, when the sequences are uploaded)."""
    import torch
    from gkmqc_amd import synth
    seqs = [dev.encode(x) for x in synth.make_peak_sequences(11, 5000, 600, True) +
            synth.make_peak_sequences(12, 5000, 600, False)]
    one = dev.gram_matrix(seqs, 4, 10, 6, 3)["K"]
    for trial in range(2):
        res = dev.gram_matrix_multi(seqs, 4, 10, 6, 3, devices=[0, 0])
        for g, K in enumerate(res["K"]):
            bad = (K != one) | torch.isnan(K)
            rows = torch.nonzero(bad.any(dim=1)).flatten().cpu().numpy()
            assert len(rows) == 0, "trial %d copy %d: %d cells in %d rows differ (rows %s .. %s), %d NaN" % (
                trial, g, int(bad.sum()), len(rows), rows[:8], rows[-4:], int(torch.isnan(K).sum()))
        del res


@pytest.mark.parametrize("var,value,ok", [("GKM_DEVICES", "0,x", False), ("GKM_DEVICES", "7,", False), ("GKM_DEVICES", "99", False),
                                          ("GKM_DEVICE", "-1", False), ("GKM_DEVICE", "two", False), ("GKM_DEVICE", "0", True),
                                          ("GKM_DEVICES", "1", True), ("GKM_DEVICES", "all", True)])
def test_device_selection_is_validated(dev, monkeypatch, var, value, ok):
    """With a GPU present: values that name no device of the node are errors (atoi used to turn "0,x" into
    device 0 twice), valid ones work, and the caller's current device is left alone."""
    import torch
    monkeypatch.delenv("GKM_DEVICES", raising=False)
    monkeypatch.delenv("GKM_DEVICE", raising=False)
    monkeypatch.setenv(var, value)
    n = 45
    kmat = np.full((n, n), -7.0)
    rows = (kmat.ctypes.data + np.arange(n) * kmat.strides[0]).astype(np.uintp)
    sizes = np.full(2, -1, dtype=np.int32)
    opt = dev.gkmOpt(4, 11, 7, 3, 50, 50.0, 1.0, helpers.QUIRK_POS.encode(), helpers.QUIRK_NEG.encode(), 1, 0)
    before = torch.cuda.current_device()
    rc = dev.load().gkm_main_pywrapper(ctypes.byref(opt), rows.ctypes.data, sizes.ctypes.data)
    assert torch.cuda.current_device() == before
    if ok:
        assert rc == 0 and kmat[n - 1, n - 1] == 1.0
    else:
        assert rc != 0 and (kmat == -7.0).all() and (sizes == -1).all()
