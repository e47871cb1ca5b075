"""The CPU oracle (oracle/gkm_oracle.c) against fixtures produced by the compiled reference."""
import ctypes

import numpy as np
import pytest

from tests import helpers


@pytest.fixture(scope="module")
def O(built):
    from oracle import oracle
    return oracle


def test_mismatch_weights_bit_identical(O):
    for (t, L, k, d), ref in helpers.golden_weights().items():
        got = O.mismatch_weights(t, L, k)[: d + 1]
        assert got.tobytes() == ref.tobytes(), (t, L, k, d)


def test_known_answers_from_survey(O):
    # %.17g values printed from the reference
    assert ["%.17g" % v for v in O.mismatch_weights(2, 10, 6)[:4]] == [
        "0.16959830600535497", "0.05573480820748955", "0.014950497890822588", "0.0027788987208623426"]
    assert list(O.mismatch_weights(0, 11, 7)[:4]) == [330.0, 120.0, 36.0, 8.0]
    assert "%.17g" % O.mismatch_weights(1, 11, 7)[3] == "-0.0023174285888671905"


def test_position_weights(O):
    cases, lens, _ = helpers.quirks_expected()
    for c in cases:
        for i, ln in enumerate(lens):
            n = int(ln) - c["L"] + 1
            got = O.position_weights(c["kernel_type"], n, c["M"], c["H"])
            assert (got == c["wt"][i, :n]).all(), (c["idx"], i)
    w = O.position_weights(4, 290, 50, 50.0)  # C2 shape
    assert int(w.sum()) == 6389 and w[0] == 7 and w[145] == 50


def test_fasta_reader_quirks(O):
    seqs, n_pos, n_invalid, n_trunc = O.read_problem(helpers.QUIRK_POS, helpers.QUIRK_NEG)
    _, lens, npos = helpers.quirks_expected()
    assert n_pos == npos and len(seqs) == len(lens)
    assert [len(s) for s in seqs] == list(lens)
    assert n_trunc == 1 and max(lens) == 2047
    assert n_invalid > 0


@pytest.mark.parametrize("idx", range(11))
def test_quirks_profiles_and_kernel(O, idx):
    cases, lens, npos = helpers.quirks_expected()
    c = cases[idx]
    opt = O.make_opt(c["kernel_type"], c["L"], c["k"], c["d"], c["M"], c["H"], c["gamma"],
                     helpers.QUIRK_POS, helpers.QUIRK_NEG)
    r = O.gram(opt, want_profiles=True, nthreads=8)
    n = r["n"]
    il = np.tril_indices(n)
    assert (r["P"][il] == c["P"][il]).all(), "integer mismatch profiles"
    assert helpers.max_rel_err(r["sqnorm"], c["sqnorm"]) < 1e-14
    assert helpers.max_rel_err(helpers.tril_pack(r["K"]), c["K"]) < 1e-12


def test_c1_block_and_cuts(O, tmp_path):
    from gkmqc_amd import synth
    z = helpers.synthetic_expected()
    # C1: a 48+48 sub-problem reproduces the matching block of the full 400x400 reference matrix
    sub = 48
    pos = synth.make_sequences(1, sub, 300)
    neg = synth.make_sequences(2, sub, 300)
    pf, nf = str(tmp_path / "p.fa"), str(tmp_path / "n.fa")
    synth.write_fasta(pf, pos, "p")
    synth.write_fasta(nf, neg, "n")
    r = O.gram(O.make_opt(2, 10, 6, 3, posfile=pf, negfile=nf), want_profiles=False, nthreads=8)
    full = helpers.tril_unpack(z["c1_full_K"], 400)
    idx = np.r_[0:sub, 200:200 + sub]
    want = full[np.ix_(idx, idx)]
    il = np.tril_indices(2 * sub, -1)
    assert helpers.max_rel_err(r["K"][il], want[il]) < 1e-12


def test_pywrapper_restatement_matches_reference_cells(O):
    """Same cells written as the reference: strict lower triangle + unit diagonal only."""
    cases, lens, npos = helpers.quirks_expected()
    c = cases[0]
    n = len(lens)
    opt = O.make_opt(c["kernel_type"], c["L"], c["k"], c["d"], c["M"], c["H"], c["gamma"],
                     helpers.QUIRK_POS, helpers.QUIRK_NEG, nthreads=3)
    rc, kmat, a, b = O.oracle_pywrapper(opt, n + 5)
    assert rc == 0 and (a, b) == (npos, n - npos)
    assert (np.triu(kmat, 1) == 0).all() and (kmat[n:, :] == 0).all()
    assert (np.diag(kmat)[:n] == 1.0).all()
    assert helpers.max_rel_err(helpers.tril_pack(kmat[:n, :n]), c["K"]) < 1e-12


@pytest.mark.skipif(not __import__("os").path.exists("/root/reference/src/libgkm.c"),
                    reason="the reference only exists in the development container")
def test_oracle_vs_live_reference(O, tmp_path):
    """Where the reference can be run: a fresh random problem through both."""
    if not O.have_ref():
        pytest.skip("oracle/_ref not built")
    from gkmqc_amd import synth
    pos = synth.make_sequences(11, 20, 120, (40, 400), 5)
    neg = synth.make_sequences(12, 25, 120, (40, 400), 6)
    pf, nf = str(tmp_path / "p.fa"), str(tmp_path / "n.fa")
    synth.write_fasta(pf, pos, "p")
    synth.write_fasta(nf, neg, "n")
    for (t, L, k, d) in [(4, 11, 7, 3), (5, 12, 8, 4), (0, 7, 4, 3)]:
        opt = O.make_opt(t, L, k, d, 50, 50.0, 1.5, pf, nf, nthreads=4)
        ref = O.ref_profiles(opt)
        mine = O.gram(opt, nthreads=8)
        il = np.tril_indices(ref["n"])
        assert (mine["P"][il] == ref["P"][il]).all()
        rc, kref, _, _ = O.ref_pywrapper(opt, ref["n"])
        assert rc == 0
        assert helpers.max_rel_err(helpers.tril_pack(mine["K"]), helpers.tril_pack(kref)) < 1e-12


def test_batch_rows_fixture_is_the_symmetric_completion(O, tmp_path):
    """The reference's batch-vs-set entry (gkmkernel_kernelfunc_batch, src/libgkm.c:1115-1153; fixture made by
    tests/golden/make_golden.py --only-batch) scores a row against EVERY sequence of the problem: its values are the
    symmetric completion of the triangle that gkm_main_pywrapper writes -- checked here with the CPU restatement, so
    that the GPU test of gkmhip_gram_rows_full compares with reference output whose meaning is pinned."""
    from gkmqc_amd import synth
    for c in helpers.batch_rows_expected():
        if c["name"] not in ("ragged_p1", "fixed300_p1"):     # one weighted and one RBF case: the oracle is brute force
            continue
        pf, nf = str(tmp_path / "p.fa"), str(tmp_path / "n.fa")
        synth.write_problem(pf, nf, c["n_support"], c["n_query"], c["length"], c["length_range"])
        opt = O.make_opt(c["kernel_type"], c["L"], c["k"], c["d"], c["M"], c["H"], c["gamma"], pf, nf)
        r = O.gram(opt, want_profiles=False, nthreads=8)
        K = np.tril(r["K"], -1) + np.tril(r["K"], -1).T
        n = r["n"]
        assert c["K"].shape == (c["n_query"], n)
        for i in range(c["n_query"]):
            a = c["n_support"] + i
            off = np.arange(n) != a
            assert helpers.max_rel_err(c["K"][i][off], K[a][off]) < 1e-12, (c["name"], a)
            assert abs(c["K"][i][a] - 1.0) < 1e-12      # G / sqnorm^2, not forced to 1.0 by this entry
