"""gkmqc_amd/gkmsvm.py (host-side mirror of the reference's scripts/gkmsvm.py) against numbers
returned by the reference's own module (tests/golden/make_golden_gkmsvm.py)."""
import json
import os

import numpy as np
import pytest

from tests import helpers

EXPECTED = json.load(open(os.path.join(helpers.GOLDEN, "gkmsvm_expected.json")))
POS = os.path.join(helpers.GOLDEN, "motif_pos.fa")
NEG = os.path.join(helpers.GOLDEN, "motif_neg.fa")


def _args_gkm(case):
    a = list(case["args_gkm"])
    a[7], a[8] = POS, NEG
    return a


@pytest.mark.parametrize("name", sorted(EXPECTED))
def test_cross_validate_matches_reference_on_cpu(built, name):
    """CV half of the mirror, fed with the oracle's matrix symmetrised the reference's way."""
    from gkmqc_amd import gkmsvm
    from oracle import oracle as O
    case = EXPECTED[name]
    t, L, k, d, M, H, g = case["args_gkm"][:7]
    r = O.gram(O.make_opt(t, L, k, d, M, H, g, POS, NEG), want_profiles=False, nthreads=8)
    kmat = np.maximum(np.tril(r["K"]), np.tril(r["K"]).T)
    assert (r["n_pos"], r["n"] - r["n_pos"]) == (case["n_pos"], case["n_neg"])
    got = kmat[case["sample_i"], case["sample_j"]]
    assert helpers.max_rel_err(got, np.array(case["sample_v"])) < 1e-9
    auc, std = gkmsvm.crossValidate(list(case["args_svm"]), kmat, case["n_pos"], case["n_neg"])
    assert abs(auc - case["auc_mean"]) < 1e-9 and abs(std - case["auc_std"]) < 1e-9


def test_plan_folds_follows_the_reference_seeding():
    """The folds a pipeline draws ahead of time are the reference's (scripts/gkmsvm.py:134-150): StratifiedKFold with
    shuffle and the SAME random_state for every repeat -- so a fixed seed repeats the same ncv folds (solved once, counted
    `repeats` times) and no seed draws ncv x repeats different ones; every fold is stratified and a partition."""
    from sklearn.model_selection import StratifiedKFold
    from gkmqc_amd import svmcv
    args = [1.0, 0.001, 0, 512, 5, 10, 0, 7, 1]
    plan = svmcv.plan_folds(args, 60, 90)
    assert plan["sizes"] == (60, 90) and plan["n_folds"] == 50 and len(plan["u_trains"]) == 5
    assert plan["which"] == [0, 1, 2, 3, 4] * 10
    y = plan["y"]
    assert (y[:60] == 1).all() and (y[60:] == 0).all()
    want = list(StratifiedKFold(n_splits=5, shuffle=True, random_state=7).split(np.zeros(150), y))
    for (tr, te), ptr, pte in zip(want, plan["u_trains"], plan["u_tests"]):
        assert np.array_equal(tr, ptr) and np.array_equal(te, pte)
        assert len(np.intersect1d(ptr, pte)) == 0 and len(ptr) + len(pte) == 150
        assert int(y[pte].sum()) == 12 and len(pte) == 30                      # stratified: 60/5 positives per test fold
    free = svmcv.plan_folds([1.0, 0.001, 0, 512, 5, 10, 0, -1, 1], 60, 90)       # -1 = no seed, as the reference's default
    assert free["n_folds"] == 50 and len(free["u_trains"]) > 5


def test_fast_estimation_is_rejected(built):
    from gkmqc_amd import gkmsvm
    with pytest.raises(NotImplementedError):
        gkmsvm.crossValidate([1.0, 0.001, 0, 512, 5, 1, 1, 7, 1], np.eye(4), 2, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["device", "boundary"])
@pytest.mark.parametrize("name", sorted(EXPECTED))
def test_compute_kernel_and_auc_on_gpu(built, name, backend):
    from gkmqc_amd import gkmsvm
    case = EXPECTED[name]
    kmat, n_pos, n_neg = gkmsvm.computeGkmKernel(_args_gkm(case), backend=backend)
    assert (n_pos, n_neg) == (case["n_pos"], case["n_neg"])
    assert kmat.shape == (n_pos + n_neg, n_pos + n_neg) and (kmat == kmat.T).all()
    tol = 1e-10 if case["args_gkm"][0] in (3, 5) else 1e-12   # RBF types use the device exp()
    assert helpers.max_rel_err(kmat[case["sample_i"], case["sample_j"]], np.array(case["sample_v"])) < tol
    assert abs(kmat.min() - case["kmat_min"]) < 1e-12          # type 1: negatives clamp to 0 as in the reference
    assert abs(kmat.sum() - case["kmat_sum"]) < 1e-9 * abs(case["kmat_sum"])
    auc, std = gkmsvm.crossValidate(list(case["args_svm"]), kmat, n_pos, n_neg)
    assert abs(auc - case["auc_mean"]) < 1e-9 and abs(std - case["auc_std"]) < 1e-9


@pytest.mark.gpu
def test_init_writes_the_eval_line(built, tmp_path):
    from gkmqc_amd import gkmsvm
    case = EXPECTED["wgkm_L10"]
    out = str(tmp_path / "run1")
    argv = ["-p", POS, "-n", NEG, "-w", out, "-s", "7", "-@", "2", "-v", "0", "-t", "4", "-L", "10", "-k", "6", "-d", "3",
            "-r", "2"]
    auc, std = gkmsvm.main(argv)
    line = open(out + ".gkmqc.eval.out").read().rstrip("\n").split("\t")
    assert line[0] == POS and line[1] == NEG and int(line[2]) == case["n_pos"]
    assert abs(float(line[3]) - case["auc_mean"]) < 1e-9 and abs(float(line[4]) - case["auc_std"]) < 1e-9


@pytest.mark.gpu
def test_solver_announces_its_launch(built):
    """init_many holds the next subset's Gram kernel back until the solver of the current one is about to be enqueued
    (its few big workgroups would otherwise wait for room): the hook is called exactly once, before the result exists,
    and the solver calls keep their device scratch (no hipMalloc / hipFree beside a running Gram kernel)."""
    from gkmqc_amd import gkmsvm
    case = EXPECTED["wgkm_L10"]
    K, n_pos, n_neg = gkmsvm.computeGkmKernel(_args_gkm(case), resident=True)
    calls = []
    auc, std = gkmsvm.crossValidate(list(case["args_svm"]), K, n_pos, n_neg, about_to_launch=lambda: calls.append(len(calls)))
    assert calls == [0]
    assert abs(auc - case["auc_mean"]) < 1e-12 and abs(std - case["auc_std"]) < 1e-12
    auc2, _ = gkmsvm.crossValidate(list(case["args_svm"]), K, n_pos, n_neg)      # pooled scratch reused
    assert auc2 == auc
    from gkmqc_amd import svmcv
    svmcv._lib().gkmsvm_release_cache()
    assert gkmsvm.crossValidate(list(case["args_svm"]), K, n_pos, n_neg)[0] == auc


@pytest.mark.gpu
def test_init_many_overlaps_and_matches_init(built, tmp_path):
    """Several subsets in a row, the cross-validation of one on a second stream beside the matrix of the
    next: same AUCs and the same eval lines as one `init` per subset."""
    from gkmqc_amd import gkmsvm
    case = EXPECTED["wgkm_L10"]
    base = ["-s", "7", "-v", "0", "-t", "4", "-L", "10", "-k", "6", "-d", "3", "-r", "2"]
    a1 = gkmsvm.build_parser().parse_args(["-p", POS, "-n", NEG, "-w", str(tmp_path / "seq")] + base)
    a2 = gkmsvm.build_parser().parse_args(["-p", POS, "-n", NEG, "-w", str(tmp_path / "ovl")] + base)
    pairs = [(POS, NEG), (NEG, POS), (POS, NEG), (NEG, POS)]
    want = [gkmsvm.init(p, n, a1) for p, n in pairs]
    got = gkmsvm.init_many(pairs, a2)
    assert got == want
    assert abs(got[0][0] - case["auc_mean"]) < 1e-12
    assert open(str(tmp_path / "seq") + ".gkmqc.eval.out").read() == open(str(tmp_path / "ovl") + ".gkmqc.eval.out").read()
    # the subsets dealt to several workers (here two on the box's one GPU, each with its own context and streams):
    # same numbers and the same lines in the order of `pairs`
    a3 = gkmsvm.build_parser().parse_args(["-p", POS, "-n", NEG, "-w", str(tmp_path / "farm")] + base)
    assert gkmsvm.init_many(pairs + pairs[:1], a3, gpus=[0, 0]) == want + want[:1]
    lines = open(str(tmp_path / "farm") + ".gkmqc.eval.out").read().splitlines()
    assert lines[:4] == open(str(tmp_path / "seq") + ".gkmqc.eval.out").read().splitlines() and len(lines) == 5


@pytest.mark.gpu
@pytest.mark.parametrize("gpus", [1, [0, 0]])
def test_config4_stand_in_at_full_size(built, tmp_path, gpus):
    """BASELINE configs[3] at its real size: ONE `gkmqc.py evaluate` subset (5 000 peak-like positives +
    5 000 matched nulls x 600 bp, wgkm L=10 k=6 d=3, 5-fold CV; reference bin/gkmqc.py:150-154,181-185,
    213-215).  Expected numbers come from the reference's OWN module on the same files
    (tests/golden/make_golden_gkmsvm.py --c4: compiled reference for the matrix, its scikit-learn harness
    for the AUC): the symmetrised matrix must be bit-identical, the AUC equal.  gpus=[0, 0]: the same
    through the one-process multi-GPU entry (two contexts on the box's one GPU)."""
    import hashlib
    import torch
    from gkmqc_amd import gkmsvm, synth
    case = json.load(open(os.path.join(helpers.GOLDEN, "gkmsvm_expected_c4.json")))["c4_peaks"]
    pf, nf = str(tmp_path / "p.fa"), str(tmp_path / "n.fa")
    synth.write_peak_problem(pf, nf, 5000, 5000, 600)
    a = list(case["args_gkm"])
    a[7], a[8] = pf, nf
    K, n_pos, n_neg = gkmsvm.computeGkmKernel(a, resident=True, gpus=gpus)
    assert (n_pos, n_neg) == (case["n_pos"], case["n_neg"])
    auc, std = gkmsvm.crossValidate(list(case["args_svm"]), K, n_pos, n_neg)      # every fold on the GPU
    assert abs(auc - case["auc_mean"]) < 1e-12 and abs(std - case["auc_std"]) < 1e-12
    kmat = K.cpu().numpy()
    del K
    torch.cuda.empty_cache()
    assert helpers.max_rel_err(kmat[case["sample_i"], case["sample_j"]], np.array(case["sample_v"])) < 1e-12
    assert hashlib.sha256(np.ascontiguousarray(kmat).tobytes()).hexdigest() == case["kmat_sha256"], \
        "matrix differs from the reference's in the last bits"


@pytest.mark.gpu
def test_config4_as_the_pipeline_runs_it(built, tmp_path):
    """BASELINE configs[3] the way `gkmqc.py evaluate` drives it (reference bin/gkmqc.py:150-154,213-216,338-343): one
    subset after the other, each 5 000 peaks of 600 bp + 5 000 matched nulls, wgkm L=10 k=6 d=3, 5-fold x 10 repeats
    (`-x 5 -r 10`).  Three full-size subsets through init_many (cross-validation of one subset beside the matrix of the
    next, the device context kept from subset to subset): the same AUCs and the same three eval lines as one init per
    subset, and for the subset the fixture was made from, the AUC the reference's own module returned at 10 repeats
    (tests/golden/make_golden_gkmsvm.py --c4)."""
    from gkmqc_amd import gkmsvm, synth
    case = json.load(open(os.path.join(helpers.GOLDEN, "gkmsvm_expected_c4.json")))["c4_peaks_r10"]
    pairs = []
    for s, (sp, sn) in enumerate(((11, 12), (21, 22), (31, 32))):      # (11, 12): the fixture's subset
        pf, nf = str(tmp_path / ("p%d.fa" % s)), str(tmp_path / ("n%d.fa" % s))
        synth.write_peak_problem(pf, nf, 5000, 5000, 600, seed_pos=sp, seed_neg=sn)
        pairs.append((pf, nf))
    C, eps, shrinking, cache, ncv, repeats, _fast, seed, _procs = case["args_svm"]
    assert (ncv, repeats) == (5, 10)
    base = ["-s", str(seed), "-v", "0", "-t", "4", "-L", "10", "-k", "6", "-d", "3", "-x", str(ncv), "-r", str(repeats),
            "-C", str(C), "-e", str(eps), "-u", str(shrinking), "-c", str(cache)]
    a1 = gkmsvm.build_parser().parse_args(["-p", "x", "-n", "y", "-w", str(tmp_path / "seq")] + base)
    a2 = gkmsvm.build_parser().parse_args(["-p", "x", "-n", "y", "-w", str(tmp_path / "many")] + base)
    want = [gkmsvm.init(p, n, a1) for p, n in pairs]
    got = gkmsvm.init_many(pairs, a2)
    assert got == want
    assert abs(got[0][0] - case["auc_mean"]) < 1e-12 and abs(got[0][1] - case["auc_std"]) < 1e-12
    lines = open(str(tmp_path / "many") + ".gkmqc.eval.out").read().splitlines()
    assert len(lines) == 3 and lines == open(str(tmp_path / "seq") + ".gkmqc.eval.out").read().splitlines()
    assert len({ln.split("\t")[3] for ln in lines}) == 3               # three different subsets, three different AUCs


@pytest.mark.gpu
@pytest.mark.parametrize("n,rng", [(4400, None), (2000, (150, 600))])
def test_boundary_blocks_and_pieces_against_one_launch(built, tmp_path, monkeypatch, n, rng):
    """The drop-in call cuts the matrix into geometrically shrinking row blocks (one Gram launch each) that travel
    to the caller's rows in staging-sized pieces; at 4 400 rows the first block needs two pieces.  Bit for bit the
    matrix the device layer computes in one launch, also with the equal-area blocks kept for A/B runs."""
    from gkmqc_amd import gkmsvm, synth
    pos, neg = str(tmp_path / "p.fa"), str(tmp_path / "n.fa")
    synth.write_problem(pos, neg, n // 2, n - n // 2, 300, rng)
    L, k, d = (11, 7, 3) if rng is None else (12, 8, 4)
    args = [4, L, k, d, 50, 50.0, 1.0, pos, neg, 8, 0]
    want, n_pos, n_neg = gkmsvm.computeGkmKernel(args, backend="device")
    assert (n_pos, n_neg) == (n // 2, n - n // 2)
    got, _, _ = gkmsvm.computeGkmKernel(args, backend="boundary")
    assert np.array_equal(got, want)
    monkeypatch.setenv("GKM_EQUAL_BLOCKS", "1")
    got, _, _ = gkmsvm.computeGkmKernel(args, backend="boundary")
    assert np.array_equal(got, want)
    monkeypatch.delenv("GKM_EQUAL_BLOCKS")
    monkeypatch.setenv("GKM_BLOCK_FRACTIONS", "0.3,0.3,0.2,0.1")     # any cut of the rows into blocks gives the same matrix
    got, _, _ = gkmsvm.computeGkmKernel(args, backend="boundary")
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_boundary_keeps_its_device_memory_between_calls(built, tmp_path, monkeypatch):
    """The drop-in call keeps its context and the n x n device matrix for the next call (bin/gkmqc.py:341-343 makes ~20
    per run): a second call of the same parameters finds them (gkm_device_cache_hits), a smaller problem reuses the
    matrix, other parameters rebuild the context, GKM_KEEP_DEVICE=0 and gkm_release_device_cache() give the memory
    back -- and every call returns the matrix the device layer computes in one launch."""
    import ctypes
    from gkmqc_amd import device, gkmsvm, synth
    lib = device.load()
    lib.gkm_device_cache_hits.restype = ctypes.c_long
    lib.gkm_release_device_cache.restype = None
    big = (str(tmp_path / "p.fa"), str(tmp_path / "n.fa"))
    small = (str(tmp_path / "p2.fa"), str(tmp_path / "n2.fa"))
    synth.write_problem(big[0], big[1], 700, 700, 300, None)
    synth.write_problem(small[0], small[1], 300, 200, 300, (150, 500), seed_pos=5, seed_neg=6)
    lib.gkm_release_device_cache()
    h0 = lib.gkm_device_cache_hits()
    for files, (L, k, d), hit in ((big, (11, 7, 3), 0), (big, (11, 7, 3), 1), (small, (11, 7, 3), 1),
                                  (small, (10, 6, 3), 0), (big, (10, 6, 3), 0), (big, (10, 6, 3), 1)):
        args = [4, L, k, d, 50, 50.0, 1.0, files[0], files[1], 4, 0]
        want, _, _ = gkmsvm.computeGkmKernel(args, backend="device")
        got, _, _ = gkmsvm.computeGkmKernel(args, backend="boundary")
        assert np.array_equal(got, want), (files, L)
        h1 = lib.gkm_device_cache_hits()
        assert h1 - h0 == hit, (files, L, h1 - h0)
        h0 = h1
    monkeypatch.setenv("GKM_KEEP_DEVICE", "0")
    args = [4, 10, 6, 3, 50, 50.0, 1.0, big[0], big[1], 4, 0]
    got, _, _ = gkmsvm.computeGkmKernel(args, backend="boundary")
    assert np.array_equal(got, gkmsvm.computeGkmKernel(args, backend="device")[0]) and lib.gkm_device_cache_hits() == h0
    monkeypatch.delenv("GKM_KEEP_DEVICE")
    lib.gkm_release_device_cache()
    got, _, _ = gkmsvm.computeGkmKernel(args, backend="boundary")
    assert lib.gkm_device_cache_hits() == h0
