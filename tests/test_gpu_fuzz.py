"""A short randomised parity sweep (tools/fuzz_parity.py: random kernel type, L, k, d, M, H, length
distributions incl. duplicates / poly-A / reverse complements) as part of the GPU suite.  Longer
sweeps were run by hand on the box: about 2 000 cases / 4 000 kernel runs without a mismatch."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_random_parameter_sweep(built, seed, monkeypatch, capsys):
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", ["fuzz_parity.py", "--seconds", "12", "--seed", str(seed)])
    mod.main()                                   # raises SystemExit with the failing case on a mismatch
    assert "fuzz ok" in capsys.readouterr().out


@pytest.mark.gpu
def test_random_svm_problems(built, monkeypatch, capsys):
    """tools/fuzz_svm.py: random kernels, sizes, class balance, C, tol, duplicated samples -- the GPU
    C-SVC bit-identical to scikit-learn (about 3 300 cases in the hand-run sweeps of round 1)."""
    spec = importlib.util.spec_from_file_location("fuzz_svm", os.path.join(ROOT, "tools", "fuzz_svm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setenv("GKM_SVM_MAX_ITER", "300000")     # a pathological draw is skipped after ~2 s instead of running 10^7 iterations
    monkeypatch.setattr(sys, "argv", ["fuzz_svm.py", "--seconds", "15", "--seed", "21"])
    mod.main()
    assert "svm fuzz ok" in capsys.readouterr().out
