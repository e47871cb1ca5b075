"""Standalone `gkmkern` command-line front end."""
import os
import subprocess

import numpy as np
import pytest

from tests import helpers

CLI = os.path.join(helpers.ROOT, "gkmqc_amd", "bin", "gkmkern")


def test_cli_usage_and_loud_failure(built, tmp_path):
    import torch
    assert os.path.exists(CLI)
    assert subprocess.run([CLI], capture_output=True).returncode == 2
    r = subprocess.run([CLI, "-v", "0", str(tmp_path / "missing.fa"), helpers.QUIRK_NEG, str(tmp_path / "o")],
                       capture_output=True)
    assert r.returncode == 1
    if not torch.cuda.is_available():   # no GPU: the product must fail, not fall back to a CPU path
        r = subprocess.run([CLI, "-v", "0", helpers.QUIRK_POS, helpers.QUIRK_NEG, str(tmp_path / "o")],
                           capture_output=True)
        assert r.returncode == 1 and not os.path.exists(str(tmp_path / "o"))


@pytest.mark.gpu
def test_cli_text_and_binary_output(built, tmp_path):
    cases, lens, npos = helpers.quirks_expected()
    c = cases[0]
    n = len(lens)
    txt, binf = str(tmp_path / "k.txt"), str(tmp_path / "k.bin")
    base = [CLI, "-v", "0", "-t", str(c["kernel_type"]), "-l", str(c["L"]), "-k", str(c["k"]), "-d", str(c["d"])]
    subprocess.check_call(base + [helpers.QUIRK_POS, helpers.QUIRK_NEG, txt])
    subprocess.check_call(base + ["-b", helpers.QUIRK_POS, helpers.QUIRK_NEG, binf])
    rows = [ln.rstrip("\n").split("\t") for ln in open(txt)]
    assert len(rows) == n and all(len(r) == a + 1 and r[-1] == "1.0" for a, r in enumerate(rows))
    raw = open(binf, "rb").read()
    sizes = np.frombuffer(raw[:8], dtype=np.int32)
    tri = np.frombuffer(raw[8:], dtype=np.float64)
    assert tuple(sizes) == (npos, n - npos) and tri.size == n * (n + 1) // 2
    K = np.zeros((n, n))
    K[np.tril_indices(n)] = tri
    assert (np.diag(K) == 1.0).all()
    assert helpers.max_rel_err(helpers.tril_pack(K), c["K"]) < 1e-12
    got = np.array([float(v) for a, r in enumerate(rows) for v in r[:-1]])
    assert np.allclose(got, c["K"], rtol=1e-6, atol=0)      # "%e" keeps 7 significant digits
