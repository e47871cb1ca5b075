"""The drop-in call (row blocks, copy pieces, host scatter) against the device layer's one-launch matrix, bit for bit,
at sizes where the call cuts the matrix into several blocks and pieces.
python tools/check_boundary_blocks.py [--sizes 10000 3000 777]"""
import argparse
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=int, nargs="*", default=[10000, 3000, 777])
    a = ap.parse_args()
    from gkmqc_amd import gkmsvm, synth
    tmp = tempfile.mkdtemp()
    for n in a.sizes:
        for rng in (None, (150, 600)):
            pos, neg = os.path.join(tmp, "p.fa"), os.path.join(tmp, "n.fa")
            synth.write_problem(pos, neg, n // 2, n - n // 2, 300, rng)
            L, k, d = (11, 7, 3) if rng is None else (12, 8, 4)
            args = [4, L, k, d, 50, 50.0, 1.0, pos, neg, 8, 0]
            Kb, _, _ = gkmsvm.computeGkmKernel(args, backend="boundary")
            Kd, _, _ = gkmsvm.computeGkmKernel(args, backend="device")
            same = np.array_equal(Kb, Kd)
            print("n=%d lengths=%s: boundary == device layer: %s" % (n, rng or 300, same), flush=True)
            if not same:
                raise SystemExit(1)
    print("boundary blocks ok")


if __name__ == "__main__":
    main()
