#!/usr/bin/env python3
"""A/B timing of alternative builds of gkmkern_pylib.so (build_variants/lib_*.so) on the GPU box.

    python3 tools/kernel_ab.py [--rounds 2] [--libs build_variants/lib_a.so ...] [--workloads c2 peaks c5]

Every (library, round) runs in its own process (GKM_LIB_PATH selects the library at load time):
first a bit-exact check of the bit-sliced kernel against the general kernel on a mixed small
problem (a fast wrong kernel is of no interest), then the hot kernel's HIP-event time on each
workload.  Rounds are interleaved over the libraries (cdna guide §5.4 rule 24); the table at the end
gives min / median per library and workload.  Worker mode: --worker (internal).
"""
import argparse
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(args):
    import numpy as np
    import torch
    import bench
    from gkmqc_amd import device
    from tests import helpers
    out = {"lib": os.environ.get("GKM_LIB_PATH", "default")}
    # correctness first: ragged + fixed lengths, two parameter sets, profiles bit-exact vs the general kernel
    ok = True
    for seqs, (t, L, k, d) in ((helpers.synth_codes(100, 100, 300, (100, 700)), (4, 12, 8, 4)),
                               (helpers.synth_codes(150, 150, 300), (4, 11, 7, 3)),
                               ([device.encode(s) for s in bench.make_problem(bench.parse_args(
                                   ["--workload", "peaks", "--n-pos", "100", "--n-neg", "100"]))], (4, 10, 6, 3))):
        a = device.gram_matrix(seqs, t, L, k, d, want_profiles=True, kernel=device.KERNEL_BITSLICE)
        b = device.gram_matrix(seqs, t, L, k, d, want_profiles=True, kernel=device.KERNEL_DIRECT)
        il = np.tril_indices(len(seqs))
        ok = ok and bool((a["P"].cpu().numpy()[il] == b["P"].cpu().numpy()[il]).all()) and torch.equal(a["K"], b["K"])
    out["exact"] = ok
    for wl in args.workloads:
        a = bench.parse_args(["--workload", wl])
        seqs = [device.encode(s) for s in bench.make_problem(a)]
        n = len(seqs)
        ctx = device.GramContext(a.kernel_type, a.L, a.k, a.d, 50, 50.0, 1.0, 0)
        stream = torch.cuda.current_stream().cuda_stream
        ctx.set_sequences(seqs, stream)
        G = torch.zeros((n, n), dtype=torch.float64, device="cuda")
        ms = []
        for i in range(args.launches + 1):
            ctx.gram_rows(np.arange(n), G.data_ptr(), n, None, 0, False, stream)
            torch.cuda.synchronize()
            if i:
                ms.append(ctx.last_kernel_ms())
        out[wl] = ms
        out[wl + "_sum"] = float(G.sum().item())     # equal across libraries if the results are
        out[wl + "_kernel"] = ctx.last_kernel_name()
        ctx.close()
        del G
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--launches", type=int, default=3)
    ap.add_argument("--libs", nargs="*", default=None)
    ap.add_argument("--workloads", nargs="*", default=["c2", "peaks"])
    args = ap.parse_args()
    if args.worker:
        return worker(args)
    import numpy as np
    libs = args.libs if args.libs is not None else sorted(glob.glob(os.path.join(ROOT, "build_variants", "lib_*.so")))
    libs = ["default"] + [os.path.abspath(p) for p in libs]
    res = {lib: {wl: [] for wl in args.workloads} for lib in libs}
    sums, exact = {}, {}
    for r in range(args.rounds):
        for lib in libs:
            env = dict(os.environ)
            env.pop("GKM_LIB_PATH", None)
            if lib != "default":
                env["GKM_LIB_PATH"] = lib
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", "--launches", str(args.launches),
                                "--workloads"] + args.workloads, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            if p.returncode:
                print("%s: FAILED\n%s" % (lib, p.stderr.decode()[-1500:]), flush=True)
                exact[lib] = False
                continue
            d = json.loads(p.stdout.decode().strip().splitlines()[-1])
            exact[lib] = exact.get(lib, True) and d["exact"]
            for wl in args.workloads:
                res[lib][wl] += d[wl]
                sums.setdefault(wl, {})[lib] = d[wl + "_sum"]
            print("round %d %s: %s" % (r, os.path.basename(lib), {wl: ["%.1f" % x for x in d[wl]] for wl in args.workloads}),
                  flush=True)
    print("\n%-44s %s" % ("library", "  ".join("%-26s" % (wl + " ms min/median") for wl in args.workloads)))
    for lib in libs:
        cells = []
        for wl in args.workloads:
            v = res[lib][wl]
            same = sums.get(wl, {}).get(lib) == sums.get(wl, {}).get("default")
            cells.append("%-26s" % (("%.2f / %.2f%s" % (min(v), float(np.median(v)), "" if same else " SUM DIFFERS")) if v else "-"))
        print("%-44s %s  %s" % (os.path.basename(lib), "  ".join(cells), "exact" if exact.get(lib) else "NOT EXACT"))


if __name__ == "__main__":
    main()
