#!/usr/bin/env python3
"""A/B timing of the drop-in call gkm_main_pywrapper under different environment knobs (GPU box): each setting in a
process of its own, `--calls` calls (FASTA on disk -> the caller's pageable numpy rows), first call and the warm minimum.

    python3 tools/boundary_ab.py [--workload c2|peaks|c3|c5] [--calls 5] [--rounds 2] [--trace]
        [--settings "default" "GKM_KEEP_DEVICE=0" "GKM_BLOCK_FRACTIONS=0.5,0.25,0.125,0.0625,0.03" "GKM_EQUAL_BLOCKS=1"]

GKM_KEEP_DEVICE=0 is round 3's behaviour (context and matrix created and freed per call), GKM_BLOCK_FRACTIONS=0.5,... its
block schedule (halvings), GKM_EQUAL_BLOCKS=1 round 1's.  Worker mode: --worker (internal).
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(args):
    import numpy as np
    import bench
    from gkmqc_amd import device
    a = bench.parse_args(["--workload", args.workload])
    tmp = tempfile.mkdtemp(prefix="gkm_bab_")
    pf, nf = bench.write_problem_files(a, a.n_pos, a.n_neg, tmp)
    n = a.n_pos + a.n_neg
    kmat = np.zeros((n, n))
    rows = (kmat.ctypes.data + np.arange(n) * kmat.strides[0]).astype(np.uintp)
    sizes = np.zeros(2, dtype=np.int32)
    opt = device.gkmOpt(a.kernel_type, a.L, a.k, a.d, 50, 50.0, 1.0, os.fsencode(pf), os.fsencode(nf),
                        args.threads or bench.host_cores(), 3 if args.trace else 0)
    lib = device.load()
    walls = []
    for _ in range(args.calls):
        t0 = time.perf_counter()
        rc = lib.gkm_main_pywrapper(ctypes.byref(opt), rows.ctypes.data, sizes.ctypes.data)
        walls.append((time.perf_counter() - t0) * 1e3)
        assert rc == 0
    print(json.dumps({"first_ms": walls[0], "warm_min_ms": min(walls[1:]), "warm_ms": walls[1:],
                      "checksum": float(kmat[n - 1, :64].sum())}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--calls", type=int, default=5)
    ap.add_argument("--threads", type=int, default=0, help="the call's nthreads (the caller's -@); 0 = the cores of the box")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--trace", action="store_true", help="GKM_TRACE=1 and verbosity 3: the call's own phase timings on stderr")
    ap.add_argument("--settings", nargs="*", default=["default", "GKM_KEEP_DEVICE=0", "GKM_BLOCK_FRACTIONS=0.5,0.25,0.125,0.0625,0.03",
                                                      "GKM_EQUAL_BLOCKS=1"])
    args = ap.parse_args()
    if args.worker:
        return worker(args)
    res = {s: [] for s in args.settings}
    for r in range(args.rounds):
        for setting in args.settings:
            env = dict(os.environ)
            for kv in ([] if setting == "default" else setting.split()):
                k, v = kv.split("=", 1)
                env[k] = v
            if args.trace:
                env["GKM_TRACE"] = "1"
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", "--workload", args.workload,
                                "--calls", str(args.calls), "--threads", str(args.threads)] + (["--trace"] if args.trace else []),
                               env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            if p.returncode:
                print("%s: FAILED\n%s" % (setting, p.stderr.decode()[-1500:]), flush=True)
                continue
            lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
            d = json.loads(lines[-1])
            res[setting].append(d)
            print("round %d %-40s first %.1f ms, warm %s" % (r, setting, d["first_ms"], ["%.1f" % x for x in d["warm_ms"]]), flush=True)
            if args.trace:
                err = [ln for ln in p.stderr.decode().splitlines() if "gkmhip_gram_to_host_rows" in ln or "pieces (rows" in ln]
                dbg = [ln for ln in p.stdout.decode().splitlines() if "timing: read" in ln]
                print("\n".join("    " + ln for ln in err[-2:] + dbg[-1:]), flush=True)
    print("\n%-44s %-14s %-14s" % ("setting", "first call ms", "warm min ms"))
    for setting in args.settings:
        if res[setting]:
            print("%-44s %-14.1f %-14.1f" % (setting, min(d["first_ms"] for d in res[setting]), min(d["warm_min_ms"] for d in res[setting])))
    sums = {s: res[s][0]["checksum"] for s in args.settings if res[s]}
    print("checksums equal: %s" % (len(set(sums.values())) == 1))


if __name__ == "__main__":
    main()
