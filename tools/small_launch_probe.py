#!/usr/bin/env python3
"""Why does 1/8 of the matrix take more than 1/8 of the time?  (round 5)

    python3 tools/small_launch_probe.py [--workload c2] [--ranks 8] [--rank 0]

Rank `rank`'s folded rows of a `ranks`-way split as ONE Gram launch on one GPU, timed with the launch's own HIP events:
  (a) launches enqueued back to back, no host wait in between (the GPU never idles: sustained clocks)
  (b) the same with a device synchronisation and a host pause of 0 / 0.3 / 1 / 3 / 10 ms before every launch (what a
      host-driven step looks like: the GPU idles in between)
  (the work-item order -- GKM_COL_CHUNK=0 / 1024 against the default -- was measured too: no difference, profiles/r5_small_launch_probe.txt)
and beside them the whole matrix in one launch / `ranks`.
"""
import argparse
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(args):
    import torch
    import bench
    from gkmqc_amd import device, sharding
    a = bench.parse_args(["--workload", args.workload])
    seqs = [device.encode(s) for s in bench.make_problem(a)]
    n = len(seqs)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    ctx = device.GramContext(a.kernel_type, a.L, a.k, a.d, 50, 50.0, 1.0, 0)
    ctx.set_sequences(seqs, stream)
    full = torch.zeros((n, n), dtype=torch.float64, device=dev)
    rows = np.concatenate(sharding.chunked_layout(n, args.ranks, args.rank, 1)[0]).astype(np.int32)
    buf = torch.zeros((len(rows), n), dtype=torch.float64, device=dev)

    def series(what, count, pause):
        """pause: None = enqueue back to back; else synchronise, then sleep `pause` seconds before every launch"""
        ctx.kernel_timeline(True)
        for _ in range(count):
            if pause is not None:
                torch.cuda.synchronize(dev)
                if pause > 0:
                    time.sleep(pause)
            what()
        torch.cuda.synchronize(dev)
        ms, k = ctx.kernel_timeline_ms()
        ctx.kernel_timeline(False)
        return ms / k

    whole = lambda: ctx.gram_rows(np.arange(n), full.data_ptr(), n, None, 0, False, stream)     # noqa: E731
    if args.blocks:
        # row ranges "a:b" (several joined by "+"): one launch each, back to back; work items = sum over its 64-row tiles of
        # (largest row + 1) columns -- every item is 64 lanes x one column, whatever its rows -- and the time per item
        series(whole, 2, None)
        w = series(whole, 4, None)
        tiles_all = (n + 63) // 64
        items_all = sum(min(64 * (t + 1), n) for t in range(tiles_all))
        print("whole matrix: %.2f ms, %d items, %.3f us per 1 000 items" % (w, items_all, w * 1e6 / items_all), flush=True)
        for spec in args.blocks:
            rr = np.concatenate([np.arange(int(x.split(":")[0]), int(x.split(":")[1])) for x in spec.split("+")]).astype(np.int32)
            items = sum(int(rr[min(i + 63, len(rr) - 1)]) + 1 for i in range(0, len(rr), 64))
            bb = torch.zeros((len(rr), n), dtype=torch.float64, device=dev)
            one = lambda: ctx.gram_rows(rr, bb.data_ptr(), n, None, 0, True, stream)            # noqa: E731
            series(one, 3, None)
            t = series(one, 12, None)
            print("rows %-22s %5d rows, %7d items: %.2f ms, %.3f us per 1 000 items (x %.3f of the whole matrix's)"
                  % (spec, len(rr), items, t, t * 1e6 / items, (t / items) / (w / items_all)), flush=True)
            del bb
        ctx.close()
        return
    part = lambda: ctx.gram_rows(rows, buf.data_ptr(), n, None, 0, True, stream)                # noqa: E731
    series(whole, 2, None)
    w = series(whole, 4, None)
    series(part, 3, None)
    b2b = series(part, 20, None)
    paused = {p: series(part, 20, p) for p in (0.0, 0.0003, 0.001, 0.003, 0.010)}
    print("%s%s: whole matrix %.2f ms -> / %d = %.2f ms; rank %d's rows (%d) in one launch: back to back %.2f ms (x %.3f); "
          "after a device synchronisation + a host pause of %s"
          % (args.workload, " GKM_COL_CHUNK=" + os.environ["GKM_COL_CHUNK"] if "GKM_COL_CHUNK" in os.environ else "",
             w, args.ranks, w / args.ranks, args.rank, len(rows), b2b, b2b / (w / args.ranks),
             ", ".join("%g ms: %.2f ms (x %.3f)" % (p * 1e3, v, v / (w / args.ranks)) for p, v in paused.items())), flush=True)
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--blocks", nargs="*", default=None, help='row ranges to time as one launch each, e.g. 9375:10000 0:625+9375:10000')
    args = ap.parse_args()
    if args.worker or args.blocks:
        return run(args)
    for env in ({},):
        for rank in sorted({args.rank, args.ranks // 2, args.ranks - 1}):
            e = dict(os.environ, **env)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", "--workload", args.workload, "--ranks", str(args.ranks),
                            "--rank", str(rank)], env=e)


if __name__ == "__main__":
    main()
