#!/usr/bin/env python3
"""Does a collective's kernel get onto the GPU while a Gram kernel holds every wave slot?  (round 5)

    python3 tools/collective_beside_probe.py [--workload c2] [--ranks 8] [--mb 90]

No round has had a second GPU, so RCCL's all-gather has never run beside the Gram kernel.  What CAN be measured on one GPU is
the part that does not need a peer: the all-gather of chunk c is a kernel of a few large workgroups (RCCL: one per channel,
256-512 threads) that must START while the Gram kernel of chunk c+1 -- ~100 000 one-wave workgroups, 28 resident per CU,
7 of 8 wave slots and 504 of 512 VGPRs per SIMD -- is running.  This tool times a stand-in of that shape
(gkmhip_probe_copy: `blocks` workgroups of `threads` threads copying `mb` MB inside the device, on a second stream that is
proven to run beside the first) alone and beside a rank's Gram launch (enqueued `delay-ms` after it, when the kernel holds the whole device), and reports
when the copy ENDS relative to the launch: a copy that ends with the kernel has hidden behind nothing.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--mb", type=int, default=90)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--delay-ms", type=float, default=3.0)
    ap.add_argument("--reserve", type=int, default=0,
                    help="the Gram kernel runs on a stream that leaves this many CUs (8, 16, ..) to others (gkmhip_create_stream_reserving)")
    args = ap.parse_args()
    import ctypes
    import torch
    import bench
    from gkmqc_amd import device, sharding
    lib = device.load()
    a = bench.parse_args(["--workload", args.workload])
    seqs = [device.encode(s) for s in bench.make_problem(a)]
    n = len(seqs)
    dev = torch.device("cuda", 0)
    main_stream = torch.cuda.current_stream()
    ctx = device.GramContext(a.kernel_type, a.L, a.k, a.d, 50, 50.0, 1.0, 0)
    ctx.set_sequences(seqs, main_stream.cuda_stream)
    torch.cuda.synchronize(dev)
    if args.reserve:
        lib.gkmhip_create_stream_reserving.restype = ctypes.c_void_p
        lib.gkmhip_create_stream_reserving.argtypes = (ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int)
        h = lib.gkmhip_create_stream_reserving(None, 0, None, args.reserve)
        if not h:
            raise SystemExit("gkmhip_create_stream_reserving failed")
        main_stream = torch.cuda.ExternalStream(h, device=dev)
        print("the Gram kernel runs on a stream that leaves %d CUs free" % args.reserve, flush=True)
    rows = np.concatenate(sharding.chunked_layout(n, args.ranks, args.rank, 1)[0]).astype(np.int32)
    buf = torch.zeros((len(rows), n), dtype=torch.float64, device=dev)
    nbytes = args.mb << 20
    src = torch.ones(nbytes // 8, dtype=torch.float64, device=dev)
    dst = torch.zeros(nbytes // 8, dtype=torch.float64, device=dev)
    # a second stream PROVEN to run beside the first (include/gkm_hip.h), at the highest priority the device offers
    busy = (ctypes.c_void_p * 1)(main_stream.cuda_stream)
    beside = ctypes.c_int(0)
    lib.gkmhip_create_stream_beside_prio.restype = ctypes.c_void_p
    lib.gkmhip_create_stream_beside_prio.argtypes = (ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int)
    greatest = -1      # (HIP clamps a priority to the device's range; the numerically lowest is the highest)
    side_handle = lib.gkmhip_create_stream_beside_prio(busy, 1, ctypes.byref(beside), greatest)
    side = torch.cuda.ExternalStream(side_handle, device=dev)
    print("side stream: priority %d, runs beside the compute stream: %s" % (greatest, bool(beside.value)), flush=True)

    def gram():
        ctx.gram_rows(rows, buf.data_ptr(), n, None, 0, True, main_stream.cuda_stream)

    def copy(blocks, threads):
        if blocks == 0:     # hipMemcpyAsync device to device (what the one-process peer-copy path uses; SDMA or a blit kernel)
            with torch.cuda.stream(side):
                dst.copy_(src, non_blocking=True)
        else:
            rc = lib.gkmhip_probe_copy(dst.data_ptr(), src.data_ptr(), nbytes, blocks, threads, side_handle)
            if rc:
                raise SystemExit(lib.gkmhip_last_error().decode())

    for _ in range(3):
        gram()
    torch.cuda.synchronize(dev)
    ctx.kernel_timeline(True)
    for _ in range(args.reps):
        gram()
    torch.cuda.synchronize(dev)
    ms, k = ctx.kernel_timeline_ms()
    ctx.kernel_timeline(False)
    gram_alone = ms / k
    print("rank %d of %d, %d rows: Gram kernel alone %.2f ms" % (args.rank, args.ranks, len(rows), gram_alone), flush=True)
    print("%-34s %10s %14s %16s %14s" % ("copy of %d MB by" % args.mb, "alone ms", "beside: ms", "ends at (kernel", "kernel beside"))
    print("%-34s %10s %14s %16s %14s" % ("", "", "", "= 0 .. 1)", "ms"))
    for blocks, threads in ((0, 0), (16, 256), (32, 256), (64, 256), (32, 512), (64, 512), (128, 64), (1024, 64), (1024, 256)):
        ev = lambda: torch.cuda.Event(enable_timing=True)       # noqa: E731
        alone = []
        for _ in range(args.reps + 1):
            e0, e1 = ev(), ev()
            e0.record(side)
            copy(blocks, threads)
            e1.record(side)
            torch.cuda.synchronize(dev)
            alone.append(e0.elapsed_time(e1))
        both = []
        for _ in range(args.reps + 1):
            g0, g1, e0, e1 = ev(), ev(), ev(), ev()
            g0.record(main_stream)
            gram()
            g1.record(main_stream)
            t0 = time.perf_counter()        # the kernel has been running for a third of its time when the copy is enqueued
            while time.perf_counter() - t0 < args.delay_ms * 1e-3:
                pass
            e0.record(side)
            copy(blocks, threads)
            e1.record(side)
            torch.cuda.synchronize(dev)
            both.append((e0.elapsed_time(e1), g0.elapsed_time(e1) / g0.elapsed_time(g1), g0.elapsed_time(g1), g0.elapsed_time(e0)))
        both = np.array(both[1:])
        name = "hipMemcpyAsync (device to device)" if blocks == 0 else "%d workgroups x %d threads" % (blocks, threads)
        print("%-34s %10.2f %14.2f %16.2f %14.2f   (enqueued %.2f ms into the step)"
              % (name, min(alone[1:]), np.median(both[:, 0]), np.median(both[:, 1]), np.median(both[:, 2]), np.median(both[:, 3])),
              flush=True)

    # ---- the remedy: the copy becomes runnable when chunk A ends (it waits for A's event), chunk B's launch follows A on
    # the compute stream after a pause of p microseconds (gkmhip_pause_stream): who is on the device first?
    lib.gkmhip_pause_stream.restype = ctypes.c_int
    lib.gkmhip_pause_stream.argtypes = (ctypes.c_void_p, ctypes.c_int)
    parts = sharding.chunked_layout(n, args.ranks, args.rank, 2)[0]
    ra, rb = (np.asarray(p, dtype=np.int32) for p in parts)
    bufa = torch.zeros((len(ra), n), dtype=torch.float64, device=dev)
    bufb = torch.zeros((len(rb), n), dtype=torch.float64, device=dev)
    print("\ntwo chunks (%d + %d rows) on the compute stream, the copy (64 workgroups x 256 threads) waits for the first:" % (len(ra), len(rb)))
    print("%-22s %16s %22s %14s" % ("pause before chunk 2", "copy ms", "copy ends .. ms after", "both chunks ms"))
    print("%-22s %16s %22s %14s" % ("", "(alone %.2f)" % 0.10, "chunk 1's end", ""))
    for pause in (0, 10, 20, 50, 100):
        got = []
        for _ in range(args.reps + 1):
            g0, ga, g1, e0, e1 = ev(), ev(), ev(), ev(), ev()
            g0.record(main_stream)
            ctx.gram_rows(ra, bufa.data_ptr(), n, None, 0, True, main_stream.cuda_stream)
            ga.record(main_stream)
            side.wait_event(ga)
            e0.record(side)
            copy(64, 256)
            e1.record(side)
            if pause:
                lib.gkmhip_pause_stream(main_stream.cuda_stream, pause)
            ctx.gram_rows(rb, bufb.data_ptr(), n, None, 0, True, main_stream.cuda_stream)
            g1.record(main_stream)
            torch.cuda.synchronize(dev)
            got.append((e0.elapsed_time(e1), ga.elapsed_time(e1), g0.elapsed_time(g1)))
        got = np.array(got[1:])
        print("%-22s %16.2f %22.2f %14.2f" % ("%d us" % pause, np.median(got[:, 0]), np.median(got[:, 1]), np.median(got[:, 2])), flush=True)

    # ---- what a RESIDENT collective costs the kernel: workgroups that hold their wave slots and wait (for their peers, in
    # the real thing; for the clock here) from just before the kernel starts until `hold` ms later
    lib.gkmhip_probe_spin.restype = ctypes.c_int
    lib.gkmhip_probe_spin.argtypes = (ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p)
    print("\nthe rank's Gram kernel (alone %.2f ms) beside workgroups that were on the device first and wait:" % gram_alone)
    print("%-34s %12s %14s" % ("waiting workgroups", "held for ms", "kernel ms"))
    for blocks, threads, hold in ((0, 0, 0), (32, 256, 6), (64, 256, 6), (64, 512, 6), (128, 512, 6)):
        got = []
        for _ in range(args.reps + 1):
            g0, g1 = ev(), ev()
            if blocks:
                lib.gkmhip_probe_spin(blocks, threads, hold * 1000, side_handle)
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 2e-4:      # the waiting workgroups are resident before the kernel comes
                    pass
            ctx.kernel_timeline(True)
            g0.record(main_stream)
            gram()
            g1.record(main_stream)
            torch.cuda.synchronize(dev)
            ms, k = ctx.kernel_timeline_ms()
            ctx.kernel_timeline(False)
            got.append(ms / k)
        print("%-34s %12s %14.2f" % ("none" if not blocks else "%d x %d threads" % (blocks, threads), hold or "-", float(np.median(got[1:]))),
              flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
