#!/usr/bin/env python3
"""Condense a tools/collect_profiles.sh output directory into the small files kept under
profiles/: the kernel-stats CSV as is, and one JSON with per-launch PMC averages of the
dominant kernel (HBM bytes corrected as MI355X_MICROARCH.md §HBM prescribes)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out, rnd = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "c2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (kernel_source_hash: the summary is only valid for the kernel code it was taken on)
dst = os.path.join(out, "summary")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, "%s_kernel_stats_%s.csv" % (rnd, workload)))

pmc = {}
launches = {}
kernel = None
for d in ("pmc_inst", "pmc_wait", "pmc_fetch", "pmc_write"):
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(float)
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            if "k_gram_" not in r["Kernel_Name"]:
                continue
            kernel = r["Kernel_Name"]
            per[r["Counter_Name"]] += float(r["Counter_Value"])
            disp[r["Counter_Name"]].add(r["Dispatch_Id"])
        for k, v in per.items():
            pmc[k] = v / max(1, len(disp[k]))
            launches[k] = len(disp[k])

summary = {"round": rnd, "workload": workload, "kernel": kernel, "per_launch": pmc, "launches_averaged": launches,
           "kernel_source_sha256": bench.kernel_source_hash(), "kernel_sources": list(bench.KERNEL_SOURCES)}
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    # rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE counts 128-B read requests as 64 B
    # (MI355X_MICROARCH.md §HBM): double the read side.
    summary["hbm_bytes_per_launch"] = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
    summary["hbm_note"] = "(2*FETCH_SIZE + WRITE_SIZE) * 1024, separate --pmc passes"
json.dump(summary, open(os.path.join(dst, "%s_pmc_%s.json" % (rnd, workload)), "w"), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True))
