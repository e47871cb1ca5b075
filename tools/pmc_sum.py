#!/usr/bin/env python3
"""Per-launch averages of the rocprofv3 counters of the hot kernel under a tools/pmc_ab.sh output directory."""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)):
    per, disp, kernel = collections.defaultdict(float), collections.defaultdict(set), None
    for r in csv.DictReader(open(f)):
        if "k_gram_" not in r["Kernel_Name"]:
            continue
        kernel = r["Kernel_Name"]
        per[r["Counter_Name"]] += float(r["Counter_Value"])
        disp[r["Counter_Name"]].add(r["Dispatch_Id"])
    for k in sorted(per):
        print("%-24s %.6e  (%d launches)  %s" % (k, per[k] / max(1, len(disp[k])), len(disp[k]), kernel))
