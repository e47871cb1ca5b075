#!/bin/bash
# Does the re-streaming of the column bit-plane tables (SB: 51 MB at config 2, read once per work item through the scalar
# cache, 5.45 GB of L2 fills per launch = 12x the algorithmic bytes) cost anything when several contexts compete for one
# GPU's caches?  One GPU, 1 / 2 / 4 contexts of the one-process multi-GPU entry (each context computes its folded row
# blocks in 4 chunks, alternating between two streams, all on the same device), rocprofv3 kernel trace for the kernels'
# durations and a separate FETCH_SIZE pass for the L2 fill bytes.  VERDICT r2 "Next" #7.
#   gpurun -- tools/sb_traffic.sh r3 [workload]     -> gpurun_out/sb_traffic_<round>.txt (copy to profiles/)
set -u
R=${1:-r3}
WL=${2:-c2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/sb_traffic_${R}
mkdir -p "$OUT"
TXT=gpurun_out/sb_traffic_${R}.txt
echo "# tools/sb_traffic.sh $R $WL: contexts sharing ONE MI355X; hot kernel per-launch averages" > "$TXT"
rocprofv3 -L 2>/dev/null | grep -i -E "mall|dram|hbm|TCC_EA0_RDREQ|TCC_HIT|TCC_MISS|TCC_REQ" | head -40 > "$OUT/counters_avail.txt"
for G in 1 2 4; do
  if [ "$G" = 1 ]; then ARGS="bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-also";
  else ARGS="bench.py --workload $WL --gpus $G --assembly cabi --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end"; fi
  GKM_BENCH_SHARE_GPU=1 rocprofv3 --kernel-trace --stats -d "$OUT/stats_$G" --output-format csv -- python3 $ARGS > "$OUT/bench_$G.json" 2> "$OUT/bench_$G.err" || exit 1
  GKM_BENCH_SHARE_GPU=1 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch_$G" --output-format csv -- python3 $ARGS > /dev/null 2> "$OUT/fetch_$G.err" || exit 1
  GKM_BENCH_SHARE_GPU=1 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d "$OUT/tcc_$G" --output-format csv -- python3 $ARGS > /dev/null 2> "$OUT/tcc_$G.err" || echo "(no TCC_* pass for $G contexts)" >> "$TXT"
  python3 - "$OUT" "$G" >> "$TXT" <<'PY'
import csv, glob, json, sys, collections
out, G = sys.argv[1], int(sys.argv[2])
def per_kernel(d, names):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, d), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_gram_bitslice" not in r["Kernel_Name"]: continue
            acc[r["Counter_Name"]]["sum"] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add(r["Dispatch_Id"])
    return {k: (v["sum"], len(disp[k])) for k, v in acc.items()}
dur, calls = 0.0, 0
for f in glob.glob("%s/stats_%d/**/*kernel_stats.csv" % (out, G), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_gram_bitslice" in r["Name"]:
            dur += float(r["TotalDurationNs"]); calls += int(r["Calls"])
fetch = per_kernel("fetch_%d" % G, ["FETCH_SIZE"]); tcc = per_kernel("tcc_%d" % G, [])
line = json.load(open("%s/bench_%d.json" % (out, G)))
# matrices' worth of hot-kernel launches in one run of the program: 1 warm-up + 2 timed steps, plus bench.py's own
# kernel timing afterwards: 3 whole matrices on one context, 3 x rank 0's share (1/G of the work) otherwise
meq = 6.0 if G == 1 else 3.0 + 3.0 / G
fs, fl = fetch.get("FETCH_SIZE", (0, 1))
print("%d context(s): ms_per_step %.2f | hot kernel: %d launches, %.2f ms per WHOLE matrix | FETCH_SIZE x2 (gfx950): %.2f GB "
      "per whole matrix" % (G, line["ms_per_step"], calls, dur / 1e6 / meq, 2 * fs * 1024 / 1e9 / meq))
for k, (s, n) in sorted(tcc.items()):
    print("    %s per whole matrix: %.4g" % (k, s / meq))
PY
done
cat "$TXT"
