#!/bin/bash
# Timeline of one warm drop-in call (GPU box): rocprofv3 kernel + memory-copy trace of tools/boundary_ab.py's worker,
# then every kernel / copy of the LAST call with its start (ms after the call's first kernel), duration and stream.
#   tools/boundary_timeline.sh [workload] [outdir]        env knobs (GKM_ONE_STREAM=1 ...) are passed through
set -u
WL=${1:-c2}
OUT=${2:-gpurun_out/timeline_$WL}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
rocprofv3 --kernel-trace --memory-copy-trace -d "$OUT/trace" --output-format csv -- python3 tools/boundary_ab.py --worker --workload $WL --calls 3 > "$OUT/worker.json" 2> "$OUT/worker.err" || { tail -5 "$OUT/worker.err"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
ev = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], "q" + r.get("Queue_Id", "?")))
for f in glob.glob(out + "/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " %s B" % r.get("Size", "?"), "copy"))
ev.sort()
# the last call = everything after the last gap of more than 20 ms without any event
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e[1] for e in ev[max(0, i - 50):i]) > 20e6:
        cut = i
ev = ev[cut:]
t0 = ev[0][0]
lines = []
for s, e, name, q in ev:
    if (e - s) > 20e3 or "gram" in name:
        lines.append("%9.3f ms  +%8.3f ms  %-6s %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, name))
gram = [(s, e) for s, e, name, q in ev if "k_gram_bitslice" in name]
busy = sum(e - s for s, e in gram) / 1e6
span = (max(e for s, e, n, q in ev) - t0) / 1e6
gaps = [(gram[i + 1][0] - gram[i][1]) / 1e6 for i in range(len(gram) - 1)]
lines.append("gram kernels: %d, busy %.2f ms, gaps between them %s ms, first event -> last event %.2f ms" % (len(gram), busy, ["%.2f" % g for g in gaps], span))
open(out + "/timeline.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-40:]))
PY
