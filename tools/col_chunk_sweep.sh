# Work-item order (GPU box): plain tile-major (GKM_COL_CHUNK=0) against (column chunk, tile) entries of several chunk sizes:
# kernel ms + parity of the line, then L2 fill bytes (FETCH_SIZE, separate rocprofv3 pass) for the plain and the default order.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/colchunk; mkdir -p $O
for wl in c2 peaks c5; do
  for ch in 0 default 1024 2048 4096; do
    if [ "$ch" = default ]; then unset GKM_COL_CHUNK; else export GKM_COL_CHUNK=$ch; fi
    echo "$wl chunk $ch: $(python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-also 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.2f ms parity %s" % (d["roofline"]["kernel_ms"], d["parity"]["ok"]))')"
  done
done
unset GKM_COL_CHUNK
for wl in c2 peaks; do
  for ch in 0 default; do
    if [ "$ch" = default ]; then unset GKM_COL_CHUNK; else export GKM_COL_CHUNK=$ch; fi
    rocprofv3 --pmc FETCH_SIZE -d $O/fetch_${wl}_$ch --output-format csv -- python3 bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-also > /dev/null 2> $O/fetch_${wl}_$ch.err
    python3 - "$O/fetch_${wl}_$ch" "$wl chunk $ch" <<'PY'
import csv, glob, sys
tot, n = 0.0, set()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_gram_bitslice" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
print("%s: FETCH_SIZE %.3g KiB per launch -> %.2f GB read (x2 per the guide)" % (sys.argv[2], tot / max(1, len(n)), 2 * tot / max(1, len(n)) * 1024 / 1e9))
PY
  done
done
