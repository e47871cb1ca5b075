cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/colchunk2; mkdir -p $O
for wl in peaks c5 c2; do
  for ch in 0 2560 3072 3584 4096 5120; do
    export GKM_COL_CHUNK=$ch
    t=$(python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-also 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.2f ms parity %s" % (d["roofline"]["kernel_ms"], d["parity"]["ok"]))')
    rocprofv3 --pmc FETCH_SIZE -d $O/f_${wl}_$ch --output-format csv -- python3 bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-also > /dev/null 2> $O/f_${wl}_$ch.err
    f=$(python3 - "$O/f_${wl}_$ch" <<'PY'
import csv, glob, sys
tot, n = 0.0, set()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_gram_bitslice" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
print("%.2f GB read" % (2 * tot / max(1, len(n)) * 1024 / 1e9))
PY
)
    echo "$wl chunk $ch: $t, $f"
  done
done
