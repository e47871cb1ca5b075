#!/bin/bash
# kernel throughput (1e12 l-mer comparisons per second) for several length distributions and both word counts
run() { python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $1 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%s %.1f ms %.1f Tcmp/s" % (r["kernel"], r["kernel_ms"], r["comparisons_per_launch"]/r["kernel_ms"]/1e9))'; }
for a in "--length 300" "--length-range 150 600 -L 12 -k 8 -d 4" "--length 600" "--length 150" "--length 321"; do
  echo "$a: $(run "$a")"
done
