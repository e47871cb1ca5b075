#!/bin/bash
for cj in 4 8 16 32 64; do
  echo "GKM_CJ=$cj: $(GKM_CJ=$cj python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"])')"
done
