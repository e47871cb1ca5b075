#!/bin/bash
cd $GRAFT_REPO_ROOT
for wl in c2 d600; do
for v in 0 32 2048 2080; do
  echo "$wl variant $v: $(GKM_LIB_PATH=$PWD/build_variants/lib_timing.so GKM_VARIANT=$v python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"])')"
done
done
