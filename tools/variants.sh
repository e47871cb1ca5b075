#!/bin/bash
# timing experiments: bench.py under each GKM_VARIANT (debug kernels, wrong results for != 0)
for v in ${VARIANTS:-0 1 2 8}; do
  echo "variant $v: $(GKM_VARIANT=$v python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"])')"
done
