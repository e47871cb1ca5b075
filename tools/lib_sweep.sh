#!/bin/bash
# A/B timing of alternative builds of gkmkern_pylib.so (build_variants/lib_*.so) against the default
run() { python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"])'; }
echo "default: $(run)"
for f in build_variants/lib_*.so; do echo "$f: $(GKM_LIB_PATH=$PWD/$f run)"; done
echo "default again: $(run)"
