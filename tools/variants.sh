#!/bin/bash
# Timing ablations of the hot kernel (GPU box).  The VARIANT != 0 kernels skip parts of the work and
# return WRONG results; they exist only in this separate build (-DGKM_TIMING_VARIANTS, output under
# build_variants/), never in gkmqc_amd/bin/gkmkern_pylib.so, and are loaded through GKM_LIB_PATH.
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants
make -s -C gkmqc_amd/csrc BUILD="$PWD/build_variants/obj" BIN="$PWD/build_variants" EXTRA=-DGKM_TIMING_VARIANTS "$PWD/build_variants/gkmkern_pylib.so"
for v in ${VARIANTS:-0 1 2 16 32}; do
  echo "variant $v: $(GKM_LIB_PATH=$PWD/build_variants/gkmkern_pylib.so GKM_VARIANT=$v python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"])')"
done
