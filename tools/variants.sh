#!/bin/bash
# Timing ablations of the hot kernel (GPU box).  The VARIANT != 0 kernels skip parts of the work and
# return WRONG results.  Since round 4 they are not in the source tree any more: this script builds them from
# revision a4bed73 (the last one that carried them, round 3's kernel) with -DGKM_TIMING_VARIANTS into
# build_variants/, never into gkmqc_amd/bin/gkmkern_pylib.so; they are loaded through GKM_LIB_PATH.
#   VARIANT 1: hits only counted (no ring)   2: ring filled, never consumed (no trips at all)
#          16: trips without the two l-mer table loads   32: records pushed, trips skipped
set -e
cd "$(dirname "$0")/.."
[ -f build_variants/lib_timing.so ] || tools/build_variant.sh timing "-DGKM_TIMING_VARIANTS" ${VARIANT_REV:-a4bed73}
for wl in ${WORKLOADS:-c2}; do
for v in ${VARIANTS:-0 1 2 16 32}; do
  echo "$wl variant $v: $(GKM_LIB_PATH=$PWD/build_variants/lib_timing.so GKM_VARIANT=$v python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"])')"
done
done
