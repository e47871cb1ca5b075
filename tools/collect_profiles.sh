#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun).
#   tools/collect_profiles.sh r2 [workload]   -> gpurun_out/prof_r2_<workload>/{stats,pmc_*}/... + summary/
# workload: c2 (default, the headline), peaks, c5 ... (bench.py --workload)
# Counters are collected in their own runs (never combined with tracing), HBM counters in
# separate passes as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit
# one pass).  The program itself follows `--` (no wrapper scripts between rocprofv3 and python3).
set -u
R=${1:-r2}
WL=${2:-c2}
OUT=gpurun_out/prof_${R}_${WL}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
ARGS="bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-also"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" --output-format csv -- python3 $ARGS > "$OUT/bench_stats.json" 2> "$OUT/bench_stats.err" || exit 1
P1="bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-also"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc_inst" --output-format csv -- python3 $P1 > /dev/null 2> "$OUT/pmc_inst.err" || exit 1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD -d "$OUT/pmc_wait" --output-format csv -- python3 $P1 > /dev/null 2> "$OUT/pmc_wait.err" || exit 1
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" --output-format csv -- python3 $P1 > /dev/null 2> "$OUT/pmc_fetch.err" || exit 1
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" --output-format csv -- python3 $P1 > /dev/null 2> "$OUT/pmc_write.err" || exit 1
python3 tools/summarize_profiles.py "$OUT" "$R" "$WL"
