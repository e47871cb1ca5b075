#!/bin/bash
# PMC counters of the hot kernel for one library build (GPU box):  tools/pmc_ab.sh <name> [<lib.so>|default] [workload]
#   -> gpurun_out/pmc_ab/<name>_<workload>.txt  (per-launch averages of the counters, via tools/pmc_sum.py)
# Counter passes only (never combined with tracing); the program itself follows `--`.
set -u
NAME=$1; LIB=${2:-default}; WL=${3:-peaks}
OUT=gpurun_out/pmc_ab/${NAME}_${WL}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
if [ "$LIB" != "default" ]; then export GKM_LIB_PATH="$LIB"; fi
P1="bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-also"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc_inst" --output-format csv -- python3 $P1 > /dev/null 2> "$OUT/pmc_inst.err" || exit 1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD -d "$OUT/pmc_wait" --output-format csv -- python3 $P1 > /dev/null 2> "$OUT/pmc_wait.err" || exit 1
python3 tools/pmc_sum.py "$OUT" > "gpurun_out/pmc_ab/${NAME}_${WL}.txt"
cat "gpurun_out/pmc_ab/${NAME}_${WL}.txt"
