#!/bin/bash
# rocprofv3 PMC passes of the hot kernel for the product build and for ablation builds (GPU box):
# where do the cycles of the hit path go?   tools/pmc_variants.sh "0 32 16" [workload]
# (variants != 0: build_variants/lib_timing.so, built by tools/variants.sh from revision a4bed73 -- the ablation
# branches left the source tree in round 4)
set -u
VARS=${1:-"0 32"}
WL=${2:-c2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcv_$WL
mkdir -p "$OUT"
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
P1="bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end"
for v in $VARS; do
  export GKM_VARIANT=$v
  [ "$v" != "0" ] && export GKM_LIB_PATH=$PWD/build_variants/lib_timing.so || unset GKM_LIB_PATH
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$OUT/v${v}_a" --output-format csv -- python3 $P1 > /dev/null 2> "$OUT/v${v}_a.err" || exit 1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d "$OUT/v${v}_b" --output-format csv -- python3 $P1 > /dev/null 2> "$OUT/v${v}_b.err" || exit 1
  rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU -d "$OUT/v${v}_c" --output-format csv -- python3 $P1 > /dev/null 2> "$OUT/v${v}_c.err" || echo "pass c failed for $v"
  rocprofv3 --pmc TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum -d "$OUT/v${v}_d" --output-format csv -- python3 $P1 > /dev/null 2> "$OUT/v${v}_d.err" || echo "pass d failed for $v"
done
python3 - "$OUT" $VARS <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for v in sys.argv[2:]:
    per = collections.defaultdict(float); cnt = collections.defaultdict(set)
    for f in glob.glob("%s/v%s_*/**/*counter_collection.csv" % (out, v), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_gram_bitslice" not in r["Kernel_Name"]:
                continue
            per[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]].add(r["Dispatch_Id"])
    res[v] = {k: per[k] / max(1, len(cnt[k])) for k in per}
json.dump(res, open(out + "/summary.json", "w"), indent=1, sort_keys=True)
keys = sorted(set(k for v in res.values() for k in v))
print("%-32s" % "counter" + "".join("%16s" % ("V" + v) for v in res))
for k in keys:
    print("%-32s" % k + "".join("%16.4g" % res[v].get(k, float("nan")) for v in res))
PY
