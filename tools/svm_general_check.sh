#!/bin/bash
# The general C-SVC (k_smo_general) after a change (GPU box): its tests, cross-validation wall of the headline matrix's
# 5 folds in every variant (state in LDS / in global memory, 1024 threads, with shrinking, k_smo for comparison), the phase
# timers of the -DSVM_PROF build (tools/build_variant.sh svmprof "-DSVM_PROF" first) and the randomised sweep against
# scikit-learn.      tools/svm_general_check.sh <out dir under gpurun_out/> [fuzz seconds]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${1:-svm_general}; FUZZ=${2:-100}
mkdir -p "$OUT"; cd "$OUT"
timeout -k 10 300 python -m pytest $R/tests/test_svm_gpu.py -m gpu -x -q > tests.log 2>&1
timeout -k 10 120 python $R/tools/svm_bench.py --general > general.log 2>&1
GKM_SVM_GEN_LDS=0 timeout -k 10 120 python $R/tools/svm_bench.py --general > general_nolds.log 2>&1
GKM_SVM_GEN_T=1024 timeout -k 10 120 python $R/tools/svm_bench.py --general > general_1024.log 2>&1
timeout -k 10 120 python $R/tools/svm_bench.py --shrinking 1 > shrink.log 2>&1
timeout -k 10 120 python $R/tools/svm_bench.py > fast.log 2>&1
if [ -f $R/build_variants/lib_svmprof.so ]; then
  GKM_LIB_PATH=$R/build_variants/lib_svmprof.so timeout -k 10 120 python $R/tools/svm_bench.py --general > prof.log 2>&1
  GKM_LIB_PATH=$R/build_variants/lib_svmprof.so timeout -k 10 120 python $R/tools/svm_bench.py > prof_fast.log 2>&1
fi
timeout -k 10 $((FUZZ + 100)) python $R/tools/fuzz_svm.py --shrinking --seconds $FUZZ > fuzz.log 2>&1
GKM_SVM_GEN_LDS=0 timeout -k 10 200 python $R/tools/fuzz_svm.py --shrinking --seconds 50 > fuzz_nolds.log 2>&1
for f in *.log; do echo "== $f"; tail -n 2 $f; done
grep -h "cycles/iter" prof.log prof_fast.log | sort | uniq
