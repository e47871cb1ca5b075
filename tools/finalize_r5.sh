# Last step of round 5 (GPU box): PMC summaries on the FINAL kernel source (bench.py refuses summaries whose source hash differs
# from the tree: bench.KERNEL_SOURCES), then the bench lines that quote them.  Do not edit the kernel sources after.
#   gpurun -- 'bash tools/finalize_r5.sh'   ->  gpurun_out/final_profiles/ (copy to profiles/), gpurun_out/r5final/
set -e
R=r5
mkdir -p gpurun_out/final_profiles
for wl in c2 peaks c5; do
  timeout -k 10 300 bash tools/collect_profiles.sh $R $wl > gpurun_out/collect_$wl.log 2>&1
  cp gpurun_out/prof_${R}_$wl/summary/${R}_* profiles/
  cp gpurun_out/prof_${R}_$wl/summary/${R}_* gpurun_out/final_profiles/
  cp gpurun_out/prof_${R}_$wl/bench_stats.json gpurun_out/final_profiles/${R}_bench_under_rocprof_$wl.json
  echo "collected $wl"
done
O=gpurun_out/r5final
mkdir -p $O
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $O/${R}_bench_line_c2.json 2> $O/bench_c2.err
timeout -k 10 200 python3 bench.py --workload peaks --steps 5 --warmup 2 --no-cpu-baseline > $O/${R}_bench_line_peaks.json 2> $O/bench_peaks.err
timeout -k 10 200 python3 bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline > $O/${R}_bench_line_c5.json 2> $O/bench_c5.err
timeout -k 10 300 python3 bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline > $O/${R}_bench_line_c3.json 2> $O/bench_c3.err
GKM_BENCH_SHARE_GPU=1 GKM_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 5 --warmup 2 > $O/${R}_bench_line_gpus2_auto_rehearsal_one_gpu.json 2> $O/bench_gpus2.err
GKM_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also --no-end-to-end > $O/${R}_bench_line_one_rank_rccl.json 2> $O/bench_rccl1.err
python3 -c "
import json
for f in ('${R}_bench_line_c2','${R}_bench_line_peaks','${R}_bench_line_c5','${R}_bench_line_c3','${R}_bench_line_gpus2_auto_rehearsal_one_gpu','${R}_bench_line_one_rank_rccl'):
    d=json.load(open('$O/'+f+'.json')); rf=d['roofline']; print(f, d['value'], d['ms_per_step'], rf['frac'], rf.get('kernel_ms'), rf.get('small_kernels_ms'), d['parity']['ok'])
"
