# Last step of a round (GPU box): PMC summaries on the FINAL kernel source (bench.py refuses summaries whose source hash
# differs from the tree), then the bench lines that quote them.  Do not edit gkm_device.hip / gkm_bitslice.h / gkm_pack.h after.
set -e
for wl in c2 peaks c5; do timeout -k 10 200 bash tools/collect_profiles.sh r4 $wl > gpurun_out/collect_$wl.log 2>&1; cp gpurun_out/prof_r4_$wl/summary/r4_* profiles/; cp gpurun_out/prof_r4_$wl/bench_stats.json profiles/r4_bench_under_rocprof_$wl.json; done
O=gpurun_out/r4e
mkdir -p $O
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r4_bench_line_c2.json 2> $O/bench_c2.err
timeout -k 10 200 python3 bench.py --workload peaks --steps 5 --warmup 2 --no-cpu-baseline > $O/r4_bench_line_peaks.json 2> $O/bench_peaks.err
timeout -k 10 200 python3 bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline > $O/r4_bench_line_c5.json 2> $O/bench_c5.err
GKM_BENCH_SHARE_GPU=1 GKM_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 5 --warmup 2 > $O/r4_bench_line_gpus2_auto_rehearsal_one_gpu.json 2> $O/bench_gpus2.err
GKM_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also --no-end-to-end > $O/r4_bench_line_one_rank_rccl.json 2> $O/bench_rccl1.err
mkdir -p gpurun_out/final_profiles && cp profiles/r4_pmc_*.json profiles/r4_kernel_stats_*.csv profiles/r4_bench_under_rocprof_*.json gpurun_out/final_profiles/
python3 -c "
import json
for f in ('r4_bench_line_c2','r4_bench_line_peaks','r4_bench_line_c5','r4_bench_line_gpus2_auto_rehearsal_one_gpu','r4_bench_line_one_rank_rccl'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity']['ok'])
"
