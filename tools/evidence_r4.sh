set -e
O=gpurun_out/r4e
mkdir -p $O
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r4_bench_line_c2.json 2> $O/bench_c2.err
timeout -k 10 200 python3 bench.py --workload peaks --steps 5 --warmup 2 --no-cpu-baseline > $O/r4_bench_line_peaks.json 2> $O/bench_peaks.err
timeout -k 10 200 python3 bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline > $O/r4_bench_line_c5.json 2> $O/bench_c5.err
timeout -k 10 300 python3 bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline > $O/r4_bench_line_c3.json 2> $O/bench_c3.err
GKM_BENCH_SHARE_GPU=1 GKM_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 5 --warmup 2 > $O/r4_bench_line_gpus2_auto_rehearsal_one_gpu.json 2> $O/bench_gpus2.err
GKM_BENCH_SHARE_GPU=1 GKM_BENCH_BACKEND=gloo GKM_BENCH_CABI_TIMEOUT=2 timeout -k 10 300 python3 bench.py --gpus 2 --steps 5 --warmup 2 > $O/r4_bench_line_gpus2_cabi_killed_fallback_rehearsal_one_gpu.json 2> $O/bench_gpus2_kill.err
GKM_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also --no-end-to-end > $O/r4_bench_line_one_rank_rccl.json 2> $O/bench_rccl1.err
timeout -k 10 300 python3 tools/boundary_ab.py --rounds 2 --trace > $O/r4_boundary_ab_c2.txt 2>&1
timeout -k 10 200 python3 tools/boundary_ab.py --rounds 1 --trace --threads 1 --settings default "GKM_BLOCK_FRACTIONS=0.5,0.25,0.125,0.0625,0.03" > $O/r4_boundary_ab_c2_one_thread.txt 2>&1
timeout -k 10 200 tools/boundary_timeline.sh c2 $O/timeline_c2 > /dev/null 2>&1; cp $O/timeline_c2/timeline.txt $O/r4_boundary_timeline_c2.txt
(tools/anyorder_probe) > $O/r4_anyorder_probe.txt 2>&1
echo evidence1 done
