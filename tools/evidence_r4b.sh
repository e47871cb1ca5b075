# Round-4 evidence, second part (GPU box): the full GPU suite, the randomised sweeps on the final code, the pipeline.
O=gpurun_out/r4f
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/gputests.log
timeout -k 10 420 python3 tools/fuzz_parity.py --seconds 300 --seed 41 > $O/r4_fuzz_parity.log 2>&1; tail -2 $O/r4_fuzz_parity.log
timeout -k 10 300 python3 tools/fuzz_boundary.py --seconds 200 --seed 42 > $O/r4_fuzz_boundary.log 2>&1; tail -2 $O/r4_fuzz_boundary.log
timeout -k 10 200 python3 tools/fuzz_svm.py --seconds 60 --seed 43 > $O/r4_fuzz_svm.log 2>&1; tail -2 $O/r4_fuzz_svm.log
timeout -k 10 300 python3 tools/many_subsets.py --workload peaks --subsets 20 --repeats 10 --skip-sequential > $O/r4_many_subsets_peaks20.txt 2>&1; tail -3 $O/r4_many_subsets_peaks20.txt
timeout -k 10 200 python3 tools/first_call_profile.py 2>&1 | grep -v amdgpu.ids > $O/r4_first_call_profile_c2.txt; cat $O/r4_first_call_profile_c2.txt
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also > $O/bench_c2_pipeline.json 2> $O/bench_c2_pipeline.err; python3 -c "import json; d=json.load(open('$O/bench_c2_pipeline.json')); print(d['end_to_end'])"
