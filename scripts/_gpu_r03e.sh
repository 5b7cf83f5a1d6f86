set -x
O=gpurun_out/r03e; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_knn_ties.py tests/test_gpu_knn_symmetric.py tests/test_gpu_parity.py tests/test_gpu_precluster.py tests/test_gpu_edges.py tests/test_gpu_cfg45.py -m gpu -x -q > $O/tests_knn.log 2>&1; tail -15 $O/tests_knn.log
timeout 900 python -m pytest tests/test_cli_gpu.py -m gpu -x -q > $O/tests_cli.log 2>&1; tail -8 $O/tests_cli.log
export BENCH_ONLY_ONCE=1 BENCH_KERNEL_ONLY=1
timeout 900 bash scripts/profile_cmd.sh r03_cfg4 'pair_kernel' stats,sq,stall_a,tcc,fetch,write -- python3 scripts/bench_modes.py cfg4 > $O/prof_cfg4.log 2>&1; tail -5 $O/prof_cfg4.log
timeout 1800 bash scripts/profile_cmd.sh r03_cfg5 'pair_kernel|topk_merge' stats,sq,stall_a,tcc,fetch,write -- python3 scripts/bench_modes.py cfg5full_r > $O/prof_cfg5.log 2>&1; tail -5 $O/prof_cfg5.log
timeout 600 bash scripts/profile_cmd.sh r03_sketch 'nthash' stats,sq,stall_a,fetch,write -- python3 scripts/bench_sketch.py 512 2000000 > $O/prof_sketch.log 2>&1; tail -5 $O/prof_sketch.log
timeout 900 bash scripts/profile_cmd.sh r03_precluster 'pair_cand|topk_kernel' stats,sq,stall_a,tcc,fetch,write -- python3 scripts/bench_precluster.py > $O/prof_precluster.log 2>&1; tail -5 $O/prof_precluster.log
timeout 900 bash scripts/profile_cmd.sh r03_candgen 'cand_|first_greater|pair_cand' stats,sq,stall_a,fetch,write -- python3 scripts/bench_shared_bins.py > $O/prof_candgen.log 2>&1; tail -5 $O/prof_candgen.log
ls gpurun_out/prof_r03_*/summary.md
