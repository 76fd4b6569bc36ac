mkdir -p gpurun_out/s3 gpurun_out/oracle_cache
export HM_ORACLE_RECORD=$PWD/gpurun_out/oracle_cache
timeout 1200 python -u -m pytest tests/test_gpu_cli.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py "tests/test_gpu_round5.py::test_host_pacing_off_is_a_pure_enqueue_with_the_same_bits" tests/test_gpu_bench_contract.py::test_bench_two_ranks_real_job_on_one_gpu tests/test_gpu_bench_contract.py::test_bench_started_bare_spawns_ranks_that_run_the_real_job tests/test_gpu_round6.py -m gpu -q --timeout=400 --durations=15 > gpurun_out/s3/t3.txt 2>&1; tail -30 gpurun_out/s3/t3.txt
ls gpurun_out/oracle_cache | wc -l
