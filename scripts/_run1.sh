mkdir -p gpurun_out/s3
timeout 300 python -u -m pytest tests/test_gpu_round6.py "tests/test_gpu_arch.py::test_instance_groups_share_one_generic_decoder_without_sharing_scratch" tests/test_gpu_configs.py -m gpu -q --timeout=120 --durations=10 > gpurun_out/s3/t1.txt 2>&1; tail -15 gpurun_out/s3/t1.txt
HORTIHIP_LIB=$PWD/hortimapping_amd/variants/libhortihip_k1ptrace.so timeout 120 python scripts/gpu_trace_k1p.py 64 0 > gpurun_out/s3/k1p_trace_fwd.txt 2>&1
HORTIHIP_LIB=$PWD/hortimapping_amd/variants/libhortihip_k1ptrace.so timeout 120 python scripts/gpu_trace_k1p.py 64 1 > gpurun_out/s3/k1p_trace_fb.txt 2>&1
cat gpurun_out/s3/k1p_trace_fwd.txt
timeout 200 python bench.py --shipped-only configs4_lab_pepper_berry --steps 5 > gpurun_out/s3/c4_conc.json 2>gpurun_out/s3/c4_conc.err
HM_SERIAL_GROUPS=1 timeout 200 python bench.py --shipped-only configs4_lab_pepper_berry --steps 5 > gpurun_out/s3/c4_serial.json 2>gpurun_out/s3/c4_serial.err
timeout 200 python bench.py --shipped-only configs4_lab_pepper_berry --steps 5 > gpurun_out/s3/c4_conc2.json 2>/dev/null
for f in c4_conc c4_serial c4_conc2; do python -c "
import json,sys; d=json.loads(open('gpurun_out/s3/$f.json').read().strip().splitlines()[-1]); print('$f', d.get('value'), d.get('ms_per_fruit'))"; done
timeout 400 python scripts/logistic_screen_bound.py > gpurun_out/s3/logistic_bound.txt 2>gpurun_out/s3/logistic_bound.err; cat gpurun_out/s3/logistic_bound.txt; tail -3 gpurun_out/s3/logistic_bound.err
