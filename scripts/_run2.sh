mkdir -p gpurun_out/s3
V=$PWD/hortimapping_amd/variants
for rep in 1 2; do
timeout 100 python scripts/gpu_time_k1p.py lead2 2>/dev/null | grep -v amdgpu
for v in pace0 lead1 lead3; do HORTIHIP_LIB=$V/libhortihip_$v.so timeout 100 python scripts/gpu_time_k1p.py $v 2>/dev/null | grep -v amdgpu; done
done > gpurun_out/s3/k1p_pace_ab.txt 2>&1
cat gpurun_out/s3/k1p_pace_ab.txt
HORTIHIP_LIB=$V/libhortihip_k1ptrace.so timeout 120 python scripts/gpu_trace_k1p.py 64 0 > gpurun_out/s3/k1p_trace_fwd_paced.txt 2>&1
HORTIHIP_LIB=$V/libhortihip_k1ptrace.so timeout 120 python scripts/gpu_trace_k1p.py 64 1 > gpurun_out/s3/k1p_trace_fb_paced.txt 2>&1
cat gpurun_out/s3/k1p_trace_fwd_paced.txt
timeout 300 python -u -m pytest tests/test_gpu_f16.py tests/test_gpu_round6.py "tests/test_gpu_configs.py::test_config4_mixed_pepper_and_berry_grouping" -m gpu -q --timeout=200 --durations=5 > gpurun_out/s3/t2.txt 2>&1; tail -12 gpurun_out/s3/t2.txt
for q in 4 8; do GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --shipped-only configs4_lab_pepper_berry --steps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['configs4_lab_pepper_berry']; print('queues $q concurrent', d['value'])"
GPU_MAX_HW_QUEUES=$q HM_SERIAL_GROUPS=1 timeout 200 python bench.py --shipped-only configs4_lab_pepper_berry --steps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['configs4_lab_pepper_berry']; print('queues $q serial', d['value'])"
done
